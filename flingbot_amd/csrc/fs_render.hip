// fs_render.hip -- rasteriser, vertex normals and coverage reward (placeholder until the kernels land).
#include <hip/hip_runtime.h>

#include "../../include/flingsim.h"
#include "fs_context.h"

int fs_render_env(fs_ctx *, int, unsigned char *, float *) {
    fs_set_error("fs_render: not implemented yet");
    return FS_ERR_STATE;
}
int fs_normals_env(fs_ctx *, int, float *) {
    fs_set_error("fs_get_normals: not implemented yet");
    return FS_ERR_STATE;
}
int fs_coverage_all(fs_ctx *, float *) {
    fs_set_error("fs_coverage: not implemented yet");
    return FS_ERR_STATE;
}
