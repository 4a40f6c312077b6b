// fs_hostapi.hip -- host-only entry points of the C-ABI (no HIP device needed): scene builder and camera set-up.
// They let the CPU test-suite check the host logic (bit-exact topology, camera matrices) without a GPU.
#include <cstring>

#include "../../include/flingsim.h"
#include "fs_context.h"
#include "fs_camera.h"
#include "fs_sphere_mesh.h"

struct fs_host_scene { FsHostScene s; };  // (same definition in fs_capi.hip: fs_set_scene_prebuilt)

extern "C" fs_host_scene *fs_host_scene_build(const float *scene_params, int n_params, const float *verts,
                                              int n_vert_floats, const int *stretch, int n_stretch_ints, const int *bend,
                                              int n_bend_ints, const int *shear, int n_shear_ints, const int *faces,
                                              int n_face_ints) {
    fs_host_scene *h = new fs_host_scene();
    std::string err = fs_build_scene(h->s, scene_params, n_params, verts, n_vert_floats, stretch, n_stretch_ints, bend,
                                     n_bend_ints, shear, n_shear_ints, faces, n_face_ints);
    if (!err.empty()) {
        fs_set_error(err);
        delete h;
        return nullptr;
    }
    return h;
}
extern "C" void fs_host_scene_free(fs_host_scene *h) { delete h; }
extern "C" int fs_host_scene_counts(const fs_host_scene *h, int *n, int *m, int *t, int *max_deg) {
    if (!h) return FS_ERR_ARG;
    if (n) *n = h->s.n;
    if (m) *m = h->s.m;
    if (t) *t = h->s.t;
    if (max_deg) *max_deg = h->s.max_deg;
    return FS_OK;
}
extern "C" int fs_host_scene_copy(const fs_host_scene *h, int what, void *out, int n_elems) {
    if (!h || !out) return FS_ERR_ARG;
    const FsHostScene &s = h->s;
    const void *src = nullptr;
    size_t count = 0;
    FsParams p = s.params;
    float packed[32];
    switch (what) {
        case FS_SCENE_POSITIONS: src = s.pos.data(); count = s.pos.size(); break;
        case FS_SCENE_VELOCITIES: src = s.vel.data(); count = s.vel.size(); break;
        case FS_SCENE_PHASES: src = s.phase.data(); count = s.phase.size(); break;
        case FS_SCENE_SPRINGS: src = s.springs.data(); count = s.springs.size(); break;
        case FS_SCENE_SPRING_LENGTHS: src = s.spring_len.data(); count = s.spring_len.size(); break;
        case FS_SCENE_SPRING_STIFFNESS: src = s.spring_k.data(); count = s.spring_k.size(); break;
        case FS_SCENE_TRIANGLES: src = s.tris.data(); count = s.tris.size(); break;
        case FS_SCENE_TRI_NORMALS: src = s.tri_normals.data(); count = s.tri_normals.size(); break;
        case FS_SCENE_ADJ_OFFSETS: src = s.adj_off.data(); count = s.adj_off.size(); break;
        case FS_SCENE_ADJ_NEIGHBORS: src = s.adj_j.data(); count = s.adj_j.size(); break;
        case FS_SCENE_BOUNDS: {
            if (n_elems < 6) { fs_set_error("buffer too small"); return FS_ERR_ARG; }
            memcpy(out, s.scene_lower, 12);
            memcpy((char *)out + 12, s.scene_upper, 12);
            return FS_OK;
        }
        case FS_SCENE_PARAMS: {
            if (n_elems < 32) { fs_set_error("buffer too small"); return FS_ERR_ARG; }
            memset(packed, 0, sizeof(packed));
            packed[0] = (float)p.numIterations; packed[1] = (float)p.numSubsteps; packed[2] = p.dt;
            packed[3] = p.gravity[0]; packed[4] = p.gravity[1]; packed[5] = p.gravity[2];
            packed[6] = p.radius; packed[7] = p.solidRestDistance; packed[8] = p.collisionDistance;
            packed[9] = p.shapeCollisionMargin; packed[10] = p.particleCollisionMargin; packed[11] = p.dynamicFriction;
            packed[12] = p.staticFriction; packed[13] = p.particleFriction; packed[14] = p.damping;
            packed[15] = p.sleepThreshold; packed[16] = p.relaxationFactor; packed[17] = p.maxAcceleration;
            packed[18] = p.maxSpeed; packed[19] = p.restitution; packed[20] = p.adhesion; packed[21] = p.dissipation;
            packed[22] = (float)p.numPlanes; packed[23] = p.planes[0][0]; packed[24] = p.planes[0][1];
            packed[25] = p.planes[0][2]; packed[26] = p.planes[0][3]; packed[27] = (float)p.maxNeighbors;
            packed[28] = (float)p.maxContacts; packed[29] = (float)p.relaxationMode;
            memcpy(out, packed, sizeof(packed));
            return FS_OK;
        }
        case FS_SCENE_FLAGS: {
            if (n_elems < 4) { fs_set_error("buffer too small"); return FS_ERR_ARG; }
            const int flags[4] = {s.restnear_ok, s.g64_ok, s.gp_L_ok, s.gp_halvable};
            memcpy(out, flags, sizeof(flags));
            return FS_OK;
        }
        case FS_SCENE_RESTNEAR: src = s.restnear_w.data(); count = s.restnear_ok ? s.restnear_w.size() : 0; break;
        case FS_SCENE_STREAM_CODES: src = s.scode.data(); count = s.sdict_size > 0 ? s.scode.size() : 0; break;
        case FS_SCENE_STREAM_DICT: src = s.sdict.data(); count = s.sdict_size > 0 ? (size_t)4 * s.sdict_size : 0; break;
        default: fs_set_error("unknown scene array id"); return FS_ERR_ARG;
    }
    if ((size_t)n_elems < count) { fs_set_error("buffer too small"); return FS_ERR_ARG; }
    if (count) memcpy(out, src, count * 4);   // (an empty array -- the springs of a 1 x 1 cloth -- has no data pointer to copy from)
    return FS_OK;
}

// RenderScene's camera / light set-up (main.cpp:1411-1438): out[0:16] view, [16:32] proj, [32:48] lightTransform
// (all row-major, column-vector convention), [48:51] lightPos, [51:54] lightDir.
extern "C" int fs_camera_matrices(const float *cam_pos3, const float *cam_angle3, int width, int height,
                                  const float *scene_lower3, const float *scene_upper3, float *out54) {
    if (!cam_pos3 || !cam_angle3 || !scene_lower3 || !scene_upper3 || !out54 || width <= 0 || height <= 0) return FS_ERR_ARG;
    FsRasterFrame fr;
    fs_raster_setup(fr, cam_pos3, cam_angle3, width, height, scene_lower3, scene_upper3);
    memcpy(out54, fr.view, 64);
    memcpy(out54 + 16, fr.proj, 64);
    memcpy(out54 + 32, fr.light_vp, 64);
    memcpy(out54 + 48, fr.light_pos, 12);
    memcpy(out54 + 51, fr.light_dir, 12);
    return FS_OK;
}

// The mesh fs_render rasterises for ONE kinematic sphere, computed by the same functions on the host
// (fs_raster_kernels.h: fs_sphere_trig, fs_quat_axes, fs_sphere_vertex, fs_sphere_tri) = what the reference draws
// (core/mesh.cpp:858-902 + main.cpp:1739-1751).  verts / normals: float[4 * 441], tris: int[3 * 800]; each may be null.
extern "C" int fs_host_sphere_mesh(float radius, const float *prev_pos3, const float *prev_quat4, float *verts,
                                   float *normals, int *tris) {
    if (!prev_pos3 || !prev_quat4) return FS_ERR_ARG;
    FsSphereTrig trig;
    fs_sphere_trig(trig);
    float a[9];
    fs_quat_axes(prev_quat4, a);
    for (int v = 0; v < FS_SPHERE_VERTS; ++v) {
        FsVec4 p, n;
        fs_sphere_vertex(trig, a, radius, prev_pos3[0], prev_pos3[1], prev_pos3[2], v, p, n);
        if (verts) memcpy(verts + 4 * v, &p, 16);
        if (normals) memcpy(normals + 4 * v, &n, 16);
    }
    if (tris)
        for (int t = 0; t < FS_SPHERE_TRIS; ++t) fs_sphere_tri(t, tris[3 * t], tris[3 * t + 1], tris[3 * t + 2]);
    return FS_OK;
}
