// fs_observe.hip -- the observation stage between pyflex.render and prepare_image on the device.
//
// Reference (all host-side Python over a 4 MB download of the rendered frame):
//   get_image            environment/flex_utils.py:418-427   flip rows, drop alpha, cv2.resize (INTER_LINEAR) to image_dim
//   SimEnv.get_cloth_mask environment/simEnv.py:699-708      cv2.cvtColor(RGB2HSV) -> inRange((0,0,0),(100,100,100)) -> == 0
//   get_largest_component environment/utils.py:585-601       skimage.measure.label (8-connectivity), biggest foreground one
//   SimEnv.get_obs        environment/simEnv.py:710-737      bounding box of that component -> adaptive scale
//   preprocess_obs        environment/utils.py:579-582       cat(rgb / 255, depth) -> float32 [4, S, S]
// Here the frame never leaves the GPU: fs_k_obs_resize reads the rendered frame (bottom-up RGBA + depth) and writes the
// observation tensor, the raw cloth mask and the initial component labels; the labelling runs as min-label propagation with
// pointer jumping until a pass changes nothing; three small kernels pick the largest component (ties: the one whose
// first pixel comes first in raster order, like skimage's label order + Python's stable sort) and its bounding box.
//
// cv2 / skimage are absent from this image: the arithmetic follows OpenCV's documented scalar algorithms (resize.cpp
// HResizeLinear / VResizeLinear: 11-bit fixed-point taps for 8-bit images, float taps for 32-bit ones; color_hsv.cpp
// RGB2HSV_b: hsv_shift 12, sdiv / hdiv180 tables) -- restated on the CPU in oracle/observe.py, which the GPU path equals bit
// for bit (tests/test_observe_gpu.py); parity with the reference's own cv2 build stays unpinned.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/flingsim.h"
#include "fs_context.h"

#define HIP_TRY(call)                                     \
    do {                                                  \
        if (!fs_hip_ok((call), #call)) return FS_ERR_HIP; \
    } while (0)

struct FsObsResult {
    unsigned long long best;  // (count << 32) | (0xffffffff - root): max = most pixels, then lowest root
    int xmin, xmax, ymin, ymax;  // rows (x) and columns (y) of the winner, as np.where(mask) names them
    int changed;
    int pad;
};

// cv::resize INTER_LINEAR tap of destination index d: source index and fraction
__device__ __forceinline__ void fs_linear_tap(int d, double scale, int src, int &s, float &f) {
    f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= src - 1) { f = 0.0f; s = src - 1; }
}

// saturate_cast<int>((num << 12) / (mul * i)), round half to even like cvRound
__device__ __forceinline__ int fs_hsv_div(int num, double mul, int i) {
    return i == 0 ? 0 : (int)rint((double)(num << 12) / (mul * (double)i));
}

__device__ __forceinline__ bool fs_is_cloth(int r, int g, int b) {
    const int v = max(max(r, g), b), vmin = min(min(r, g), b), diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int s = (diff * fs_hsv_div(255, 1.0, v) + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + (~vg & (r - g + 4 * diff))));
    h = (h * fs_hsv_div(180, 6.0, diff) + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    return !(h <= 100 && s <= 100 && v <= 100);  // (inRange == 0)
}

__global__ __launch_bounds__(256) void fs_k_obs_resize(const unsigned char *__restrict__ rgba, const float *__restrict__ depth,
                                                       int W, int H, int S, float *__restrict__ obs,
                                                       int *__restrict__ label, int *__restrict__ count) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S) return;
    const int dy = idx / S, dx = idx % S;
    int r, g, b;
    float d;
    if (W == S && H == S) {
        const size_t at = (size_t)(H - 1 - dy) * W + dx;
        r = rgba[4 * at]; g = rgba[4 * at + 1]; b = rgba[4 * at + 2];
        d = depth[at];
    } else {
        int sx, sy;
        float fx, fy;
        fs_linear_tap(dx, (double)W / (double)S, W, sx, fx);
        fs_linear_tap(dy, (double)H / (double)S, H, sy, fy);
        const int sx1 = min(sx + 1, W - 1), sy1 = min(sy + 1, H - 1);
        // rows of the observation are the rendered rows in reverse order (np.flip(..., 0))
        const size_t r0 = (size_t)(H - 1 - sy) * W, r1 = (size_t)(H - 1 - sy1) * W;
        const int ax0 = (int)rintf((1.0f - fx) * 2048.0f), ax1 = (int)rintf(fx * 2048.0f);
        const int by0 = (int)rintf((1.0f - fy) * 2048.0f), by1 = (int)rintf(fy * 2048.0f);
        int out[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int s0 = rgba[4 * (r0 + sx) + c] * ax0 + rgba[4 * (r0 + sx1) + c] * ax1;
            const int s1 = rgba[4 * (r1 + sx) + c] * ax0 + rgba[4 * (r1 + sx1) + c] * ax1;
            const int v = (((by0 * (s0 >> 4)) >> 16) + ((by1 * (s1 >> 4)) >> 16) + 2) >> 2;
            out[c] = min(max(v, 0), 255);
        }
        r = out[0]; g = out[1]; b = out[2];
        const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
        const float d0 = depth[r0 + sx] * a0 + depth[r0 + sx1] * a1;
        const float d1 = depth[r1 + sx] * a0 + depth[r1 + sx1] * a1;
        d = d0 * b0 + d1 * b1;
    }
    const size_t plane = (size_t)S * S;
    obs[idx] = (float)r / 255.0f;
    obs[plane + idx] = (float)g / 255.0f;
    obs[2 * plane + idx] = (float)b / 255.0f;
    obs[3 * plane + idx] = d;
    label[idx] = fs_is_cloth(r, g, b) ? idx : -1;
    count[idx] = 0;
}

// one pass of min-label propagation over the 8-neighbourhood + pointer jumping; labels only ever decrease
// (blockIdx.y = observation of a batch: its labels / counts / result follow the first one's at a stride of S * S / 1)
__global__ __launch_bounds__(256) void fs_k_obs_ccl(int S, int *label, FsObsResult *res) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S) return;
    label += (size_t)blockIdx.y * S * S;
    res += blockIdx.y;
    const int own = label[idx];
    if (own < 0) return;
    const int y = idx / S, x = idx % S;
    int m = own;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if ((dy | dx) != 0 && yy >= 0 && yy < S && xx >= 0 && xx < S) {
                const int l = label[yy * S + xx];
                if (l >= 0 && l < m) m = l;
            }
        }
    for (int hop = 0; hop < 64; ++hop) {  // follow the chain towards the root
        const int up = label[m];
        if (up >= m || up < 0) break;
        m = up;
    }
    if (m < own) {
        atomicMin(&label[idx], m);
        atomicMin(&label[own], m);  // pull the old root along
        res->changed = 1;
    }
}

__global__ __launch_bounds__(256) void fs_k_obs_count(int S, const int *__restrict__ label, int *count) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    label += (size_t)blockIdx.y * S * S;
    count += (size_t)blockIdx.y * S * S;
    const int l = idx < S * S ? label[idx] : -1;
    // a wavefront's pixels mostly share one root: one atomic per distinct root and wavefront instead of one per pixel
    bool todo = l >= 0;
    while (__any(todo)) {
        const unsigned long long live = __ballot(todo);
        const int leader = __ffsll((long long)live) - 1;
        const int root = __shfl(l, leader);
        const unsigned long long same = __ballot(todo && l == root);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&count[root], __popcll(same));
        if (l == root) todo = false;
    }
}

__global__ __launch_bounds__(256) void fs_k_obs_best(int S, const int *__restrict__ count, FsObsResult *res) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S) return;
    count += (size_t)blockIdx.y * S * S;
    res += blockIdx.y;
    const int c = count[idx];
    if (c > 0) atomicMax(&res->best, ((unsigned long long)c << 32) | (unsigned long long)(0xffffffffu - (unsigned)idx));
}

__global__ __launch_bounds__(256) void fs_k_obs_bbox(int S, const int *__restrict__ label, FsObsResult *res,
                                                     unsigned char *__restrict__ mask) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S) return;
    label += (size_t)blockIdx.y * S * S;
    res += blockIdx.y;
    if (mask) mask += (size_t)blockIdx.y * S * S;
    const unsigned long long best = res->best;
    const int root = best ? (int)(0xffffffffu - (unsigned)(best & 0xffffffffull)) : -2;
    const bool in = label[idx] == root;
    if (mask) mask[idx] = in ? 1 : 0;
    if (in) {
        const int x = idx / S, y = idx % S;
        atomicMin(&res->xmin, x); atomicMax(&res->xmax, x);
        atomicMin(&res->ymin, y); atomicMax(&res->ymax, y);
    }
}

extern "C" {

size_t fs_observe_work_bytes(int image_dim) {
    if (image_dim <= 0 || image_dim > 4096) return 0;
    return size_t(8) * image_dim * image_dim + 256;
}

int fs_observe(fs_ctx *ctx, int env, int image_dim, float *d_obs, unsigned char *d_mask, int *bbox, void *d_work) {
    if (!ctx || env < 0 || env >= ctx->n_envs || !ctx->envs[env].has_scene || image_dim <= 0 || image_dim > 4096 ||
        !d_obs || !bbox || !d_work) {
        fs_set_error("fs_observe: bad arguments");
        return FS_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned char *d_rgba = nullptr;
    float *d_depth = nullptr;
    int rc = fs_render_device(ctx, env, &d_rgba, &d_depth);
    if (rc != FS_OK) return rc;
    const FsEnv &e = ctx->envs[env];
    const int S = image_dim, W = e.cam.width, H = e.cam.height;
    const size_t px = size_t(S) * S;
    int *label = (int *)d_work, *count = label + px;
    FsObsResult *res = (FsObsResult *)(count + px);
    hipStream_t st = ctx->stream;
    FsObsResult init;
    init.best = 0; init.xmin = init.ymin = 0x7fffffff; init.xmax = init.ymax = -1; init.changed = 0; init.pad = 0;
    FsObsResult *h = (FsObsResult *)fs_stage(ctx, sizeof(FsObsResult));
    if (!h) return FS_ERR_HIP;
    *h = init;
    HIP_TRY(hipMemcpyAsync(res, h, sizeof(FsObsResult), hipMemcpyHostToDevice, st));
    const dim3 grid((unsigned)((px + 255) / 256)), block(256);
    hipLaunchKernelGGL(fs_k_obs_resize, grid, block, 0, st, d_rgba, d_depth, W, H, S, d_obs, label, count);
    // label propagation: passes in batches, until a whole batch's last pass changed nothing
    for (int round = 0; round < 4096; ++round) {
        for (int k = 0; k < 7; ++k) hipLaunchKernelGGL(fs_k_obs_ccl, grid, block, 0, st, S, label, res);
        HIP_TRY(hipMemsetAsync(&res->changed, 0, sizeof(int), st));
        hipLaunchKernelGGL(fs_k_obs_ccl, grid, block, 0, st, S, label, res);
        HIP_TRY(hipMemcpyAsync(h, res, sizeof(FsObsResult), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (!h->changed) break;
    }
    hipLaunchKernelGGL(fs_k_obs_count, grid, block, 0, st, S, label, count);
    hipLaunchKernelGGL(fs_k_obs_best, grid, block, 0, st, S, count, res);
    hipLaunchKernelGGL(fs_k_obs_bbox, grid, block, 0, st, S, label, res, d_mask);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h, res, sizeof(FsObsResult), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int n_px = (int)(h->best >> 32);
    bbox[0] = n_px ? h->xmin : -1; bbox[1] = n_px ? h->xmax : -1;
    bbox[2] = n_px ? h->ymin : -1; bbox[3] = n_px ? h->ymax : -1;
    bbox[4] = n_px;
    return FS_OK;
}

// fs_observe for n episodes with the host round trips of ONE: every episode is rendered and resized in turn (the frame
// buffer is shared), then the labelling passes, the component count and the bounding boxes run for all of them per launch
// (blockIdx.y = observation) and the convergence flags / results come back in one copy per round.  Extra passes over
// labels that have converged change nothing, so each result equals the single call's.
//   d_obs [n][4][S][S], d_mask [n][S][S] or null, bbox [n][5], d_work: n * fs_observe_work_bytes(S) bytes.
int fs_observe_batch(fs_ctx *ctx, int n, const int *envs, int image_dim, float *d_obs, unsigned char *d_mask, int *bbox,
                     void *d_work) {
    if (!ctx || n <= 0 || !envs || image_dim <= 0 || image_dim > 4096 || !d_obs || !bbox || !d_work) {
        fs_set_error("fs_observe_batch: bad arguments");
        return FS_ERR_ARG;
    }
    for (int k = 0; k < n; ++k)
        if (envs[k] < 0 || envs[k] >= ctx->n_envs || !ctx->envs[envs[k]].has_scene) {
            fs_set_error("fs_observe_batch: bad episode");
            return FS_ERR_ARG;
        }
    HIP_TRY(hipSetDevice(ctx->device));
    const int S = image_dim;
    const size_t px = size_t(S) * S;
    int *label = (int *)d_work, *count = label + px * n;
    FsObsResult *res = (FsObsResult *)(count + px * n);
    hipStream_t st = ctx->stream;
    FsObsResult *h = (FsObsResult *)fs_stage(ctx, sizeof(FsObsResult) * n);
    if (!h) return FS_ERR_HIP;
    for (int k = 0; k < n; ++k) {
        h[k].best = 0; h[k].xmin = h[k].ymin = 0x7fffffff; h[k].xmax = h[k].ymax = -1; h[k].changed = 0; h[k].pad = 0;
    }
    HIP_TRY(hipMemcpyAsync(res, h, sizeof(FsObsResult) * n, hipMemcpyHostToDevice, st));
    const dim3 grid1((unsigned)((px + 255) / 256)), gridn((unsigned)((px + 255) / 256), (unsigned)n), block(256);
    for (int k = 0; k < n; ++k) {
        unsigned char *d_rgba = nullptr;
        float *d_depth = nullptr;
        int rc = fs_render_device(ctx, envs[k], &d_rgba, &d_depth);
        if (rc != FS_OK) return rc;
        const FsEnv &e = ctx->envs[envs[k]];
        hipLaunchKernelGGL(fs_k_obs_resize, grid1, block, 0, st, d_rgba, d_depth, e.cam.width, e.cam.height, S,
                           d_obs + size_t(4) * px * k, label + px * k, count + px * k);
    }
    for (int round = 0; round < 4096; ++round) {
        for (int q = 0; q < 7; ++q) hipLaunchKernelGGL(fs_k_obs_ccl, gridn, block, 0, st, S, label, res);
        for (int k = 0; k < n; ++k) HIP_TRY(hipMemsetAsync(&res[k].changed, 0, sizeof(int), st));
        hipLaunchKernelGGL(fs_k_obs_ccl, gridn, block, 0, st, S, label, res);
        HIP_TRY(hipMemcpyAsync(h, res, sizeof(FsObsResult) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        bool any = false;
        for (int k = 0; k < n; ++k) any = any || h[k].changed;
        if (!any) break;
    }
    hipLaunchKernelGGL(fs_k_obs_count, gridn, block, 0, st, S, label, count);
    hipLaunchKernelGGL(fs_k_obs_best, gridn, block, 0, st, S, count, res);
    hipLaunchKernelGGL(fs_k_obs_bbox, gridn, block, 0, st, S, label, res, d_mask);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h, res, sizeof(FsObsResult) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < n; ++k) {
        const int n_px = (int)(h[k].best >> 32);
        bbox[5 * k + 0] = n_px ? h[k].xmin : -1; bbox[5 * k + 1] = n_px ? h[k].xmax : -1;
        bbox[5 * k + 2] = n_px ? h[k].ymin : -1; bbox[5 * k + 3] = n_px ? h[k].ymax : -1;
        bbox[5 * k + 4] = n_px;
    }
    return FS_OK;
}

}  // extern "C"
