// fs_action.hip -- device-side action selection.
//
// Reference: SimEnv.get_max_value_valid_action (environment/simEnv.py:560-661) sorts all P*T*(D-2g)^2 value-map entries
// (221 184 for FlingBot's 96 x 48 x 48) on the CPU and walks them in descending order, running per candidate
// get_action_params (:517-537), pixels_to_3d_positions (environment/utils.py:237-276: transform-matrix product,
// truncation to pretransform pixels, bounds, two depth look-ups + unprojection) and the arm reachability test
// (:539-558, stretchdrag's end points :624-642) until one passes.  Here every candidate is validated in parallel with the
// same float64 expressions and the answer is one masked arg-max: highest value among the valid candidates, lowest
// flattened index among equals -- exactly the candidate the reference's walk stops at.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "../../include/flingsim.h"
#include "fs_context.h"

struct FsActionCfg {
    int P, T, D, S, g, drag, place;
    int kind[FS_ACTION_MAX_PRIMITIVES];  // FS_ACTION_FLING ...
    double fx, cx;                       // compute_intrinsics: focal length, principal point (= S / 2)
    double pose[16];                     // compute_pose(...) row-major
    double left[3], right[3];
    double reach, stretchdrag_dist, grasp_height;
};

__device__ __forceinline__ bool fs_act_unproject(const FsActionCfg &c, const float *depth, int x, int y, double *out) {
    const double z = (double)depth[(size_t)y * c.S + x];  // depth_im[y, x]
    if (z == 0.0) return false;                           // the reference raises 'Invalid pick point'
    const double px = ((double)x - c.cx) * z / c.fx;
    const double py = ((double)y - c.cx) * z / c.fx;
    for (int i = 0; i < 3; ++i) out[i] = c.pose[4 * i] * px + c.pose[4 * i + 1] * py + c.pose[4 * i + 2] * z + c.pose[4 * i + 3] * 1.0;
    out[0] = -out[0];
    return true;
}

__device__ __forceinline__ bool fs_act_reach(const FsActionCfg &c, const double *base, const double *p) {
    const double dx = base[0] - p[0], dy = base[1] - p[1], dz = base[2] - p[2];
    return sqrt(dx * dx + dy * dy + dz * dz) < c.reach;
}

// validity of candidate (primitive p, transform t, pixel (y, z) in the D x D map)
__device__ bool fs_act_valid(const FsActionCfg &c, const double *mats, const float *depth, int p, int t, int y, int z) {
    const int kind = c.kind[p];
    int a0, a1, b0, b1;  // reach points (row, column) in the transformed image
    if (kind == FS_ACTION_FLING || kind == FS_ACTION_STRETCHDRAG) { a0 = y + c.g; a1 = z; b0 = y - c.g; b1 = z; }
    else if (kind == FS_ACTION_DRAG) { a0 = y; a1 = z; b0 = y + c.drag; b1 = z; }
    else { a0 = y; a1 = z; b0 = y + c.place; b1 = z; }
    if (a0 < 0 || a1 < 0 || b0 < 0 || b1 < 0 || a0 >= c.D || a1 >= c.D || b0 >= c.D || b1 >= c.D) return false;
    const double *m = mats + 9 * (size_t)t;  // get_transform_matrix(S, D, -rotation, scale), row-major
    // np.matmul([[a0, a1, 1], [b0, b1, 1]], mat)[:, :2].astype(int)
    const int pa0 = (int)((double)a0 * m[0] + (double)a1 * m[3] + 1.0 * m[6]);
    const int pa1 = (int)((double)a0 * m[1] + (double)a1 * m[4] + 1.0 * m[7]);
    const int pb0 = (int)((double)b0 * m[0] + (double)b1 * m[3] + 1.0 * m[6]);
    const int pb1 = (int)((double)b0 * m[1] + (double)b1 * m[4] + 1.0 * m[7]);
    if (pa0 < 0 || pa1 < 0 || pb0 < 0 || pb1 < 0 || pa0 >= c.S || pa1 >= c.S || pb0 >= c.S || pb1 >= c.S) return false;
    double P1[3], P2[3];
    if (!fs_act_unproject(c, depth, pa0, pa1, P1) || !fs_act_unproject(c, depth, pb0, pb1, P2)) return false;
    bool reachable;
    if (kind == FS_ACTION_FLING || kind == FS_ACTION_STRETCHDRAG)
        reachable = fs_act_reach(c, c.left, P1) && fs_act_reach(c, c.right, P2);
    else
        reachable = (fs_act_reach(c, c.left, P1) && fs_act_reach(c, c.left, P2)) ||
                    (fs_act_reach(c, c.right, P1) && fs_act_reach(c, c.right, P2));
    if (kind == FS_ACTION_STRETCHDRAG) {
        P1[1] = c.grasp_height; P2[1] = c.grasp_height;
        const double e0 = P1[0] - P2[0], e1 = P1[1] - P2[1], e2 = P1[2] - P2[2];
        // np.cross(e, [0, 1, 0])
        double d0 = e1 * 0.0 - e2 * 1.0, d1 = e2 * 0.0 - e0 * 0.0, d2 = e0 * 1.0 - e1 * 0.0;
        const double nrm = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        d0 = c.stretchdrag_dist * d0 / nrm; d1 = c.stretchdrag_dist * d1 / nrm; d2 = c.stretchdrag_dist * d2 / nrm;
        const double L[3] = {P1[0] + d0, P1[1] + d1, P1[2] + d2}, R[3] = {P2[0] + d0, P2[1] + d1, P2[2] + d2};
        reachable = (fs_act_reach(c, c.left, L) && fs_act_reach(c, c.right, R)) && reachable;
    }
    return reachable;
}

struct FsActionBest { float val; long long idx; };

__device__ __forceinline__ bool fs_act_better(float v, long long i, float bv, long long bi) {
    return bi < 0 || v > bv || (v == bv && i < bi);
}

__global__ __launch_bounds__(256) void fs_k_action_scan(const float *values, const float *depth, const double *mats,
                                                        FsActionCfg c, FsActionBest *block_best) {
    __shared__ float sv[256];
    __shared__ long long si[256];
    const int W = c.D - 2 * c.g;
    const long long total = (long long)c.P * c.T * W * W;
    float bv = 0.0f;
    long long bi = -1;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long long)gridDim.x * blockDim.x) {
        const int zz = (int)(k % W), yy = (int)((k / W) % W);
        const int t = (int)((k / ((long long)W * W)) % c.T), p = (int)(k / ((long long)W * W * c.T));
        const int y = yy + c.g, z = zz + c.g;
        const float v = values[(((size_t)p * c.T + t) * c.D + y) * c.D + z];
        if (!(v == v)) continue;                      // NaN never equals a sorted value in the reference's walk
        if (!fs_act_better(v, k, bv, bi)) continue;   // cannot win: skip the validity work
        if (fs_act_valid(c, mats, depth, p, t, y, z)) { bv = v; bi = k; }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const long long oi = si[threadIdx.x + s];
            if (oi >= 0 && fs_act_better(sv[threadIdx.x + s], oi, sv[threadIdx.x], si[threadIdx.x])) {
                sv[threadIdx.x] = sv[threadIdx.x + s];
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { block_best[blockIdx.x].val = sv[0]; block_best[blockIdx.x].idx = si[0]; }
}

__global__ __launch_bounds__(256) void fs_k_action_final(const FsActionBest *block_best, int n_blocks, FsActionBest *out) {
    __shared__ float sv[256];
    __shared__ long long si[256];
    float bv = 0.0f;
    long long bi = -1;
    for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) {
        const FsActionBest x = block_best[b];
        if (x.idx >= 0 && fs_act_better(x.val, x.idx, bv, bi)) { bv = x.val; bi = x.idx; }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const long long oi = si[threadIdx.x + s];
            if (oi >= 0 && fs_act_better(sv[threadIdx.x + s], oi, sv[threadIdx.x], si[threadIdx.x])) {
                sv[threadIdx.x] = sv[threadIdx.x + s];
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out->val = sv[0]; out->idx = si[0]; }
}

#define FS_ACTION_BLOCKS 1024

extern "C" size_t fs_select_action_work_bytes(int n_transforms) {
    if (n_transforms < 0) return 0;
    return sizeof(double) * 9 * (size_t)n_transforms + sizeof(FsActionBest) * (FS_ACTION_BLOCKS + 1) + 512;
}

extern "C" int fs_select_action(const float *d_values, int n_primitives, const int *primitive_kinds, int n_transforms,
                                int obs_dim, int pix_grasp_dist, int pix_drag_dist, int pix_place_dist,
                                const double *transform_mats, const float *d_depth, int depth_dim, double focal_length,
                                const double *pose_matrix, const double *left_arm_base, const double *right_arm_base,
                                double reach_distance_limit, double stretchdrag_dist, double grasp_height,
                                long long *best_index_out, float *best_value_out, void *d_work, void *stream) {
    if (!d_values || !primitive_kinds || !transform_mats || !d_depth || !pose_matrix || !left_arm_base || !right_arm_base ||
        !best_index_out || !d_work || n_primitives <= 0 || n_primitives > FS_ACTION_MAX_PRIMITIVES || n_transforms <= 0 ||
        obs_dim <= 2 * pix_grasp_dist || pix_grasp_dist < 0 || depth_dim <= 0) {
        fs_set_error("fs_select_action: bad arguments");
        return FS_ERR_ARG;
    }
    FsActionCfg c;
    memset(&c, 0, sizeof(c));
    c.P = n_primitives; c.T = n_transforms; c.D = obs_dim; c.S = depth_dim;
    c.g = pix_grasp_dist; c.drag = pix_drag_dist; c.place = pix_place_dist;
    for (int p = 0; p < n_primitives; ++p) {
        if (primitive_kinds[p] < FS_ACTION_FLING || primitive_kinds[p] > FS_ACTION_PLACE) {
            fs_set_error("fs_select_action: unknown primitive kind");
            return FS_ERR_ARG;
        }
        c.kind[p] = primitive_kinds[p];
    }
    c.fx = focal_length; c.cx = (double)depth_dim / 2.0;
    memcpy(c.pose, pose_matrix, sizeof(c.pose));
    memcpy(c.left, left_arm_base, sizeof(c.left));
    memcpy(c.right, right_arm_base, sizeof(c.right));
    c.reach = reach_distance_limit; c.stretchdrag_dist = stretchdrag_dist; c.grasp_height = grasp_height;
    hipStream_t st = (hipStream_t)stream;
    double *d_mats = (double *)d_work;
    FsActionBest *d_best = (FsActionBest *)((char *)d_work + ((sizeof(double) * 9 * (size_t)n_transforms + 255) & ~(size_t)255));
    hipError_t err = hipMemcpyAsync(d_mats, transform_mats, sizeof(double) * 9 * (size_t)n_transforms, hipMemcpyHostToDevice, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);  // pageable source
    if (!fs_hip_ok(err, "fs_select_action upload")) return FS_ERR_HIP;
    const int W = obs_dim - 2 * pix_grasp_dist;
    const long long total = (long long)n_primitives * n_transforms * W * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > FS_ACTION_BLOCKS) blocks = FS_ACTION_BLOCKS;
    hipLaunchKernelGGL(fs_k_action_scan, dim3(blocks), dim3(256), 0, st, d_values, d_depth, d_mats, c, d_best);
    hipLaunchKernelGGL(fs_k_action_final, dim3(1), dim3(256), 0, st, d_best, blocks, d_best + FS_ACTION_BLOCKS);
    FsActionBest h;
    err = hipMemcpyAsync(&h, d_best + FS_ACTION_BLOCKS, sizeof(h), hipMemcpyDeviceToHost, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (!fs_hip_ok(err, "fs_select_action download")) return FS_ERR_HIP;
    *best_index_out = h.idx;
    if (best_value_out) *best_value_out = h.val;
    return FS_OK;
}
