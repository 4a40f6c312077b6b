// fs_loops.hip -- device-side feedback loops and reductions of the reference's manipulation primitives (SURVEY.md 8f
// row f1): wait_until_stable (environment/flex_utils.py:430-441), the height / velocity tests of lift_cloth and
// is_cloth_grasped (environment/simEnv.py:186-200, 809-813) and stretch_cloth's probe (simEnv.py:155-168).
// The reference downloads all positions / velocities through pyflex and reduces them with numpy once per simulation
// step or loop trip; here one small kernel per call produces the few numbers the host logic needs, and
// wait_until_stable runs its whole data-dependent loop without the host: a check kernel retires finished episodes from
// the launch list (id -> -1), which every solver kernel skips.
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/flingsim.h"
#include "fs_context.h"

#define HIP_TRY(call)                                     \
    do {                                                  \
        if (!fs_hip_ok((call), #call)) return FS_ERR_HIP; \
    } while (0)

// numpy's max / min propagate NaN (a NaN velocity never compares below the tolerance: wait_until_stable keeps stepping)
__device__ __forceinline__ float fs_nanmax(float a, float b) { return (a != a || b != b) ? __int_as_float(0x7fc00000) : fmaxf(a, b); }

// block-wide maximum, result valid in thread 0
__device__ __forceinline__ float fs_block_max(float v, float *red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (t < s) red[t] = fs_nanmax(red[t], red[t + s]);
        __syncthreads();
    }
    return red[0];
}

// One block per launch-list slot: the wait_until_stable test, BEFORE the step (flex_utils.py:434-437).
__global__ __launch_bounds__(256) void fs_k_stable_check(const FsEnvDev *envs, int *ids, double tol, int *steps, int *stable,
                                                         int *remaining) {
    __shared__ float red[256];
    const int slot = blockIdx.x;
    const int e = ids[slot];
    if (e < 0) return;
    const FsEnvDev &E = envs[e];
    float m = 0.0f;
    for (int i = threadIdx.x; i < E.n; i += blockDim.x) {
        const FsVec4 v = E.vel[i];
        m = fs_nanmax(m, fs_nanmax(fabsf(v.x), fs_nanmax(fabsf(v.y), fabsf(v.z))));
    }
    m = fs_block_max(m, red);
    if (threadIdx.x == 0) {
        if ((double)m < tol) {
            stable[slot] = 1;
            ids[slot] = -1;
            atomicSub(remaining, 1);
        } else {
            steps[slot] += 1;  // the step that follows
        }
    }
}

extern "C" int fs_wait_until_stable(fs_ctx *ctx, int n, const int *envs, int max_steps, double tolerance, int *steps_out,
                                    int *stable_out) {
    if (!ctx || !envs || !steps_out || !stable_out || n <= 0 || n > ctx->n_envs || max_steps < 0) {
        fs_set_error("fs_wait_until_stable: bad arguments");
        return FS_ERR_ARG;
    }
    if (const int guard_rc = fs_step_guard(ctx, "fs_wait_until_stable")) return guard_rc;
    std::vector<int> ids(envs, envs + n);
    for (int e : ids)
        if (e < 0 || e >= ctx->n_envs || !ctx->envs[e].has_scene) {
            fs_set_error("fs_wait_until_stable: bad episode id / no scene");
            return FS_ERR_ARG;
        }
    int *d_buf = (int *)fs_svc_scratch(ctx, sizeof(int) * (3 * n + 1));  // ids[n] | steps[n] | stable[n] | remaining
    if (!d_buf) return FS_ERR_HIP;
    int *d_ids = d_buf, *d_steps = d_buf + n, *d_stable = d_buf + 2 * n, *d_remaining = d_buf + 3 * n;
    std::vector<int> init(3 * n + 1, 0);
    for (int k = 0; k < n; ++k) init[k] = ids[k];
    init[3 * n] = n;
    int rc = FS_OK;
    hipError_t herr = hipMemcpyAsync(d_buf, init.data(), sizeof(int) * init.size(), hipMemcpyHostToDevice, ctx->stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(ctx->stream);  // `init` is pageable
    if (!fs_hip_ok(herr, "fs_wait_until_stable upload")) rc = FS_ERR_HIP;
    int *h_remaining = (int *)fs_stage(ctx, sizeof(int));
    for (int s = 0; s < max_steps && rc == FS_OK; ++s) {
        hipLaunchKernelGGL(fs_k_stable_check, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->d_envs, d_ids, tolerance,
                           d_steps, d_stable, d_remaining);
        rc = fs_step_ids(ctx, ids, 1, d_ids);
        if (rc == FS_OK && (s & 15) == 15) {  // every 16 steps: has everybody finished?
            if (!fs_hip_ok(hipMemcpyAsync(h_remaining, d_remaining, sizeof(int), hipMemcpyDeviceToHost, ctx->stream), "poll") ||
                !fs_hip_ok(hipStreamSynchronize(ctx->stream), "poll sync"))
                rc = FS_ERR_HIP;
            else if (*h_remaining == 0)
                break;
        }
    }
    if (rc == FS_OK) {
        std::vector<int> out(2 * n);
        herr = hipMemcpyAsync(out.data(), d_steps, sizeof(int) * 2 * n, hipMemcpyDeviceToHost, ctx->stream);
        if (herr == hipSuccess) herr = hipStreamSynchronize(ctx->stream);
        if (!fs_hip_ok(herr, "fs_wait_until_stable download")) rc = FS_ERR_HIP;
        for (int k = 0; k < n && rc == FS_OK; ++k) { steps_out[k] = out[k]; stable_out[k] = out[n + k]; }
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    return rc;
}

// ---- reductions ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fs_k_cloth_stats(const FsEnvDev *envs, const int *ids, float *out) {
    __shared__ float red[256];
    const FsEnvDev &E = envs[ids[blockIdx.x]];
    float lo = 3.402823466e+38f, hi = -3.402823466e+38f, vm = 0.0f;
    for (int i = threadIdx.x; i < E.n; i += blockDim.x) {
        const float y = E.pos[i].y;
        lo = -fs_nanmax(-lo, -y);
        hi = fs_nanmax(hi, y);
        const FsVec4 v = E.vel[i];
        vm = fs_nanmax(vm, fs_nanmax(fabsf(v.x), fs_nanmax(fabsf(v.y), fabsf(v.z))));
    }
    const float nlo = fs_block_max(-lo, red);
    __syncthreads();
    const float mhi = fs_block_max(hi, red);
    __syncthreads();
    const float mv = fs_block_max(vm, red);
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = -nlo;
        out[3 * blockIdx.x + 1] = mhi;
        out[3 * blockIdx.x + 2] = mv;
    }
}

static int upload_list(fs_ctx *ctx, int n, const int *envs, int **d_ids) {
    if (!ctx || !envs || n <= 0 || n > ctx->n_envs) {
        fs_set_error("bad episode list");
        return FS_ERR_ARG;
    }
    for (int k = 0; k < n; ++k)
        if (envs[k] < 0 || envs[k] >= ctx->n_envs || !ctx->envs[envs[k]].has_scene) {
            fs_set_error("bad episode id / no scene");
            return FS_ERR_ARG;
        }
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // h_ids may still be read by an earlier copy
    for (int k = 0; k < n; ++k) ctx->h_ids[k] = envs[k];
    ctx->d_ids_valid = false;  // (the solver's upload_ids caches what d_ids holds)
    HIP_TRY(hipMemcpyAsync(ctx->d_ids, ctx->h_ids, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
    *d_ids = ctx->d_ids;
    return FS_OK;
}

extern "C" int fs_cloth_stats(fs_ctx *ctx, int n, const int *envs, float *out, int n_floats) {
    if (!out || n_floats < 3 * n) {
        fs_set_error("fs_cloth_stats: output too small");
        return FS_ERR_ARG;
    }
    int *d_ids = nullptr;
    int rc = upload_list(ctx, n, envs, &d_ids);
    if (rc != FS_OK) return rc;
    float *d_out = (float *)fs_svc_scratch(ctx, sizeof(float) * 3 * n);
    if (!d_out) return FS_ERR_HIP;
    hipLaunchKernelGGL(fs_k_cloth_stats, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->d_envs, d_ids, d_out);
    hipError_t herr = hipMemcpyAsync(out, d_out, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, ctx->stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(ctx->stream);
    return fs_hip_ok(herr, "fs_cloth_stats") ? FS_OK : FS_ERR_HIP;
}

// stretch_cloth's probe.  All arithmetic in float32, as numpy does on the float32 arrays pyflex returns.
__global__ __launch_bounds__(256) void fs_k_stretch_probe(const FsEnvDev *envs, const int *ids, const float *mid_xz,
                                                          const float *height_thr, int *single_out, float *nearest_out) {
    __shared__ float best_d[256];
    __shared__ int best_i[256];
    __shared__ int any_neg[256], any_pos[256];  // a high particle with x >= 0 / x <= 0 exists
    const int slot = blockIdx.x, t = threadIdx.x;
    const FsEnvDev &E = envs[ids[slot]];
    const float mx = mid_xz[2 * slot], mz = mid_xz[2 * slot + 1], thr = height_thr[slot];
    float bd = 3.402823466e+38f;
    int bi = 0x7fffffff, not_all_neg = 0, not_all_pos = 0;
    for (int i = t; i < E.n; i += blockDim.x) {
        const FsVec4 p = E.pos[i];
        if (p.y > thr) {
            if (!(p.x < 0.0f)) not_all_neg = 1;
            if (!(p.x > 0.0f)) not_all_pos = 1;
        }
        const float dx = p.x - mx, dz = p.z - mz;
        const float d = sqrtf(dx * dx + dz * dz);  // np.linalg.norm of the float32 2-vector
        if (d < bd) { bd = d; bi = i; }            // ascending i within a thread: the first minimum is kept
    }
    best_d[t] = bd; best_i[t] = bi; any_neg[t] = not_all_neg; any_pos[t] = not_all_pos;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (t < s) {
            if (best_d[t + s] < best_d[t] || (best_d[t + s] == best_d[t] && best_i[t + s] < best_i[t])) {
                best_d[t] = best_d[t + s];
                best_i[t] = best_i[t + s];
            }
            any_neg[t] |= any_neg[t + s];
            any_pos[t] |= any_pos[t + s];
        }
        __syncthreads();
    }
    if (t == 0) {
        single_out[slot] = (!any_neg[0] || !any_pos[0]) ? 1 : 0;  // (x < 0).all() or (x > 0).all()
        const FsVec4 p = E.pos[best_i[0] < E.n ? best_i[0] : 0];
        nearest_out[3 * slot] = p.x; nearest_out[3 * slot + 1] = p.y; nearest_out[3 * slot + 2] = p.z;
    }
}

extern "C" int fs_stretch_probe(fs_ctx *ctx, int n, const int *envs, const float *midpoint_xz, const float *height_thr,
                                int *single_grasp_out, float *nearest_out) {
    if (!midpoint_xz || !height_thr || !single_grasp_out || !nearest_out) {
        fs_set_error("fs_stretch_probe: null argument");
        return FS_ERR_ARG;
    }
    int *d_ids = nullptr;
    int rc = upload_list(ctx, n, envs, &d_ids);
    if (rc != FS_OK) return rc;
    float *d_buf = (float *)fs_svc_scratch(ctx, sizeof(float) * 7 * n);  // mid[2n] | thr[n] | nearest[3n] | single[n] (ints)
    if (!d_buf) return FS_ERR_HIP;
    float *d_mid = d_buf, *d_thr = d_buf + 2 * n, *d_near = d_buf + 3 * n;
    int *d_single = (int *)(d_buf + 6 * n);
    hipError_t herr = hipMemcpyAsync(d_mid, midpoint_xz, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream);
    if (herr == hipSuccess) herr = hipMemcpyAsync(d_thr, height_thr, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(ctx->stream);  // the sources are pageable
    if (herr == hipSuccess) {
        hipLaunchKernelGGL(fs_k_stretch_probe, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->d_envs, d_ids, d_mid, d_thr,
                           d_single, d_near);
        herr = hipMemcpyAsync(nearest_out, d_near, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (herr == hipSuccess) herr = hipMemcpyAsync(single_grasp_out, d_single, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(ctx->stream);
    return fs_hip_ok(herr, "fs_stretch_probe") ? FS_OK : FS_ERR_HIP;
}

// ---- SimEnv.preaction / postaction (environment/simEnv.py:464-475): "did the action move the cloth at all?"
// preaction keeps the positions; postaction takes max_i || |post_i - pre_i| ||_2 (float32, like the numpy expression
// np.linalg.norm(np.abs(post - pre), axis=1).max()) and ends the episode when it is below 5e-2.
__global__ __launch_bounds__(256) void fs_k_max_displacement(const FsEnvDev *envs, const int *ids, const FsVec4 *const *snap,
                                                             float *out) {
    __shared__ float red[256];
    const FsEnvDev &E = envs[ids[blockIdx.x]];
    const FsVec4 *pre = snap[blockIdx.x];
    float m = 0.0f;
    for (int i = threadIdx.x; i < E.n; i += blockDim.x) {
        const FsVec4 a = E.pos[i], b = pre[i];
        const float dx = fabsf(a.x - b.x), dy = fabsf(a.y - b.y), dz = fabsf(a.z - b.z);
        m = fs_nanmax(m, sqrtf(dx * dx + dy * dy + dz * dz));
    }
    m = fs_block_max(m, red);
    if (threadIdx.x == 0) out[blockIdx.x] = m;
}

extern "C" int fs_snapshot_positions(fs_ctx *ctx, int n, const int *envs) {
    if (!ctx || !envs || n <= 0) { fs_set_error("fs_snapshot_positions: bad arguments"); return FS_ERR_ARG; }
    for (int k = 0; k < n; ++k) {
        if (envs[k] < 0 || envs[k] >= ctx->n_envs || !ctx->envs[envs[k]].has_scene) {
            fs_set_error("fs_snapshot_positions: bad episode id / no scene");
            return FS_ERR_ARG;
        }
        FsEnv &e = ctx->envs[envs[k]];
        const size_t bytes = sizeof(FsVec4) * (size_t)e.host.n;
        if (e.d_snapshot && e.snapshot_cap < e.host.n) {  // grow-only (hipFree synchronises the device)
            fs_sync_all_streams(ctx);
            (void)hipFree(e.d_snapshot);
            e.d_snapshot = nullptr;
        }
        if (!e.d_snapshot) {
            const int cap = e.host.n < 16384 ? 16384 : e.host.n;  // room for every cloth of the reference's task sizes
            HIP_TRY(hipMalloc((void **)&e.d_snapshot, sizeof(FsVec4) * (size_t)cap));
            e.snapshot_cap = cap;
        }
        e.snapshot_n = e.host.n;
        HIP_TRY(hipMemcpyAsync(e.d_snapshot, e.dev.pos, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return FS_OK;
}

extern "C" int fs_max_displacement(fs_ctx *ctx, int n, const int *envs, float *out, int n_floats) {
    if (!out || n_floats < n) { fs_set_error("fs_max_displacement: output too small"); return FS_ERR_ARG; }
    int *d_ids = nullptr;
    int rc = upload_list(ctx, n, envs, &d_ids);
    if (rc != FS_OK) return rc;
    std::vector<const FsVec4 *> h_snap(n);
    for (int k = 0; k < n; ++k) {
        const FsEnv &e = ctx->envs[envs[k]];
        if (!e.d_snapshot || e.snapshot_n != e.host.n) { fs_set_error("fs_max_displacement: call fs_snapshot_positions first"); return FS_ERR_STATE; }
        h_snap[k] = e.d_snapshot;
    }
    char *d_buf = (char *)fs_svc_scratch(ctx, (sizeof(void *) + sizeof(float)) * n);  // pointers[n] | out[n]
    if (!d_buf) return FS_ERR_HIP;
    const FsVec4 **d_snap = (const FsVec4 **)d_buf;
    float *d_out = (float *)(d_buf + sizeof(void *) * n);
    hipError_t herr = hipMemcpyAsync(d_snap, h_snap.data(), sizeof(void *) * n, hipMemcpyHostToDevice, ctx->stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(ctx->stream);  // pageable source
    if (herr == hipSuccess) {
        hipLaunchKernelGGL(fs_k_max_displacement, dim3((unsigned)n), dim3(256), 0, ctx->stream, ctx->d_envs, d_ids, d_snap, d_out);
        herr = hipMemcpyAsync(out, d_out, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (herr == hipSuccess) herr = hipStreamSynchronize(ctx->stream);
    return fs_hip_ok(herr, "fs_max_displacement") ? FS_OK : FS_ERR_HIP;
}

// ---- one particle per episode set from the host without moving whole arrays: the task generator's pinned pick point
// (environment/tasks.py:177-224 rewrites all positions and velocities through pyflex every step to move ONE particle).
__global__ void fs_k_set_particles(const FsEnvDev *envs, const int *ids, const int *pids, const float *pos4, int zero_vel, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const FsEnvDev &E = envs[ids[k]];
    const int pid = pids[k];
    if (pid < 0 || pid >= E.n) return;
    E.pos[pid] = FsVec4{pos4[4 * k], pos4[4 * k + 1], pos4[4 * k + 2], pos4[4 * k + 3]};
    if (zero_vel) E.vel[pid] = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
}

extern "C" int fs_set_particles(fs_ctx *ctx, int n, const int *envs, const int *particle_ids, const float *pos4,
                                int zero_velocity) {
    if (!particle_ids || !pos4) { fs_set_error("fs_set_particles: null argument"); return FS_ERR_ARG; }
    int *d_ids = nullptr;
    int rc = upload_list(ctx, n, envs, &d_ids);
    if (rc != FS_OK) return rc;
    for (int k = 0; k < n; ++k) {
        if (particle_ids[k] < 0 || particle_ids[k] >= ctx->envs[envs[k]].host.n) {
            fs_set_error("fs_set_particles: particle id out of range");
            return FS_ERR_ARG;
        }
        if (const int guard_rc = fs_lane_guard(ctx, envs[k])) return guard_rc;
    }
    char *d_buf = (char *)fs_svc_scratch(ctx, (sizeof(int) + 4 * sizeof(float)) * n);  // pids[n] | pos4[4n]
    if (!d_buf) return FS_ERR_HIP;
    int *d_pids = (int *)d_buf;
    float *d_pos = (float *)(d_buf + sizeof(int) * n);
    hipError_t herr = hipMemcpyAsync(d_pids, particle_ids, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream);
    if (herr == hipSuccess) herr = hipMemcpyAsync(d_pos, pos4, sizeof(float) * 4 * n, hipMemcpyHostToDevice, ctx->stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(ctx->stream);  // pageable sources
    if (herr == hipSuccess) {
        hipLaunchKernelGGL(fs_k_set_particles, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_envs, d_ids, d_pids, d_pos,
                           zero_velocity, n);
        herr = hipStreamSynchronize(ctx->stream);
    }
    return fs_hip_ok(herr, "fs_set_particles") ? FS_OK : FS_ERR_HIP;
}
