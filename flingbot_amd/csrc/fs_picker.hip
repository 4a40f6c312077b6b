// fs_picker.hip -- on-device picker and `movep` trajectory executor.
//
// Native counterpart of the reference's host loop around the solver:
//     SimEnv.movep                  environment/simEnv.py:739-769      (move pickers toward targets by `speed` per step)
//     PickerPickPlace.step          environment/flex_utils.py:223-252  (delta_move 1.0, steps_limit 1)
//     Picker.step / _set_pos        environment/flex_utils.py:114-205  (grasp nearest particle, pin + teleport it)
// The reference pays 5-7 synchronising pyflex getters/setters plus Python per simulation step there.  The picker motion
// is kinematic, so the HOST can compute the whole trajectory of a movep call up front (float64, rounded to float32
// exactly where the numpy code rounds); the DEVICE then runs, per simulation step, one small picker kernel (un-pick,
// nearest-particle grasp search, teleport of held particles, shape prev/current update) followed by the solver step,
// without any host round trip.  Episodes of a batch finish after different step counts: per-step id lists select the
// ones still moving.
#include <hip/hip_runtime.h>

#include <chrono>

#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/flingsim.h"
#include "fs_context.h"

#define HIP_TRY(call)                                     \
    do {                                                  \
        if (!fs_hip_ok((call), #call)) return FS_ERR_HIP; \
    } while (0)

struct FsPickerCmd {  // one Picker.step for one episode
    float new_pos[FS_MAX_SHAPES][4];  // float32 picker centres after the move (w unused)
    int grasp[FS_MAX_SHAPES];
};

// One workgroup per episode.  Mirrors Picker.step: every read of particle data refers to the arrays as they were at
// entry (`particle_pos`), writes go to the updated copy (`new_particle_pos`) -- in place here, which is equivalent
// because a particle is held by at most one picker.
__global__ __launch_bounds__(256) void fs_k_picker_step(const FsEnvDev *envs, FsShapesDev *shapes, const int *ids,
                                                        const FsPickerCmd *cmds, int *const *picked_ptrs,
                                                        float *const *saved_w_ptrs, double threshold) {
    const int slot = blockIdx.x;
    const int e = ids[slot];
    const FsEnvDev &E = envs[e];
    FsShapesDev &sh = shapes[e];
    const FsPickerCmd &cmd = cmds[slot];
    int *picked = picked_ptrs[e];
    const float *saved_w = saved_w_ptrs[e];
    const int t = threadIdx.x, n = E.n, S = sh.count;
    __shared__ double best_d[256];
    __shared__ int best_i[256];
    __shared__ int s_picked[FS_MAX_SHAPES];
    if (t < S) s_picked[t] = picked[t];
    __syncthreads();
    // 1. un-pick: restore the inverse mass of released particles (flex_utils.py:134-140)
    if (t == 0) {
        for (int k = 0; k < S; ++k)
            if (!cmd.grasp[k] && s_picked[k] >= 0) {
                E.pos[s_picked[k]].w = saved_w[s_picked[k]];
                s_picked[k] = -1;
            }
    }
    __syncthreads();
    // 2. pick + teleport, picker by picker (flex_utils.py:143-173)
    for (int k = 0; k < S; ++k) {
        if (cmd.grasp[k] && s_picked[k] < 0) {
            // nearest particle within threshold of the picker BEFORE the move, not already held; ties -> lowest index
            const double px = (double)sh.pos[k].x, py = (double)sh.pos[k].y, pz = (double)sh.pos[k].z;
            double bd = 1.0e300;
            int bi = -1;
            for (int i = t; i < n; i += 256) {
                const FsVec4 p = E.pos[i];
                const double dx = (double)p.x - px, dy = (double)p.y - py, dz = (double)p.z - pz;
                const double d = sqrt(dx * dx + dy * dy + dz * dz);
                bool held = false;
                for (int q = 0; q < S; ++q) held = held || (s_picked[q] == i);
                if (d <= threshold && !held && (d < bd || (d == bd && i < bi))) { bd = d; bi = i; }
            }
            best_d[t] = bd;
            best_i[t] = bi;
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
                if (t < off) {
                    const double od = best_d[t + off];
                    const int oi = best_i[t + off];
                    if (oi >= 0 && (best_i[t] < 0 || od < best_d[t] || (od == best_d[t] && oi < best_i[t]))) {
                        best_d[t] = od;
                        best_i[t] = oi;
                    }
                }
                __syncthreads();
            }
            if (t == 0 && best_i[0] >= 0) s_picked[k] = best_i[0];
            __syncthreads();
        }
        if (t == 0 && cmd.grasp[k] && s_picked[k] >= 0) {
            FsVec4 p = E.pos[s_picked[k]];
            // float32: particle + new_picker - picker, left to right (flex_utils.py:168-170)
            p.x = (p.x + cmd.new_pos[k][0]) - sh.pos[k].x;
            p.y = (p.y + cmd.new_pos[k][1]) - sh.pos[k].y;
            p.z = (p.z + cmd.new_pos[k][2]) - sh.pos[k].z;
            p.w = 0.0f;  // "set the mass to infinity"
            E.pos[s_picked[k]] = p;
        }
        __syncthreads();
    }
    // 3. _set_pos: prev := current, current := new (flex_utils.py:114-119)
    if (t < S) {
        const float r = sh.pos[t].w;
        sh.prev[t] = FsVec4{sh.pos[t].x, sh.pos[t].y, sh.pos[t].z, r};
        sh.pos[t] = FsVec4{cmd.new_pos[t][0], cmd.new_pos[t][1], cmd.new_pos[t][2], r};
        picked[t] = s_picked[t];
    }
}

__global__ void fs_k_save_inv_mass(const FsVec4 *pos, float *saved_w, int n, int *picked) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) saved_w[i] = pos[i].w;
    if (i < FS_MAX_SHAPES) picked[i] = -1;
}

static FsEnv *picker_env(fs_ctx *ctx, int env) {
    if (!ctx || env < 0 || env >= ctx->n_envs || !ctx->envs[env].has_scene) {
        fs_set_error("picker: bad env or no scene");
        return nullptr;
    }
    return &ctx->envs[env];
}

// Picker.reset's bookkeeping (flex_utils.py:85,99-101): no particle held, remember every particle's inverse mass.
extern "C" int fs_picker_reset(fs_ctx *ctx, int env, double picker_threshold, double particle_radius) {
    FsEnv *e = picker_env(ctx, env);
    if (!e) return FS_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!e->d_picked) HIP_TRY(hipMalloc((void **)&e->d_picked, sizeof(int) * FS_MAX_SHAPES));
    if (!e->d_saved_w) HIP_TRY(hipMalloc((void **)&e->d_saved_w, sizeof(float) * e->host.n));
    e->picker_threshold = picker_threshold;
    e->particle_radius = particle_radius;
    e->picker_radius = -1.0;  // until fs_picker_set_radius says otherwise: the (float32) radius of shape 0
    hipLaunchKernelGGL(fs_k_save_inv_mass, dim3((e->host.n + 255) / 256), dim3(256), 0, ctx->stream, e->dev.pos,
                       e->d_saved_w, e->host.n, e->d_picked);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    e->picker_ready = true;
    return FS_OK;
}

// simulation steps (summed over the episodes) the most recent fs_movep* call executed: a movep iteration that finds the
// pickers on their targets (min_steps) takes none (flex_utils.py:231-233), so this is <= the sum of the iteration counts
extern "C" long long fs_last_movep_steps(const fs_ctx *ctx) { return ctx ? ctx->last_movep_steps : -1; }

// Picker.picker_radius (flex_utils.py:57, used in the grasp threshold :154-155) is a python float: the double 0.02, not
// the float32 radius pyflex.add_sphere stored.  Callers that know it pass it here so the threshold sum is the reference's.
extern "C" int fs_picker_set_radius(fs_ctx *ctx, int env, double picker_radius) {
    FsEnv *e = picker_env(ctx, env);
    if (!e) return FS_ERR_ARG;
    e->picker_radius = picker_radius;
    return FS_OK;
}

static double picker_grasp_threshold(const FsEnv &e) {
    const double r = e.picker_radius >= 0.0 ? e.picker_radius : (double)e.shapes.pos[0].w;
    return e.picker_threshold + r + e.particle_radius;  // summed left to right like flex_utils.py:154-155
}

extern "C" int fs_picker_get_picked(fs_ctx *ctx, int env, int *out, int n_ints) {
    FsEnv *e = picker_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    if (!e->picker_ready) { fs_set_error("fs_picker_reset has not been called"); return FS_ERR_STATE; }
    if (n_ints < e->shapes.count) { fs_set_error("buffer too small"); return FS_ERR_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    int tmp[FS_MAX_SHAPES];
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(tmp, e->d_picked, sizeof(tmp), hipMemcpyDeviceToHost));
    for (int k = 0; k < e->shapes.count; ++k) out[k] = tmp[k];
    return FS_OK;
}

namespace {
struct Plan {  // host-side trajectory of one episode for one movep call
    std::vector<FsPickerCmd> cmds;  // one per SIMULATION step
    int iterations = 0;             // movep loop iterations (>= cmds.size(): iterations on the target take no step)
    bool limit_hit = false;
    bool capped = false;            // stopped because max_cmds simulation steps are planned: resume at `iterations`
};

inline double norm3(const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// simEnv.py:739-769 + flex_utils.py:223-252 on the kinematic picker state only (float64 math, float32 state).
// f32_targets: the caller's targets are a float32 numpy array in the reference (stretch_cloth builds them from the
// float32 picker positions, simEnv.py:146-156,180-182), so movep's own arithmetic -- delta, its norm, the step toward the
// target -- happens in float32 there; PickerPickPlace.step below always works in float64.
// start_step / max_cmds: resume the loop at iteration `start_step` (movep is stateless apart from its loop index: every
// iteration starts from the pickers' current positions) and stop once max_cmds simulation steps are planned (< 0: no cap).
Plan plan_movep(const FsShapesDev &shapes, const double *targets, const int *grasp, double speed, int limit,
                int min_steps, double eps, bool f32_targets, int start_step = 0, int max_cmds = -1) {
    Plan plan;
    const int S = shapes.count;
    float cur[FS_MAX_SHAPES][3];
    for (int k = 0; k < S; ++k) { cur[k][0] = shapes.pos[k].x; cur[k][1] = shapes.pos[k].y; cur[k][2] = shapes.pos[k].z; }
    for (int step = start_step; step < limit; ++step) {
        double end[FS_MAX_SHAPES][3];
        bool all_close = true;
        for (int k = 0; k < S; ++k) {
            if (f32_targets) {
                float d[3];
                for (int c = 0; c < 3; ++c) d[c] = (float)targets[3 * k + c] - cur[k][c];
                const float dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                if (!((double)dist < eps)) all_close = false;
                for (int c = 0; c < 3; ++c)
                    end[k][c] = (double)dist < speed ? (double)(float)targets[3 * k + c]
                                                     : (double)(cur[k][c] + (d[c] / dist) * (float)speed);
                continue;
            }
            double d[3];
            for (int c = 0; c < 3; ++c) d[c] = targets[3 * k + c] - (double)cur[k][c];
            const double dist = norm3(d);
            if (!(dist < eps)) all_close = false;
            for (int c = 0; c < 3; ++c)
                end[k][c] = dist < speed ? targets[3 * k + c] : (double)cur[k][c] + (d[c] / dist) * speed;
        }
        if (all_close && (min_steps < 0 || step > min_steps)) {
            plan.iterations = step;
            return plan;
        }
        // PickerPickPlace.step
        double num_step = 0.0;
        for (int k = 0; k < S; ++k) {
            double d[3];
            for (int c = 0; c < 3; ++c) d[c] = (double)cur[k][c] - end[k][c];
            const double ns = ceil(norm3(d) / 1.0);
            if (ns > num_step) num_step = ns;
        }
        if (num_step < 0.1) continue;  // already on the targets: the reference returns without stepping the simulation
        if (max_cmds >= 0 && (int)plan.cmds.size() >= max_cmds) {  // this call's budget of simulation steps is planned
            plan.iterations = step;
            plan.capped = true;
            return plan;
        }
        double delta[FS_MAX_SHAPES][3], sq = 0.0;
        for (int k = 0; k < S; ++k)
            for (int c = 0; c < 3; ++c) {
                delta[k][c] = (end[k][c] - (double)cur[k][c]) / num_step;
                sq += delta[k][c] * delta[k][c];
            }
        const double norm_delta = sqrt(sq);
        bool all_within = true;
        for (int k = 0; k < S; ++k) {
            double d[3];
            for (int c = 0; c < 3; ++c) d[c] = end[k][c] - (double)cur[k][c];
            if (!(norm3(d) < norm_delta)) all_within = false;
        }
        if (all_within)
            for (int k = 0; k < S; ++k)
                for (int c = 0; c < 3; ++c) delta[k][c] = end[k][c] - (double)cur[k][c];
        FsPickerCmd cmd;
        memset(&cmd, 0, sizeof(cmd));
        for (int k = 0; k < S; ++k) {
            for (int c = 0; c < 3; ++c) {
                cur[k][c] = (float)((double)cur[k][c] + delta[k][c]);  // Picker.step: float32 <- float32 + float64
                cmd.new_pos[k][c] = cur[k][c];
            }
            cmd.grasp[k] = grasp[k] ? 1 : 0;
        }
        plan.cmds.push_back(cmd);
    }
    plan.limit_hit = true;
    plan.iterations = limit;
    return plan;
}
}  // namespace

// movep for a batch of episodes: targets double[n][S][3], grasp int[n][S] (S = shapes of the episode, identical
// for every episode of the batch), iterations_out int[n].  Returns FS_ERR_LIMIT if any episode ran into `limit`
// (MoveJointsException in the reference); the trajectories are executed up to the limit in that case.
static int movep_batch_impl(fs_ctx *ctx, int n, const int *envs, const double *targets, const int *grasp, double speed,
                            int limit, int min_steps, double eps, int *iterations_out, bool f32_targets) {
    if (ctx) ctx->last_movep_steps = 0;  // an early error return must not leave the previous call's count behind
    if (!ctx || n <= 0 || !envs || !targets || !grasp) { fs_set_error("fs_movep: bad arguments"); return FS_ERR_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    int S = -1;
    std::vector<Plan> plans(n);
    size_t max_steps = 0;
    for (int a = 0; a < n; ++a) {
        FsEnv *e = picker_env(ctx, envs[a]);
        if (!e) return FS_ERR_ARG;
        if (!e->picker_ready) { fs_set_error("fs_movep: call fs_picker_reset first"); return FS_ERR_STATE; }
        if (S < 0) S = e->shapes.count;
        if (e->shapes.count != S || S <= 0) { fs_set_error("fs_movep: episodes need the same (non-zero) picker count"); return FS_ERR_STATE; }
        if (picker_grasp_threshold(*e) != picker_grasp_threshold(ctx->envs[envs[0]])) {
            fs_set_error("fs_movep: episodes moved together need the same grasp threshold (picker / particle radius)");
            return FS_ERR_STATE;
        }
        plans[a] = plan_movep(e->shapes, targets + (size_t)a * S * 3, grasp + (size_t)a * S, speed, limit, min_steps, eps,
                              f32_targets);
        if (iterations_out) iterations_out[a] = plans[a].iterations;
        if (plans[a].cmds.size() > max_steps) max_steps = plans[a].cmds.size();
    }
    bool any_limit = false;
    for (auto &p : plans) any_limit = any_limit || p.limit_hit;
    ctx->last_movep_steps = 0;
    for (auto &p : plans) ctx->last_movep_steps += (long long)p.cmds.size();
    if (max_steps > 0) {
        // per step: the ids of the episodes still moving + their commands, all uploaded once
        std::vector<int> h_ids(max_steps * n, 0), h_count(max_steps, 0);
        std::vector<FsPickerCmd> h_cmds(max_steps * n);
        for (size_t s = 0; s < max_steps; ++s)
            for (int a = 0; a < n; ++a)
                if (s < plans[a].cmds.size()) {
                    const int slot = h_count[s]++;
                    h_ids[s * n + slot] = envs[a];
                    h_cmds[s * n + slot] = plans[a].cmds[s];
                }
        std::vector<int *> h_picked(ctx->n_envs, nullptr);
        std::vector<float *> h_saved(ctx->n_envs, nullptr);
        for (int i = 0; i < ctx->n_envs; ++i) { h_picked[i] = ctx->envs[i].d_picked; h_saved[i] = ctx->envs[i].d_saved_w; }
        struct DevBufs {  // freed on every exit path
            int *ids = nullptr;
            FsPickerCmd *cmds = nullptr;
            int **picked = nullptr;
            float **saved = nullptr;
            ~DevBufs() { (void)hipFree(ids); (void)hipFree(cmds); (void)hipFree(picked); (void)hipFree(saved); }
        } bufs;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMalloc((void **)&bufs.ids, sizeof(int) * h_ids.size()));
        HIP_TRY(hipMalloc((void **)&bufs.cmds, sizeof(FsPickerCmd) * h_cmds.size()));
        HIP_TRY(hipMalloc((void **)&bufs.picked, sizeof(int *) * ctx->n_envs));
        HIP_TRY(hipMalloc((void **)&bufs.saved, sizeof(float *) * ctx->n_envs));
        int *d_ids = bufs.ids;
        FsPickerCmd *d_cmds = bufs.cmds;
        int **d_picked = bufs.picked;
        float **d_saved = bufs.saved;
        HIP_TRY(hipMemcpy(d_ids, h_ids.data(), sizeof(int) * h_ids.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_cmds, h_cmds.data(), sizeof(FsPickerCmd) * h_cmds.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_picked, h_picked.data(), sizeof(int *) * ctx->n_envs, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_saved, h_saved.data(), sizeof(float *) * ctx->n_envs, hipMemcpyHostToDevice));
        int rc = FS_OK;
        for (size_t s = 0; s < max_steps && rc == FS_OK; ++s) {
            const int cnt = h_count[s];
            std::vector<int> ids(h_ids.begin() + s * n, h_ids.begin() + s * n + cnt);
            const double thr = picker_grasp_threshold(ctx->envs[ids[0]]);
            hipLaunchKernelGGL(fs_k_picker_step, dim3(cnt), dim3(256), 0, ctx->stream, ctx->d_envs, ctx->d_shapes,
                               d_ids + s * n, d_cmds + s * n, d_picked, d_saved, thr);
            rc = fs_step_ids(ctx, ids, 1, d_ids + s * n);
        }
        hipError_t err = hipStreamSynchronize(ctx->stream);
        if (rc != FS_OK) return rc;
        HIP_TRY(err);
        // host mirrors of the shape states follow the planned trajectory
        for (int a = 0; a < n; ++a) {
            FsEnv &e = ctx->envs[envs[a]];
            const auto &cm = plans[a].cmds;
            if (cm.empty()) continue;
            for (int k = 0; k < S; ++k) {
                const float r = e.shapes.pos[k].w;
                const float *pv = cm.size() >= 2 ? cm[cm.size() - 2].new_pos[k] : &e.shapes.pos[k].x;
                e.shapes.prev[k] = FsVec4{pv[0], pv[1], pv[2], r};
                e.shapes.pos[k] = FsVec4{cm.back().new_pos[k][0], cm.back().new_pos[k][1], cm.back().new_pos[k][2], r};
            }
        }
    }
    if (any_limit) { fs_set_error("fs_movep: step limit reached (MoveJointsException)"); return FS_ERR_LIMIT; }
    return FS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// fs_advance: one CHUNK of simulation for episodes that are in DIFFERENT phases of their primitives -- some inside a movep
// (simEnv.py:739-769), some inside wait_until_stable (flex_utils.py:430-441) -- so that all of them share every launch
// sequence.  The batched primitives of flingbot_amd/primitives.py advance the episodes phase by phase in lock step and the
// whole batch waits for the slowest episode of every phase (measured: 15.6 of 32 episodes active per launch sequence in the
// evaluation loop); the scheduler of flingbot_amd/schedule.py instead gives every episode its own state machine and calls
// this function with whatever each episode needs next, at most `cap` simulation steps per call.  Both loops are resumable
// without changing a bit: a movep iteration only reads the pickers' current positions and its own loop index (start[]),
// wait_until_stable only counts its steps.
// One device id list per launch sequence: [waiters | movers still moving at this step]; a check kernel retires a waiter
// (id -> -1) when it is stable or its step budget is used up, the picker kernel moves the movers' pickers.
__global__ __launch_bounds__(256) void fs_k_wait_check(const FsEnvDev *envs, int *row, const double *tols, const int *budget,
                                                       int *steps, int *stable, int *dead) {
    __shared__ float red[256];
    const int slot = blockIdx.x;
    if (dead[slot]) {  // retired in an earlier launch sequence: this sequence's list entry goes too
        if (threadIdx.x == 0) row[slot] = -1;
        return;
    }
    const int e = row[slot];
    if (steps[slot] >= budget[slot]) {  // this call's (or the loop's) steps are used up: not stable, no test (flex_utils.py:441)
        if (threadIdx.x == 0) { row[slot] = -1; dead[slot] = 1; }
        return;
    }
    const FsEnvDev &E = envs[e];
    float m = 0.0f;
    for (int i = threadIdx.x; i < E.n; i += 256) {
        const FsVec4 v = E.vel[i];
        const float a = fabsf(v.x), b = fabsf(v.y), c = fabsf(v.z);
        const float q = (a != a || b != b || c != c) ? __int_as_float(0x7fc00000) : fmaxf(a, fmaxf(b, c));
        m = (m != m || q != q) ? __int_as_float(0x7fc00000) : fmaxf(m, q);  // numpy's max propagates NaN
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float x = red[threadIdx.x], y = red[threadIdx.x + s];
            red[threadIdx.x] = (x != x || y != y) ? __int_as_float(0x7fc00000) : fmaxf(x, y);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if ((double)red[0] < tols[slot]) { stable[slot] = 1; dead[slot] = 1; row[slot] = -1; }  // (plain steps: tolerance -1, never true)
        else steps[slot] += 1;  // the step that follows
    }
}

extern "C" int fs_advance(fs_ctx *ctx, int n, const int *envs, const int *kind, const double *targets, const int *grasp,
                          const double *speed, const int *limit, const int *min_steps, const int *f32, const int *start,
                          double eps, const double *tolerance, int cap_min, int cap, int *progress_out, int *status_out,
                          int *steps_out) {
    if (!ctx || n <= 0 || n > ctx->n_envs || !envs || !kind || !limit || !start || !progress_out || !status_out || !steps_out ||
        cap <= 0 || cap_min <= 0 || cap_min > cap) {
        fs_set_error("fs_advance: bad arguments");
        return FS_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const auto wall0 = std::chrono::steady_clock::now();
    ctx->last_movep_steps = 0;
    std::vector<int> movers, waiters;
    std::vector<char> listed((size_t)ctx->n_envs, 0);
    int S = -1;
    for (int a = 0; a < n; ++a) {
        FsEnv *e = picker_env(ctx, envs[a]);
        if (!e) return FS_ERR_ARG;
        if (listed[envs[a]]) { fs_set_error("fs_advance: an episode is listed twice"); return FS_ERR_ARG; }
        listed[envs[a]] = 1;
        if (kind[a] == 0) {
            if (!targets || !grasp || !speed || !min_steps || !f32) { fs_set_error("fs_advance: movep arguments missing"); return FS_ERR_ARG; }
            if (!e->picker_ready) { fs_set_error("fs_advance: call fs_picker_reset first"); return FS_ERR_STATE; }
            if (S < 0) S = e->shapes.count;
            if (e->shapes.count != S || S <= 0) { fs_set_error("fs_advance: episodes need the same (non-zero) picker count"); return FS_ERR_STATE; }
            if (!movers.empty() && picker_grasp_threshold(*e) != picker_grasp_threshold(ctx->envs[envs[movers[0]]])) {
                fs_set_error("fs_advance: episodes moved together need the same grasp threshold");
                return FS_ERR_STATE;
            }
            movers.push_back(a);
        } else if (kind[a] == 1 || kind[a] == 2) {
            if (kind[a] == 1 && !tolerance) { fs_set_error("fs_advance: tolerance missing"); return FS_ERR_ARG; }
            waiters.push_back(a);
        } else {
            fs_set_error("fs_advance: kind must be 0 (movep), 1 (wait_until_stable) or 2 (plain steps)");
            return FS_ERR_ARG;
        }
    }
    const int nm = (int)movers.size(), nw = (int)waiters.size();
    // movers: plan this call's part of each trajectory.  The chunk ends when the FIRST mover finishes its movep (so that its
    // program can issue the next request without idling through the others' steps), but not before cap_min steps (the host
    // round trip per call) and not after cap.
    std::vector<Plan> plans(nm);
    size_t n_seq = 0, shortest = (size_t)cap;
    for (int q = 0; q < nm; ++q) {
        const int a = movers[q];
        plans[q] = plan_movep(ctx->envs[envs[a]].shapes, targets + (size_t)a * S * 3, grasp + (size_t)a * S, speed[a], limit[a],
                              min_steps[a], eps, f32[a] != 0, start[a], cap);
        if (plans[q].cmds.size() < shortest) shortest = plans[q].cmds.size();
    }
    const int chunk = nm == 0 ? cap : (int)(shortest < (size_t)cap_min ? (size_t)cap_min : shortest);
    for (int q = 0; q < nm; ++q) {
        const int a = movers[q];
        if ((int)plans[q].cmds.size() > chunk)  // same trajectory, cut at the chunk's end
            plans[q] = plan_movep(ctx->envs[envs[a]].shapes, targets + (size_t)a * S * 3, grasp + (size_t)a * S, speed[a], limit[a],
                                  min_steps[a], eps, f32[a] != 0, start[a], chunk);
        progress_out[a] = plans[q].iterations;
        status_out[a] = plans[q].capped ? 0 : (plans[q].limit_hit ? 2 : 1);
        steps_out[a] = (int)plans[q].cmds.size();
        if (plans[q].cmds.size() > n_seq) n_seq = plans[q].cmds.size();
    }
    const size_t mover_seq = n_seq;  // launch sequences that still have a mover
    std::vector<int> w_budget(nw, 0);
    for (int q = 0; q < nw; ++q) {
        const int a = waiters[q];
        const int left = limit[a] - start[a];
        w_budget[q] = left < 0 ? 0 : (left < chunk ? left : chunk);
        if ((size_t)w_budget[q] > n_seq) n_seq = (size_t)w_budget[q];
        if (w_budget[q] == 0) {  // budget of the whole loop already used: wait_until_stable returns False
            progress_out[a] = start[a]; status_out[a] = kind[a] == 2 ? 1 : 2; steps_out[a] = 0;
        }
    }
    if (n_seq == 0) return FS_OK;
    // device tables, ONE upload into the context's scratch.  Per launch sequence s a row of the launch list:
    // [waiters | movers still moving at s | -1 ...]; the check kernel retires a waiter from its row and, through dead[], from
    // every later row.
    const int width = nm > 0 ? nm : 1, W = nw + width, n_envs = ctx->n_envs;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) & ~size_t(15); return o; };
    const size_t o_picked = carve(sizeof(int *) * n_envs), o_saved = carve(sizeof(float *) * n_envs);
    const size_t o_tol = carve(sizeof(double) * (nw > 0 ? nw : 1));
    const size_t o_cmds = carve(sizeof(FsPickerCmd) * n_seq * width);
    const size_t o_rows = carve(sizeof(int) * n_seq * W);
    const size_t o_wait = carve(sizeof(int) * 4 * (nw > 0 ? nw : 1));  // budget | steps | stable | dead
    std::vector<char> blob(off, 0);
    int **h_picked = (int **)(blob.data() + o_picked);
    float **h_saved = (float **)(blob.data() + o_saved);
    double *h_tol = (double *)(blob.data() + o_tol);
    FsPickerCmd *h_cmds = (FsPickerCmd *)(blob.data() + o_cmds);
    int *h_rows = (int *)(blob.data() + o_rows), *h_wait = (int *)(blob.data() + o_wait);
    std::vector<int> h_cnt(n_seq, 0);
    for (int i = 0; i < n_envs; ++i) { h_picked[i] = ctx->envs[i].d_picked; h_saved[i] = ctx->envs[i].d_saved_w; }
    for (int q = 0; q < nw; ++q) {
        h_tol[q] = kind[waiters[q]] == 1 ? tolerance[waiters[q]] : -1.0;
        h_wait[q] = w_budget[q];
    }
    for (size_t s = 0; s < n_seq; ++s) {
        int *row = h_rows + s * W;
        for (int k = 0; k < W; ++k) row[k] = -1;
        for (int q = 0; q < nw; ++q) row[q] = envs[waiters[q]];
        for (int q = 0; q < nm; ++q)
            if (s < plans[q].cmds.size()) {
                const int slot = h_cnt[s]++;
                row[nw + slot] = envs[movers[q]];
                h_cmds[s * width + slot] = plans[q].cmds[s];
            }
    }
    char *dev = (char *)fs_loop_scratch(ctx, off);
    if (!dev) return FS_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(dev, blob.data(), off, hipMemcpyHostToDevice));
    if (!ctx->adv_ev0) {
        HIP_TRY(hipEventCreate(&ctx->adv_ev0));
        HIP_TRY(hipEventCreate(&ctx->adv_ev1));
    }
    HIP_TRY(hipEventRecord(ctx->adv_ev0, ctx->stream));
    const auto wall1 = std::chrono::steady_clock::now();
    int **d_picked = (int **)(dev + o_picked);
    float **d_saved = (float **)(dev + o_saved);
    const double *d_tol = (const double *)(dev + o_tol);
    const FsPickerCmd *d_cmds = (const FsPickerCmd *)(dev + o_cmds);
    int *d_rows = (int *)(dev + o_rows), *d_wait = (int *)(dev + o_wait);
    int *d_w_budget = d_wait, *d_w_steps = d_wait + nw, *d_w_stable = d_wait + 2 * nw, *d_w_dead = d_wait + 3 * nw;
    int rc = FS_OK;
    std::vector<int> ids;
    for (size_t s = 0; s < n_seq && rc == FS_OK; ++s) {
        const int cnt = h_cnt[s];
        int *d_row = d_rows + s * W;
        ids.assign(h_rows + s * W, h_rows + s * W + nw + cnt);
        if (cnt > 0) {
            const double thr = picker_grasp_threshold(ctx->envs[ids[nw]]);
            hipLaunchKernelGGL(fs_k_picker_step, dim3(cnt), dim3(256), 0, ctx->stream, ctx->d_envs, ctx->d_shapes, d_row + nw,
                               d_cmds + s * width, d_picked, d_saved, thr);
        }
        if (nw > 0)
            hipLaunchKernelGGL(fs_k_wait_check, dim3(nw), dim3(256), 0, ctx->stream, ctx->d_envs, d_row, d_tol, d_w_budget,
                               d_w_steps, d_w_stable, d_w_dead);
        rc = fs_step_ids(ctx, ids, 1, d_row);
        if (rc == FS_OK && nw > 0 && s >= mover_seq && (s & 15) == 15 && s + 1 < n_seq) {  // only waiters left: all retired?
            int *live = (int *)fs_stage(ctx, sizeof(int) * nw);
            hipError_t pe = live ? hipMemcpyAsync(live, d_w_dead, sizeof(int) * nw, hipMemcpyDeviceToHost, ctx->stream) : hipErrorOutOfMemory;
            if (pe == hipSuccess) pe = hipStreamSynchronize(ctx->stream);
            if (!fs_hip_ok(pe, "fs_advance poll")) { rc = FS_ERR_HIP; break; }
            bool any = false;
            for (int q = 0; q < nw; ++q) any = any || live[q] == 0;
            if (!any) break;
        }
    }
    std::vector<int> w_out(2 * (nw > 0 ? nw : 1), 0);
    hipError_t err = hipSuccess;
    if (rc == FS_OK && nw > 0)
        err = hipMemcpyAsync(w_out.data(), d_w_steps, sizeof(int) * 2 * nw, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipEventRecord(ctx->adv_ev1, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (rc != FS_OK) return rc;
    HIP_TRY(err);
    {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->adv_ev0, ctx->adv_ev1) == hipSuccess) ctx->adv_gpu_ms += ms;
        const auto wall2 = std::chrono::steady_clock::now();
        ctx->adv_wall_ms += std::chrono::duration<double, std::milli>(wall2 - wall0).count();
        ctx->adv_prep_ms += std::chrono::duration<double, std::milli>(wall1 - wall0).count();
        ctx->adv_calls += 1;
        ctx->adv_sequences += (long long)n_seq;
    }
    for (int q = 0; q < nm; ++q) ctx->last_movep_steps += (long long)plans[q].cmds.size();
    for (int q = 0; q < nw; ++q) {
        const int a = waiters[q];
        if (w_budget[q] == 0) continue;
        const int taken = w_out[q], stable = w_out[nw + q];
        progress_out[a] = start[a] + taken;
        steps_out[a] = taken;
        status_out[a] = stable ? 1 : (progress_out[a] >= limit[a] ? (kind[a] == 2 ? 1 : 2) : 0);
    }
    // host mirrors of the shape states follow the planned trajectories
    for (int q = 0; q < nm; ++q) {
        FsEnv &e = ctx->envs[envs[movers[q]]];
        const auto &cm = plans[q].cmds;
        if (cm.empty()) continue;
        for (int k = 0; k < S; ++k) {
            const float r = e.shapes.pos[k].w;
            const float *pv = cm.size() >= 2 ? cm[cm.size() - 2].new_pos[k] : &e.shapes.pos[k].x;
            e.shapes.prev[k] = FsVec4{pv[0], pv[1], pv[2], r};
            e.shapes.pos[k] = FsVec4{cm.back().new_pos[k][0], cm.back().new_pos[k][1], cm.back().new_pos[k][2], r};
        }
    }
    return FS_OK;
}

// fs_advance's stopwatch since the context was created: out[0] calls, [1] launch sequences, [2] wall ms inside the calls,
// [3] device ms between a call's first and last launch (the stream is idle when a call starts), [4] wall ms a call spends
// before its first launch (planning, tables, upload)
extern "C" int fs_advance_timing(const fs_ctx *ctx, double *out5) {
    if (!ctx || !out5) return FS_ERR_ARG;
    out5[0] = (double)ctx->adv_calls; out5[1] = (double)ctx->adv_sequences; out5[2] = ctx->adv_wall_ms;
    out5[3] = ctx->adv_gpu_ms; out5[4] = ctx->adv_prep_ms;
    return FS_OK;
}

extern "C" int fs_movep_batch(fs_ctx *ctx, int n, const int *envs, const double *targets, const int *grasp, double speed,
                              int limit, int min_steps, double eps, int *iterations_out) {
    return movep_batch_impl(ctx, n, envs, targets, grasp, speed, limit, min_steps, eps, iterations_out, false);
}

extern "C" int fs_movep_batch_f32(fs_ctx *ctx, int n, const int *envs, const float *targets, const int *grasp, double speed,
                                  int limit, int min_steps, double eps, int *iterations_out) {
    if (!ctx || n <= 0 || !envs || !targets) { fs_set_error("fs_movep: bad arguments"); return FS_ERR_ARG; }
    FsEnv *e0 = picker_env(ctx, envs[0]);
    if (!e0) return FS_ERR_ARG;
    const size_t count = (size_t)n * (size_t)e0->shapes.count * 3;
    std::vector<double> wide(count);
    for (size_t k = 0; k < count; ++k) wide[k] = (double)targets[k];
    return movep_batch_impl(ctx, n, envs, wide.data(), grasp, speed, limit, min_steps, eps, iterations_out, true);
}

extern "C" int fs_movep(fs_ctx *ctx, int env, const double *targets, const int *grasp, double speed, int limit,
                        int min_steps, double eps, int *iterations_out) {
    return fs_movep_batch(ctx, 1, &env, targets, grasp, speed, limit, min_steps, eps, iterations_out);
}
