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
    if (const int guard_rc = fs_lane_guard(ctx, env)) return guard_rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!e->d_picked) HIP_TRY(hipMalloc((void **)&e->d_picked, sizeof(int) * FS_MAX_SHAPES));
    if (!e->d_saved_w) {
        HIP_TRY(hipMalloc((void **)&e->d_saved_w, sizeof(float) * e->host.n));
        e->saved_w_n = e->host.n;
    }
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
    if (const int guard_rc = fs_step_guard(ctx, "fs_movep")) return guard_rc;
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
// (id -> -1) when it is stable or its loop's steps are used up, the picker kernel moves the movers' pickers.
//
// The loop state of a waiter (steps taken, stable, over) lives in ctx->d_wait[episode] ACROSS calls.  A call normally sets
// it (start[a] >= 0: "the loop stands at step start[a]"); with start[a] = -1 the call continues from whatever the previous
// call left there -- which lets a host that pipelines its calls (fs_advance_begin / fs_advance_end, flingbot_amd/schedule.py)
// queue the NEXT chunk of a wait before it has seen the result of the previous one: if the loop ended there, the episode's
// entries of the new chunk retire at once and nothing is stepped.
struct FsWaitCmd {
    double tol;   // tolerance of wait_until_stable; -1 for plain steps (never "stable")
    int env;
    int limit;    // steps of the whole loop (max_steps / n)
    int start;    // >= 0: set the loop to this many steps taken at the call's first launch sequence; -1: continue
    int pad;
};

__global__ __launch_bounds__(256) void fs_k_wait_check(const FsEnvDev *envs, int *row, const FsWaitCmd *cmds, FsWaitDev *state,
                                                       int first) {
    __shared__ float red[256];
    const int slot = blockIdx.x;
    const FsWaitCmd c = cmds[slot];
    FsWaitDev &W = state[c.env];
    int steps = W.steps, over = W.over;
    if (first && c.start >= 0) {
        steps = c.start; over = 0;
        if (threadIdx.x == 0) { W.steps = steps; W.stable = 0; W.over = 0; }
    }
    if (over) {  // the loop ended in an earlier launch sequence (or call): this sequence's list entry goes
        if (threadIdx.x == 0) row[slot] = -1;
        return;
    }
    if (steps >= c.limit) {  // the loop's steps are used up: not stable, no test (flex_utils.py:441)
        if (threadIdx.x == 0) { row[slot] = -1; W.over = 1; }
        return;
    }
    const FsEnvDev &E = envs[c.env];
    float m = 0.0f;
    for (int i = threadIdx.x; i < E.n; i += 256) {
        const FsVec4 v = E.vel[i];
        const float a = fabsf(v.x), b = fabsf(v.y), cc = fabsf(v.z);
        const float q = (a != a || b != b || cc != cc) ? __int_as_float(0x7fc00000) : fmaxf(a, fmaxf(b, cc));
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
        if ((double)red[0] < c.tol) { W.stable = 1; W.over = 1; row[slot] = -1; }  // (plain steps: tolerance -1, never true)
        else W.steps = steps + 1;  // the step that follows
    }
}

static int ticket_buffers(fs_ctx *ctx, FsAdvTicket &t, size_t bytes) {
    if (!t.done) HIP_TRY(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
    if (!t.h_wait) HIP_TRY(hipHostMalloc((void **)&t.h_wait, sizeof(FsWaitDev) * ctx->n_envs, hipHostMallocDefault));
    if (bytes > t.d_tab_bytes) {  // (the ticket is free, i.e. its previous launches have completed)
        const size_t want = bytes < (1u << 18) ? (1u << 18) : bytes * 2;
        if (t.d_tab) { fs_pool_give(ctx, t.d_tab, t.d_tab_bytes); t.d_tab = nullptr; t.d_tab_bytes = 0; }
        t.d_tab = fs_pool_take(ctx, want, &t.d_tab_bytes);
        if (!t.d_tab) return FS_ERR_HIP;
    }
    if (bytes > t.h_tab_bytes) {
        if (t.h_tab) (void)hipHostFree(t.h_tab);
        t.h_tab = nullptr; t.h_tab_bytes = 0;
        const size_t want = bytes < (1u << 18) ? (1u << 18) : bytes * 2;
        HIP_TRY(hipHostMalloc(&t.h_tab, want, hipHostMallocDefault));
        t.h_tab_bytes = want;
    }
    return FS_OK;
}

// Queues the chunk and returns: the movers' outputs are final (their trajectories are planned on the host), a waiter's
// status_out is -1 until fs_advance_end.  Returns the ticket (>= 0) or an error code (< 0).  poll: the blocking form may
// stop launching early once every waiter of a waiters-only tail has retired (it looks every 16 sequences).
static int advance_begin(fs_ctx *ctx, int n, const int *envs, const int *kind, const double *targets, const int *grasp,
                         const double *speed, const int *limit, const int *min_steps, const int *f32, const int *start,
                         double eps, const double *tolerance, int cap_min, int cap, int *progress_out, int *status_out,
                         int *steps_out, bool poll) {
    if (!ctx || n <= 0 || n > ctx->n_envs || !envs || !kind || !limit || !start || !progress_out || !status_out || !steps_out ||
        cap <= 0 || cap_min <= 0 || cap_min > cap) {
        fs_set_error("fs_advance: bad arguments");
        return FS_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->on_svc) { fs_set_error("fs_advance: chunks are queued on the main lane (call fs_service_lane(ctx, 0) first)"); return FS_ERR_STATE; }
    int tk = -1;
    for (int k = 0; k < FS_ADV_TICKETS; ++k)
        if (!ctx->tickets[k].busy) { tk = k; break; }
    if (tk < 0) { fs_set_error("fs_advance_begin: too many chunks in flight (call fs_advance_end)"); return FS_ERR_STATE; }
    FsAdvTicket &T = ctx->tickets[tk];
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
            if (start[a] < 0) { fs_set_error("fs_advance: a movep needs its loop index (start >= 0)"); return FS_ERR_ARG; }
            movers.push_back(a);
        } else if (kind[a] == 1 || kind[a] == 2) {
            if (kind[a] == 1 && !tolerance) { fs_set_error("fs_advance: tolerance missing"); return FS_ERR_ARG; }
            waiters.push_back(a);
        } else {
            fs_set_error("fs_advance: kind must be 0 (movep), 1 (wait_until_stable) or 2 (plain steps)");
            return FS_ERR_ARG;
        }
    }
    const int nm = (int)movers.size(), nw_all = (int)waiters.size();
    // movers: plan this call's part of each trajectory.  The chunk ends when the FIRST mover finishes its movep (so that its
    // program can issue the next request without idling through the others' steps), but not before cap_min steps and not
    // after cap.
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
        ctx->last_movep_steps += (long long)plans[q].cmds.size();
        if (plans[q].cmds.size() > n_seq) n_seq = plans[q].cmds.size();
    }
    const size_t mover_seq = n_seq;  // launch sequences that still have a mover
    // waiters: a loop whose steps are known to be used up is answered here; the others take part for up to `chunk` sequences
    T.w_arg.clear(); T.w_env.clear(); T.w_kind.clear(); T.w_limit.clear(); T.w_start.clear(); T.w_gen.clear();
    if (ctx->wait_over.size() != (size_t)ctx->n_envs) ctx->wait_over.assign((size_t)ctx->n_envs, 0);
    if (ctx->wait_gen.size() != (size_t)ctx->n_envs) ctx->wait_gen.assign((size_t)ctx->n_envs, 0);
    // what this call changes on the host before its launches are queued -- shape mirrors, loop flags -- is put back if queueing
    // fails, so that an error return leaves the host describing what the device really executed
    struct Undo { int env; char over; int gen; bool shapes; FsShapesDev sh; };
    std::vector<Undo> undo;
    auto rollback = [&]() {
        for (const Undo &u : undo) {
            ctx->wait_over[u.env] = u.over; ctx->wait_gen[u.env] = u.gen;
            if (u.shapes) ctx->envs[u.env].shapes = u.sh;
        }
        ctx->last_movep_steps = 0;
    };
    for (int q = 0; q < nw_all; ++q) {
        const int a = waiters[q];
        if (start[a] >= 0 && limit[a] - start[a] <= 0) {  // wait_until_stable returns False, plain steps are done
            progress_out[a] = start[a]; status_out[a] = kind[a] == 2 ? 1 : 2; steps_out[a] = 0;
            continue;
        }
        int budget = chunk;
        if (start[a] >= 0 && limit[a] - start[a] < budget) budget = limit[a] - start[a];
        if ((size_t)budget > n_seq) n_seq = (size_t)budget;
        status_out[a] = -1; progress_out[a] = start[a]; steps_out[a] = 0;
        T.w_arg.push_back(a); T.w_env.push_back(envs[a]); T.w_kind.push_back(kind[a]); T.w_limit.push_back(limit[a]);
        T.w_start.push_back(start[a]);
        if (start[a] >= 0) {  // a new loop
            undo.push_back(Undo{envs[a], ctx->wait_over[envs[a]], ctx->wait_gen[envs[a]], false, FsShapesDev{}});
            ctx->wait_over[envs[a]] = 0;
            ctx->wait_gen[envs[a]] += 1;
        }
        T.w_gen.push_back(ctx->wait_gen[envs[a]]);
    }
    const int nw = (int)T.w_arg.size();
    T.n = n; T.n_seq = n_seq;
    T.listed.clear();  // the episodes this call's launches may touch: its movers and the waiters that take part
    for (int q = 0; q < nm; ++q) T.listed.push_back(envs[movers[q]]);
    T.listed.insert(T.listed.end(), T.w_env.begin(), T.w_env.end());
    // host mirrors of the shape states follow the planned trajectories
    for (int q = 0; q < nm; ++q) {
        FsEnv &e = ctx->envs[envs[movers[q]]];
        const auto &cm = plans[q].cmds;
        if (cm.empty()) continue;
        undo.push_back(Undo{envs[movers[q]], ctx->wait_over[envs[movers[q]]], ctx->wait_gen[envs[movers[q]]], true, e.shapes});
        for (int k = 0; k < S; ++k) {
            const float r = e.shapes.pos[k].w;
            const float *pv = cm.size() >= 2 ? cm[cm.size() - 2].new_pos[k] : &e.shapes.pos[k].x;
            e.shapes.prev[k] = FsVec4{pv[0], pv[1], pv[2], r};
            e.shapes.pos[k] = FsVec4{cm.back().new_pos[k][0], cm.back().new_pos[k][1], cm.back().new_pos[k][2], r};
        }
    }
    if (n_seq == 0) {  // nothing to launch: an empty ticket keeps the begin / end protocol uniform
        T.busy = true; ctx->tickets_busy++;
        int rc0 = ticket_buffers(ctx, T, 16);
        if (rc0 != FS_OK) { T.busy = false; ctx->tickets_busy--; rollback(); return rc0; }
        if (!fs_hip_ok(hipEventRecord(T.done, ctx->stream), "fs_advance event")) { T.busy = false; ctx->tickets_busy--; rollback(); return FS_ERR_HIP; }
        return tk;
    }
    // device tables, ONE upload from the ticket's pinned image.  Per launch sequence s a row of the launch list:
    // [waiters | movers still moving at s | -1 ...]; the check kernel retires a waiter from its row and, through the loop
    // state, from every later row (and call).
    const int width = nm > 0 ? nm : 1, W = nw + width, n_envs = ctx->n_envs;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) & ~size_t(15); return o; };
    const size_t o_picked = carve(sizeof(int *) * n_envs), o_saved = carve(sizeof(float *) * n_envs);
    const size_t o_wcmd = carve(sizeof(FsWaitCmd) * (nw > 0 ? nw : 1));
    const size_t o_cmds = carve(sizeof(FsPickerCmd) * n_seq * width);
    const size_t o_rows = carve(sizeof(int) * n_seq * W);
    int rc = ticket_buffers(ctx, T, off);
    if (rc != FS_OK) { rollback(); return rc; }
    char *blob = (char *)T.h_tab;
    memset(blob, 0, off);
    int **h_picked = (int **)(blob + o_picked);
    float **h_saved = (float **)(blob + o_saved);
    FsWaitCmd *h_wcmd = (FsWaitCmd *)(blob + o_wcmd);
    FsPickerCmd *h_cmds = (FsPickerCmd *)(blob + o_cmds);
    int *h_rows = (int *)(blob + o_rows);
    std::vector<int> h_cnt(n_seq, 0);
    for (int i = 0; i < n_envs; ++i) { h_picked[i] = ctx->envs[i].d_picked; h_saved[i] = ctx->envs[i].d_saved_w; }
    for (int q = 0; q < nw; ++q) {
        const int a = T.w_arg[q];
        h_wcmd[q].tol = kind[a] == 1 ? tolerance[a] : -1.0;
        h_wcmd[q].env = envs[a]; h_wcmd[q].limit = limit[a]; h_wcmd[q].start = start[a]; h_wcmd[q].pad = 0;
    }
    for (size_t s = 0; s < n_seq; ++s) {
        int *row = h_rows + s * W;
        for (int k = 0; k < W; ++k) row[k] = -1;
        for (int q = 0; q < nw; ++q) row[q] = T.w_env[q];
        for (int q = 0; q < nm; ++q)
            if (s < plans[q].cmds.size()) {
                const int slot = h_cnt[s]++;
                row[nw + slot] = envs[movers[q]];
                h_cmds[s * width + slot] = plans[q].cmds[s];
            }
    }
    char *dev = (char *)T.d_tab;
    T.busy = true; ctx->tickets_busy++;
    auto fail = [&](int code) { (void)hipStreamSynchronize(ctx->stream); T.busy = false; ctx->tickets_busy--; rollback(); return code; };
    if (!fs_hip_ok(hipMemcpyAsync(dev, blob, off, hipMemcpyHostToDevice, ctx->stream), "fs_advance upload")) return fail(FS_ERR_HIP);
    if (!ctx->adv_ev0) {
        if (!fs_hip_ok(hipEventCreate(&ctx->adv_ev0), "event") || !fs_hip_ok(hipEventCreate(&ctx->adv_ev1), "event")) return fail(FS_ERR_HIP);
    }
    if (poll) (void)hipEventRecord(ctx->adv_ev0, ctx->stream);
    const auto wall1 = std::chrono::steady_clock::now();
    int **d_picked = (int **)(dev + o_picked);
    float **d_saved = (float **)(dev + o_saved);
    const FsWaitCmd *d_wcmd = (const FsWaitCmd *)(dev + o_wcmd);
    const FsPickerCmd *d_cmds = (const FsPickerCmd *)(dev + o_cmds);
    int *d_rows = (int *)(dev + o_rows);
    std::vector<int> ids;
    size_t launched = 0;
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
            hipLaunchKernelGGL(fs_k_wait_check, dim3(nw), dim3(256), 0, ctx->stream, ctx->d_envs, d_row, d_wcmd, ctx->d_wait,
                               s == 0 ? 1 : 0);
        rc = fs_step_ids(ctx, ids, 1, d_row);
        ++launched;
        if (poll && rc == FS_OK && nw > 0 && s >= mover_seq && (s & 15) == 15 && s + 1 < n_seq) {  // only waiters left: all retired?
            hipError_t pe = hipMemcpyAsync(T.h_wait, ctx->d_wait, sizeof(FsWaitDev) * n_envs, hipMemcpyDeviceToHost, ctx->stream);
            if (pe == hipSuccess) pe = hipStreamSynchronize(ctx->stream);
            if (!fs_hip_ok(pe, "fs_advance poll")) { rc = FS_ERR_HIP; break; }
            bool any = false;
            for (int q = 0; q < nw; ++q) any = any || T.h_wait[T.w_env[q]].over == 0;
            if (!any) break;
        }
    }
    if (rc != FS_OK) return fail(rc);
    hipError_t err = hipSuccess;
    if (nw > 0) err = hipMemcpyAsync(T.h_wait, ctx->d_wait, sizeof(FsWaitDev) * n_envs, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess && poll) err = hipEventRecord(ctx->adv_ev1, ctx->stream);
    if (err == hipSuccess) err = hipEventRecord(T.done, ctx->stream);
    if (!fs_hip_ok(err, "fs_advance results")) return fail(FS_ERR_HIP);
    const auto wall2 = std::chrono::steady_clock::now();
    ctx->adv_prep_ms += std::chrono::duration<double, std::milli>(wall1 - wall0).count();
    T.wall_begin_ms = std::chrono::duration<double, std::milli>(wall2 - wall0).count();
    ctx->adv_calls += 1;
    ctx->adv_sequences += (long long)launched;
    return tk;
}

// Waits for the ticket's launches and fills in the waiters' outputs: progress = steps of the loop taken so far, status 1 =
// ended (stable / plain steps done), 2 = ended at its limit without becoming stable, 0 = call again; steps = progress -
// start (-1 when the call continued a loop whose position the host did not state: start = -1).
static int advance_end(fs_ctx *ctx, int ticket, int *progress_out, int *status_out, int *steps_out, bool timed) {
    if (!ctx || ticket < 0 || ticket >= FS_ADV_TICKETS || !ctx->tickets[ticket].busy) {
        fs_set_error("fs_advance_end: no such chunk in flight");
        return FS_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    FsAdvTicket &T = ctx->tickets[ticket];
    const auto wall0 = std::chrono::steady_clock::now();
    const hipError_t err = hipEventSynchronize(T.done);
    T.busy = false; ctx->tickets_busy--;
    HIP_TRY(err);
    if (timed) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->adv_ev0, ctx->adv_ev1) == hipSuccess) ctx->adv_gpu_ms += ms;
    }
    const int nw = (int)T.w_arg.size();
    for (int q = 0; q < nw && T.n_seq > 0; ++q) {
        const int a = T.w_arg[q];
        const FsWaitDev w = T.h_wait[T.w_env[q]];
        if (progress_out) progress_out[a] = w.steps;
        if (steps_out) steps_out[a] = T.w_start[q] >= 0 ? w.steps - T.w_start[q] : -1;
        const int st = w.stable ? 1 : ((w.over || w.steps >= T.w_limit[q]) ? (T.w_kind[q] == 2 ? 1 : 2) : 0);
        if (status_out) status_out[a] = st;
        // (over on the device, or its next check finds the steps used up) -- but only the loop this entry was queued for: a
        // later ticket may already have started the episode's NEXT loop, which this report must not end on the host
        if (st != 0 && q < (int)T.w_gen.size() && T.w_gen[q] == ctx->wait_gen[T.w_env[q]]) ctx->wait_over[T.w_env[q]] = 1;
    }
    ctx->adv_wall_ms += T.wall_begin_ms + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return FS_OK;
}

extern "C" int fs_advance_begin(fs_ctx *ctx, int n, const int *envs, const int *kind, const double *targets, const int *grasp,
                                const double *speed, const int *limit, const int *min_steps, const int *f32, const int *start,
                                double eps, const double *tolerance, int cap_min, int cap, int *progress_out, int *status_out,
                                int *steps_out) {
    return advance_begin(ctx, n, envs, kind, targets, grasp, speed, limit, min_steps, f32, start, eps, tolerance, cap_min, cap,
                         progress_out, status_out, steps_out, false);
}
extern "C" int fs_advance_end(fs_ctx *ctx, int ticket, int *progress_out, int *status_out, int *steps_out) {
    return advance_end(ctx, ticket, progress_out, status_out, steps_out, false);
}
extern "C" int fs_advance_in_flight(const fs_ctx *ctx) { return ctx ? ctx->tickets_busy : FS_ERR_ARG; }

extern "C" int fs_advance(fs_ctx *ctx, int n, const int *envs, const int *kind, const double *targets, const int *grasp,
                          const double *speed, const int *limit, const int *min_steps, const int *f32, const int *start,
                          double eps, const double *tolerance, int cap_min, int cap, int *progress_out, int *status_out,
                          int *steps_out) {
    if (ctx && start)
        for (int a = 0; a < n; ++a)
            if (start[a] < 0) { fs_set_error("fs_advance: start < 0 (continue from the device state) is for fs_advance_begin"); return FS_ERR_ARG; }
    const int tk = advance_begin(ctx, n, envs, kind, targets, grasp, speed, limit, min_steps, f32, start, eps, tolerance, cap_min,
                                 cap, progress_out, status_out, steps_out, true);
    if (tk < 0) return tk;
    return advance_end(ctx, tk, progress_out, status_out, steps_out, ctx->tickets[tk].n_seq > 0);
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
