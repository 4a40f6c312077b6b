// fs_solver.hip -- launch logic of the two solver back-ends.
#include <hip/hip_runtime.h>

#include "../../include/flingsim.h"
#include "fs_context.h"
#include "fs_fused_kernel.h"
#include "fs_fused_grid_kernel.h"
#include "fs_stream_kernels.h"

#include <algorithm>
#include <cstdlib>

#define HIP_TRY(call)                                     \
    do {                                                  \
        if (!fs_hip_ok((call), #call)) return FS_ERR_HIP; \
    } while (0)

// number of concurrent chains of a streaming launch (0 = FLINGSIM_STREAM_GROUPS unset: the measured default below)
static int fs_default_stream_groups(int ne, size_t particles) {
    static const int env_groups = [] { const char *v = getenv("FLINGSIM_STREAM_GROUPS"); return v ? atoi(v) : 0; }();
    if (env_groups > 0) return env_groups;
    // measured on crumpled 64x64 cloths (scripts/boundary_timing.py with FLINGSIM_STREAM_GROUPS = 1 / 2 / 3, ms per step; chain 0
    // on the context's own stream):  32 episodes 1.03 / 0.99 / 1.09   64: 1.39 / 1.14 / 1.18   128: 2.25 / 1.76 / 1.74   256: 4.16 / 3.51 / 3.59
    // -- two chains from ~24 x 4096 particles on; a third hardware queue gains nothing, a fourth (and a fifth: the first
    // version kept the context's stream idle next to the chains') makes everything slower
    (void)ne;
    if (particles >= (size_t)24 * 4096) return 2;
    return 1;
}

static int upload_ids(fs_ctx *ctx, const std::vector<int> &ids) {
    // the same list as last time (a loop of fs_step calls): ctx->d_ids still holds it -- nothing on the device writes there --
    // and skipping the upload skips the synchronisation below, so the host can queue the next frame while this one runs
    if (ctx->d_ids_valid && ctx->uploaded_ids == ids) return FS_OK;
    ctx->d_ids_valid = false;
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // h_ids may still be read by an earlier copy
    for (size_t k = 0; k < ids.size(); ++k) ctx->h_ids[k] = ids[k];
    HIP_TRY(hipMemcpyAsync(ctx->d_ids, ctx->h_ids, sizeof(int) * ids.size(), hipMemcpyHostToDevice, ctx->stream));
    ctx->uploaded_ids = ids;
    ctx->d_ids_valid = true;
    return FS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming back-end: per substep  predict -> grid scan -> grid scatter -> find neighbours -> I x iterate -> finalize
int fs_step_stream(fs_ctx *ctx, const std::vector<int> &ids, int n_steps, const int *d_ids_in) {
    const int *d_ids = d_ids_in ? d_ids_in : ctx->d_ids;
    if (!d_ids_in) {
        int rc = upload_ids(ctx, ids);
        if (rc != FS_OK) return rc;
    }
    int max_n = 0, substeps = 0, iters = 0;
    for (int id : ids) {
        const FsEnv &e = ctx->envs[id];
        if (e.host.n > max_n) max_n = e.host.n;
        if (substeps == 0) { substeps = e.dev.p.numSubsteps; iters = e.dev.p.numIterations; }
        if (e.dev.p.numSubsteps != substeps || e.dev.p.numIterations != iters) {
            fs_set_error("episodes stepped together must share numSubsteps / numIterations");
            return FS_ERR_STATE;
        }
    }
    if (substeps > FS_SWEEP_MAX_SUBSTEPS) {  // (the reference runs 4: softgym_cloth.h:154)
        fs_set_error("the streaming back-end tabulates the kinematic spheres' sweeps for at most 8 substeps per frame");
        return FS_ERR_STATE;
    }
    // XCD-affine 1-D launch (fs_stream_tile): gx workgroups per episode slot, slots rounded up to a multiple of 8
    const int gx = (max_n + FS_TILE - 1) / FS_TILE, ne = (int)ids.size();
    const dim3 grid((unsigned)(((ne + 7) / 8) * 8) * (unsigned)gx);
    const dim3 block(FS_TILE);
    // launches that cannot fill the chip are latency-bound: take the form of fs_k_iterate that requests everything up front
    const bool eager = (size_t)max_n * ids.size() <= (size_t)32 * 4096;
    // one-byte spring codes + dictionary when every episode of the launch has them.  Throughput form only: measured 5-8 %
    // faster there (104x104 x 64 episodes 2.66 -> 2.52 ms/step, the 32-episode evaluation loop 5.0 -> 4.6 s), while the
    // latency form loses the time of the dictionary's barrier (one 104x104 episode 0.74 -> 0.79 ms/step)
    bool coded = !ctx->force_ell_stream && !eager;
    // grid cloths: neighbour ids computed from the particle's grid coordinates (fs_k_iterate_grid)
    bool grid_form = !ctx->force_ell_stream;
    for (int id : ids) {
        const FsEnvDev &d = ctx->envs[id].dev;
        grid_form = grid_form && d.sdict_size > 0 && d.gp_count > 0 && d.gp_count <= FS_GRID_SLOTS;
    }
    grid_form = grid_form && (size_t)max_n * ids.size() >= (size_t)96 * 4096;
    // canonical grid cloths: neighbour ids from the grid coordinates, rest lengths from the per-particle table -- no
    // adjacency, no dictionary, every load of the spring phase in one round trip (fs_k_iterate_gridl)
    bool gridl_form = !ctx->force_ell_stream && !ctx->force_coded_stream;
    bool gridl_posk = true;  // no tethers anywhere in the launch: the spring form without the slack test
    for (int id : ids) {
        gridl_form = gridl_form && ctx->envs[id].dev.gp_L_ok;
        gridl_posk = gridl_posk && ctx->envs[id].dev.gp_halvable;
    }
    size_t launch_particles = 0;
    for (int id : ids) launch_particles += (size_t)ctx->envs[id].host.n;
    bool find_stencil = true;  // every episode a grid cloth whose SelfCollideFilter test is the 8-neighbour stencil (find mode 4)
    for (int id : ids) find_stencil = find_stencil && ctx->envs[id].dev.find_mode == 4;
    const bool gridl_halvable_all = gridl_posk;   // (every episode tether-free, before the launch-size rule makes gridl_posk mean "the POSK form")
    static const int posk_env = [] { const char *v = getenv("FLINGSIM_GRIDL_POSK"); return v ? atoi(v) : 0; }();  // developer: 1 always, -1 never
    gridl_posk = gridl_posk && (posk_env > 0 || (posk_env == 0 && launch_particles <= (size_t)4 * 1024 * 64));  // one round of 4 waves per SIMD at most (see fs_k_iterate_gridl)
    hipStream_t st = ctx->stream;
    ctx->last_form = gridl_form ? FS_FORM_STREAM_GRIDL
                     : grid_form ? FS_FORM_STREAM_GRID
                                 : (eager ? FS_FORM_STREAM_EAGER : (coded ? FS_FORM_STREAM_CODED : FS_FORM_STREAM_ELL));
    // slot-indexed copies of the listed episodes' descriptors: one scalar indirection less in front of every kernel below
    // (built once for a loop of calls with the context's own, unchanged list and unchanged descriptors; a caller's device
    // list -- fs_advance, fs_wait_until_stable: slots retire on the device -- rebuilds it every call)
    const bool table_ok = !d_ids_in && ctx->table_epoch == ctx->desc_epoch && ctx->table_ids == ids;
    if (!table_ok) {
        hipLaunchKernelGGL(fs_k_slot_table, dim3((unsigned)ne), dim3(64), 0, st, ctx->d_envs, d_ids, ctx->d_slot_envs,
                           ctx->d_shapes, ctx->d_slot_sweeps);
        ctx->table_epoch = d_ids_in ? ~0ull : ctx->desc_epoch;
        if (!d_ids_in) ctx->table_ids = ids;
    }
    // substep boundaries in one launch each (finalize + predict + bucket sort, fs_k_boundary) when every cloth fits it
    // (one workgroup per episode: launches of fewer than 16 episodes are 2-3 % faster with the four small kernels spread
    // over the chip -- measured, scripts/boundary_timing.py -- and keep them unless FS_SOLVER_STREAM_MERGED asks otherwise)
    const bool merged = max_n <= FS_BOUND_MAX && !ctx->force_split_boundary && (ne >= 16 || ctx->force_merged_boundary);
    ctx->last_boundary = merged ? 1 : 0;
    if (merged && !ctx->bound_attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_boundary<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_BOUND_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_boundary<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_BOUND_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_boundary<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_BOUND_LDS_BYTES));
        ctx->bound_attr_set = true;
    }
    // Concurrent chains.  A frame is 129 DEPENDENT launches, each about one wave's critical path long whatever the amount of
    // work (measurements in EXPERIMENTS.md), so a launch list that cannot fill the chip is split into `groups` slot ranges (multiples of 8:
    // the XCD mapping of fs_stream_tile) whose chains run on streams of their own: while one chain's kernel drains and its
    // successor starts up, the other chains' kernels compute.  The chains are launched interleaved from this thread, fork
    // from and join into the context's stream through events; episodes are independent, results are bit-identical.
    int groups = ctx->stream_groups > 0 ? ctx->stream_groups : fs_default_stream_groups(ne, launch_particles);
    if (groups > FS_MAX_STREAM_GROUPS) groups = FS_MAX_STREAM_GROUPS;
    if (groups > (ne + 7) / 8) groups = (ne + 7) / 8;
    if (groups < 1) groups = 1;
    struct Chain { int first, count, gx; dim3 grid; hipStream_t st; };
    Chain chain[FS_MAX_STREAM_GROUPS];
    {
        const int blocks8 = (ne + 7) / 8;
        int at = 0;
        for (int g = 0; g < groups; ++g) {
            const int b8 = blocks8 / groups + (g < blocks8 % groups ? 1 : 0);
            Chain &c = chain[g];
            c.first = at * 8;
            c.count = (at + b8) * 8 <= ne ? b8 * 8 : ne - at * 8;
            at += b8;
            int mx = 0;
            for (int k = 0; k < c.count; ++k) mx = std::max(mx, ctx->envs[ids[c.first + k]].host.n);
            c.gx = (mx + FS_TILE - 1) / FS_TILE;
            c.grid = dim3((unsigned)(((c.count + 7) / 8) * 8) * (unsigned)c.gx);
            c.st = st;
        }
    }
    // From the fork to the join an early return would leave chains running on the episodes' slabs while the caller believes
    // the context's stream is all there is to wait for: every error in between waits for ALL streams before it returns.
    bool forked = false;
#define FORK_TRY(call)                                    \
    do {                                                  \
        if (!fs_hip_ok((call), #call)) {                  \
            if (forked) fs_sync_all_streams(ctx);         \
            return FS_ERR_HIP;                            \
        }                                                 \
    } while (0)
    if (groups > 1) {  // chain 0 stays on the context's stream, the others fork from it
        for (int g = 1; g < groups; ++g) {
            if (!ctx->aux_streams[g]) HIP_TRY(hipStreamCreateWithFlags(&ctx->aux_streams[g], hipStreamNonBlocking));
            if (!ctx->aux_events[g]) HIP_TRY(hipEventCreateWithFlags(&ctx->aux_events[g], hipEventDisableTiming));
            chain[g].st = ctx->aux_streams[g];
        }
        if (!ctx->fork_event) HIP_TRY(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ctx->fork_event, st));
        forked = true;
        for (int g = 1; g < groups; ++g) FORK_TRY(hipStreamWaitEvent(chain[g].st, ctx->fork_event, 0));
    }
    auto iter_kernel = eager ? fs_k_iterate_eager<false> : (coded ? fs_k_iterate<true> : fs_k_iterate<false>);
    if (grid_form) iter_kernel = fs_k_iterate_grid;
    if (gridl_form) iter_kernel = gridl_posk ? fs_k_iterate_gridl<true> : fs_k_iterate_gridl<false>;
    // large launches of tether-free cloths: the throughput instantiation (fs_stream_kernels.h TP; FLINGSIM_GRIDL_TP=0 keeps the general form)
    const char *tp_str = getenv("FLINGSIM_GRIDL_TP");   // (read per call: the tests switch it inside one process)
    const int tp_env = tp_str ? atoi(tp_str) : 1;
    if (gridl_form && gridl_halvable_all && ((!gridl_posk && tp_env > 0) || tp_env > 1)) {   // (FLINGSIM_GRIDL_TP=2, developer: at every launch size)
        iter_kernel = fs_k_iterate_gridl_tp;
        ctx->last_form = FS_FORM_STREAM_GRIDL_TP;
    }
    const int flip_end = iters & 1;
    // the frame's launch sequence, issued for every chain in turn (interleaved from this thread: one host thread per chain was
    // measured too and is no faster -- the host is not the limit at two or three chains)
    enum { K_BOUND_FIRST, K_BOUND_MID, K_BOUND_LAST, K_PREDICT, K_SCAN, K_SCATTER, K_FIND, K_ITER, K_FINALIZE };
    auto launch = [&](int kind, int sub, int flip) {
        for (int g = 0; g < groups; ++g) {
            const Chain &c = chain[g];
            const FsEnvDev *tab = ctx->d_slot_envs + c.first;
            const int *cids = d_ids + c.first;
            const dim3 bgrid((unsigned)c.count), bblock(FS_BOUND_THREADS);
            switch (kind) {
                case K_BOUND_FIRST: hipLaunchKernelGGL((fs_k_boundary<false, true>), bgrid, bblock, FS_BOUND_LDS_BYTES, c.st, tab, cids, 0); break;
                case K_BOUND_MID: hipLaunchKernelGGL((fs_k_boundary<true, true>), bgrid, bblock, FS_BOUND_LDS_BYTES, c.st, tab, cids, flip); break;
                case K_BOUND_LAST: hipLaunchKernelGGL((fs_k_boundary<true, false>), bgrid, bblock, FS_BOUND_LDS_BYTES, c.st, tab, cids, flip); break;
                case K_PREDICT: hipLaunchKernelGGL(fs_k_predict, c.grid, block, 0, c.st, tab, cids, c.gx, c.count); break;
                case K_SCAN: hipLaunchKernelGGL(fs_k_grid_scan, bgrid, dim3(1024), 0, c.st, tab, cids); break;
                case K_SCATTER: hipLaunchKernelGGL(fs_k_grid_scatter, c.grid, block, 0, c.st, tab, cids, c.gx, c.count); break;
                case K_FIND:
                    if (find_stencil) hipLaunchKernelGGL(fs_k_find_neighbors<true>, c.grid, block, 0, c.st, tab, ctx->d_slot_sweeps + c.first, cids, sub, c.gx, c.count);
                    else hipLaunchKernelGGL(fs_k_find_neighbors<false>, c.grid, block, 0, c.st, tab, ctx->d_slot_sweeps + c.first, cids, sub, c.gx, c.count);
                    break;
                case K_ITER: hipLaunchKernelGGL(iter_kernel, c.grid, block, 0, c.st, tab, ctx->d_slot_sweeps + c.first, cids, sub, flip, c.gx, c.count); break;
                default: hipLaunchKernelGGL(fs_k_finalize, c.grid, block, 0, c.st, tab, cids, flip, c.gx, c.count); break;
            }
        }
    };
    for (int f = 0; f < n_steps; ++f) {
        if (f > 0) FORK_TRY(hipGetLastError());  // a failed launch shows up here, one frame (129 launches) later at most
        for (int sub = 0; sub < substeps; ++sub) {
            if (merged) {
                // (K_BOUND_MID: the finalize of the previous substep -- also the previous frame's last one -- rides along)
                launch(f == 0 && sub == 0 ? K_BOUND_FIRST : K_BOUND_MID, sub, flip_end);
            } else {
                launch(K_PREDICT, sub, 0);
                launch(K_SCAN, sub, 0);
                launch(K_SCATTER, sub, 0);
            }
            launch(K_FIND, sub, 0);
            for (int it = 0; it < iters; ++it) launch(K_ITER, sub, it & 1);
            if (!merged) launch(K_FINALIZE, sub, flip_end);
        }
    }
    if (merged) launch(K_BOUND_LAST, 0, flip_end);
    for (int g = 1; g < groups; ++g) {
        FORK_TRY(hipEventRecord(ctx->aux_events[g], chain[g].st));
        FORK_TRY(hipStreamWaitEvent(st, ctx->aux_events[g], 0));
    }
    ctx->last_stream_groups = groups;
    FORK_TRY(hipGetLastError());
#undef FORK_TRY
    return FS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused back-end: one workgroup per episode, LDS-resident particle state, the whole frame(s) in one launch.
bool fs_fused_supported(const fs_ctx *ctx, const FsEnv &env) {
    (void)ctx;
    return env.has_scene && env.host.n <= FS_FUSED_MAX_PARTICLES && env.host.max_deg <= FS_FUSED_MAX_DEG &&
           env.dev.p.numPlanes <= 1;
}

int fs_step_fused(fs_ctx *ctx, const std::vector<int> &ids, int n_steps, const int *d_ids_in) {
    const int *d_ids = d_ids_in ? d_ids_in : ctx->d_ids;
    if (!d_ids_in) {
        int rc = upload_ids(ctx, ids);
        if (rc != FS_OK) return rc;
    }
    // the attribute belongs to the DEVICE's copy of the kernel: tracked per context (a context is bound to one device)
    if (!ctx->fused_attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_fused_step<12>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    FS_FUSED_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_fused_step<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    FS_FUSED_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_fused_step<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    FS_FUSED_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_fused_grid64, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    FG_LDS_BYTES));
        ctx->fused_attr_set = true;
    }
    // register-resident (dictionary-coded) adjacency when every episode of the launch has one
    int slots = ctx->force_generic_fused ? 0 : 12;
    for (int id : ids) {
        const FsEnv &e = ctx->envs[id];
        if (e.dev.dict_size <= 0 || e.host.max_deg > 16) slots = 0;
        else if (e.host.max_deg > 12 && slots == 12) slots = 16;
    }
    // 64-wide grid cloths (the 64 x 64 cloth of the headline metric): packed two-particle spring form without adjacency
    bool grid64 = !ctx->force_generic_fused && !ctx->force_coded_fused && slots == 12;
    for (int id : ids) grid64 = grid64 && ctx->envs[id].dev.g64_ok && ctx->envs[id].dev.p.numPlanes == 1;
    const dim3 grid((unsigned)ids.size()), block(FS_FUSED_THREADS);
    if (grid64) {
        ctx->last_form = FS_FORM_FUSED_GRID64;
        hipLaunchKernelGGL(fs_k_fused_grid64, grid, block, FG_LDS_BYTES, ctx->stream, ctx->d_envs, ctx->d_shapes, d_ids,
                           n_steps);
        HIP_TRY(hipGetLastError());
        return FS_OK;
    }
    ctx->last_form = slots == 12 ? FS_FORM_FUSED_12 : (slots == 16 ? FS_FORM_FUSED_16 : FS_FORM_FUSED_GENERIC);
    if (slots == 12)
        hipLaunchKernelGGL(fs_k_fused_step<12>, grid, block, FS_FUSED_LDS_BYTES, ctx->stream, ctx->d_envs, ctx->d_shapes,
                           d_ids, n_steps);
    else if (slots == 16)
        hipLaunchKernelGGL(fs_k_fused_step<16>, grid, block, FS_FUSED_LDS_BYTES, ctx->stream, ctx->d_envs, ctx->d_shapes,
                           d_ids, n_steps);
    else
        hipLaunchKernelGGL(fs_k_fused_step<0>, grid, block, FS_FUSED_LDS_BYTES, ctx->stream, ctx->d_envs, ctx->d_shapes,
                           d_ids, n_steps);
    HIP_TRY(hipGetLastError());
    return FS_OK;
}

// white box: the solver's reciprocal square root (fs_constraints.h fs_rsqrt) evaluated on the device for n host values
__global__ void fs_k_eval_rsqrt(const float *__restrict__ x, float *__restrict__ y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fs_rsqrt(x[i]);
}
extern "C" int fs_eval_rsqrt(fs_ctx *ctx, const float *x, float *y, int n) {
    if (!ctx || !x || !y || n < 0) return FS_ERR_ARG;
    if (n == 0) return FS_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, size_t(8) * n));
    int rc = FS_OK;
    if (hipMemcpy(d, x, size_t(4) * n, hipMemcpyHostToDevice) != hipSuccess) rc = FS_ERR_HIP;
    if (rc == FS_OK) {
        fs_k_eval_rsqrt<<<(n + 255) / 256, 256>>>(d, d + n, n);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(y, d + n, size_t(4) * n, hipMemcpyDeviceToHost) != hipSuccess)
            rc = FS_ERR_HIP;
    }
    (void)hipFree(d);
    if (rc != FS_OK) fs_set_error("fs_eval_rsqrt: HIP error");
    return rc;
}
