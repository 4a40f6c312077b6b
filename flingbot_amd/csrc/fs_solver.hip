// fs_solver.hip -- launch logic of the two solver back-ends.
#include <hip/hip_runtime.h>

#include "../../include/flingsim.h"
#include "fs_context.h"
#include "fs_fused_kernel.h"
#include "fs_fused_grid_kernel.h"
#include "fs_stream_kernels.h"

#define HIP_TRY(call)                                     \
    do {                                                  \
        if (!fs_hip_ok((call), #call)) return FS_ERR_HIP; \
    } while (0)

static int upload_ids(fs_ctx *ctx, const std::vector<int> &ids) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // h_ids may still be read by an earlier copy
    for (size_t k = 0; k < ids.size(); ++k) ctx->h_ids[k] = ids[k];
    HIP_TRY(hipMemcpyAsync(ctx->d_ids, ctx->h_ids, sizeof(int) * ids.size(), hipMemcpyHostToDevice, ctx->stream));
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
    gridl_posk = gridl_posk && launch_particles <= (size_t)4 * 1024 * 64;  // one round of 4 waves per SIMD at most (see fs_k_iterate_gridl)
    hipStream_t st = ctx->stream;
    ctx->last_form = gridl_form ? FS_FORM_STREAM_GRIDL
                     : grid_form ? FS_FORM_STREAM_GRID
                                 : (eager ? FS_FORM_STREAM_EAGER : (coded ? FS_FORM_STREAM_CODED : FS_FORM_STREAM_ELL));
    // slot-indexed copies of the listed episodes' descriptors: one scalar indirection less in front of every kernel below
    hipLaunchKernelGGL(fs_k_slot_table, dim3((unsigned)ne), dim3(64), 0, st, ctx->d_envs, d_ids, ctx->d_slot_envs);
    const FsEnvDev *tab = ctx->d_slot_envs;
    // substep boundaries in one launch each (finalize + predict + bucket sort, fs_k_boundary) when every cloth fits it
    // (one workgroup per episode: launches of fewer than 16 episodes are 2-3 % faster with the four small kernels spread
    // over the chip -- measured, scripts/boundary_timing.py -- and keep them unless FS_SOLVER_STREAM_MERGED asks otherwise)
    const bool merged = max_n <= FS_BOUND_MAX && !ctx->force_split_boundary && (ne >= 16 || ctx->force_merged_boundary);
    if (merged && !ctx->bound_attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_boundary<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_BOUND_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_boundary<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_BOUND_LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void *)fs_k_boundary<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FS_BOUND_LDS_BYTES));
        ctx->bound_attr_set = true;
    }
    const dim3 bgrid((unsigned)ne), bblock(FS_BOUND_THREADS);
    for (int f = 0; f < n_steps; ++f) {
        for (int sub = 0; sub < substeps; ++sub) {
            if (merged) {
                if (f == 0 && sub == 0)
                    hipLaunchKernelGGL((fs_k_boundary<false, true>), bgrid, bblock, FS_BOUND_LDS_BYTES, st, tab, d_ids, 0);
                else  // the finalize of the previous substep (also the previous frame's last one) rides along
                    hipLaunchKernelGGL((fs_k_boundary<true, true>), bgrid, bblock, FS_BOUND_LDS_BYTES, st, tab, d_ids, iters & 1);
            } else {
                hipLaunchKernelGGL(fs_k_predict, grid, block, 0, st, tab, d_ids, gx, ne);
                hipLaunchKernelGGL(fs_k_grid_scan, dim3((unsigned)ids.size()), dim3(1024), 0, st, tab, d_ids);
                hipLaunchKernelGGL(fs_k_grid_scatter, grid, block, 0, st, tab, d_ids, gx, ne);
            }
            hipLaunchKernelGGL(fs_k_find_neighbors, grid, block, 0, st, tab, d_ids, gx, ne);
            for (int it = 0; it < iters; ++it) {
                auto kern = eager ? fs_k_iterate_eager<false> : (coded ? fs_k_iterate<true> : fs_k_iterate<false>);
                if (grid_form) kern = fs_k_iterate_grid;
                if (gridl_form) kern = gridl_posk ? fs_k_iterate_gridl<true> : fs_k_iterate_gridl<false>;
                hipLaunchKernelGGL(kern, grid, block, 0, st, tab, ctx->d_shapes, d_ids, sub, it & 1, gx, ne);
            }
            if (!merged) hipLaunchKernelGGL(fs_k_finalize, grid, block, 0, st, tab, d_ids, iters & 1, gx, ne);
        }
    }
    if (merged) hipLaunchKernelGGL((fs_k_boundary<true, false>), bgrid, bblock, FS_BOUND_LDS_BYTES, st, tab, d_ids, iters & 1);
    HIP_TRY(hipGetLastError());
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
