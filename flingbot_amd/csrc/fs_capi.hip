// fs_capi.hip -- C-ABI of libflingsim (include/flingsim.h) and the context runtime.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/flingsim.h"
#include "fs_context.h"
#ifndef FS_STENCIL_FILTER
#define FS_STENCIL_FILTER 1
#endif

static thread_local std::string g_err;
void fs_set_error(const std::string &msg) { g_err = msg; }
bool fs_hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    fs_set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define HIP_TRY(call)                                   \
    do {                                                \
        if (!fs_hip_ok((call), #call)) return FS_ERR_HIP; \
    } while (0)

extern "C" const char *fs_last_error(void) { return g_err.c_str(); }
extern "C" int fs_version(void) { return 100; }

FsTopologyDev::~FsTopologyDev() {
    if (!slab) return;
    if (owner) fs_pool_give(owner, slab, bytes);
    else (void)hipFree(slab);
}

// ---- buffer pool: hipFree synchronises the device and hipMalloc costs 0.1-0.3 ms, so episode slabs and topology slabs
// of a running context are recycled by size (rounded up to 256 KiB) instead of being returned to the driver
static size_t pool_round(size_t bytes) { return (bytes + (size_t(1) << 18) - 1) & ~((size_t(1) << 18) - 1); }
void *fs_pool_take(fs_ctx *ctx, size_t bytes, size_t *got_bytes) {
    const size_t want = pool_round(bytes);
    int best = -1;
    for (int k = 0; k < (int)ctx->pool.size(); ++k)
        if (ctx->pool[k].bytes >= want && ctx->pool[k].bytes <= 2 * want && (best < 0 || ctx->pool[k].bytes < ctx->pool[best].bytes))
            best = k;
    if (best >= 0) {
        FsPoolBuf b = ctx->pool[best];
        ctx->pool.erase(ctx->pool.begin() + best);
        ctx->pool_bytes -= b.bytes;
        if (got_bytes) *got_bytes = b.bytes;
        return b.ptr;
    }
    void *p = nullptr;
    if (!fs_hip_ok(hipMalloc(&p, want), "hipMalloc(pool)")) return nullptr;
    if (got_bytes) *got_bytes = want;
    return p;
}
// Idle slabs beyond FS_POOL_IDLE_LIMIT go back to the driver, oldest first (sizes nobody asks for any more would otherwise
// pile up over a long run of differently sized tasks; hipFree waits for the device, so whatever still reads them has finished).
static const size_t FS_POOL_IDLE_LIMIT = [] {
    const char *v = getenv("FLINGSIM_POOL_IDLE_MB");  // (tests shrink it to see the trimming)
    const long long mb = v ? atoll(v) : 0;
    return mb > 0 ? size_t(mb) << 20 : size_t(4) << 30;
}();
extern "C" int fs_pool_stats(const fs_ctx *ctx, long long *out3) {
    if (!ctx || !out3) return FS_ERR_ARG;
    out3[0] = (long long)ctx->pool_bytes; out3[1] = (long long)ctx->pool.size(); out3[2] = (long long)FS_POOL_IDLE_LIMIT;
    return FS_OK;
}
void fs_pool_give(fs_ctx *ctx, void *ptr, size_t bytes) {
    if (!ptr) return;
    ctx->pool.push_back(FsPoolBuf{ptr, bytes});
    ctx->pool_bytes += bytes;
    while (ctx->pool_bytes > FS_POOL_IDLE_LIMIT && ctx->pool.size() > 1) {
        const FsPoolBuf b = ctx->pool.front();
        ctx->pool.erase(ctx->pool.begin());
        ctx->pool_bytes -= b.bytes;
        (void)hipFree(b.ptr);
    }
}

// every stream the context launches on: its own and the concurrent launch chains' (fs_step_stream).  The chains join the
// context's stream through events after a complete launch sequence; after an error in the middle of one they may not have.
void fs_sync_all_streams(fs_ctx *ctx) {
    if (ctx->main_stream) (void)hipStreamSynchronize(ctx->main_stream);
    if (ctx->svc_stream) (void)hipStreamSynchronize(ctx->svc_stream);
    for (int g = 0; g < FS_MAX_STREAM_GROUPS; ++g)
        if (ctx->aux_streams[g]) (void)hipStreamSynchronize(ctx->aux_streams[g]);
}
// What a call that is about to rewrite an episode has to wait for.  On the main lane: everything the solver may still be
// running.  On the service lane (fs_service_lane) the caller guarantees that the episode is not part of a chunk in flight
// -- that is the lane's contract -- so only the lane's own stream matters and the chunk keeps running.
void fs_sync_lane(fs_ctx *ctx) {
    if (ctx->on_svc) { (void)hipStreamSynchronize(ctx->svc_stream); return; }
    if (ctx->main_stream) (void)hipStreamSynchronize(ctx->main_stream);
    for (int g = 0; g < FS_MAX_STREAM_GROUPS; ++g)
        if (ctx->aux_streams[g]) (void)hipStreamSynchronize(ctx->aux_streams[g]);
}

fs_ctx::~fs_ctx() {
    (void)hipSetDevice(device);
    fs_sync_all_streams(this);
    for (auto &e : envs) {
        if (e.slab) fs_pool_give(this, e.slab, e.slab_bytes);
        if (e.d_picked) (void)hipFree(e.d_picked);
        if (e.d_saved_w) (void)hipFree(e.d_saved_w);
        if (e.d_snapshot) (void)hipFree(e.d_snapshot);
    }
    envs.clear();
    topo_cache.clear();
    for (auto &b : pool) (void)hipFree(b.ptr);  // (after the episodes and their topologies gave their slabs back)
    pool.clear();
    for (auto &t : tickets) {
        if (t.done) (void)hipEventDestroy(t.done);
        if (t.d_tab) (void)hipFree(t.d_tab);
        if (t.h_tab) (void)hipHostFree(t.h_tab);
        if (t.h_wait) (void)hipHostFree(t.h_wait);
    }
    if (d_wait) (void)hipFree(d_wait);
    if (svc_scratch) (void)hipFree(svc_scratch);
    if (svc_event) (void)hipEventDestroy(svc_event);
    if (d_envs) (void)hipFree(d_envs);
    if (d_shapes) (void)hipFree(d_shapes);
    if (d_ids) (void)hipFree(d_ids);
    if (d_slot_envs) (void)hipFree(d_slot_envs);
    if (d_slot_sweeps) (void)hipFree(d_slot_sweeps);
    if (h_ids) (void)hipHostFree(h_ids);
    if (h_stage) (void)hipHostFree(h_stage);
    if (render_scratch) (void)hipFree(render_scratch);
    if (loop_scratch) (void)hipFree(loop_scratch);
    if (d_coverage) (void)hipFree(d_coverage);
    for (int g = 0; g < FS_MAX_STREAM_GROUPS; ++g) {
        if (aux_streams[g]) { (void)hipStreamSynchronize(aux_streams[g]); (void)hipStreamDestroy(aux_streams[g]); }
        if (aux_events[g]) (void)hipEventDestroy(aux_events[g]);
    }
    if (fork_event) (void)hipEventDestroy(fork_event);
    if (adv_ev0) (void)hipEventDestroy(adv_ev0);
    if (adv_ev1) (void)hipEventDestroy(adv_ev1);
    if (ev_start) (void)hipEventDestroy(ev_start);
    if (ev_stop) (void)hipEventDestroy(ev_stop);
    if (svc_stream) (void)hipStreamDestroy(svc_stream);
    if (main_stream) (void)hipStreamDestroy(main_stream);
}

void *fs_stage(fs_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->h_stage_bytes) return ctx->h_stage;
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    ctx->h_stage_bytes = 0;
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes * 2;
    if (!fs_hip_ok(hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault), "hipHostMalloc(stage)")) return nullptr;
    ctx->h_stage_bytes = want;
    return ctx->h_stage;
}

// grow-only device scratch of the small reductions (fs_loops.hip); the calls run one after the other on one lane, so
// one buffer serves them all.  Growing it frees the old one: only then does a call of these wait for the whole device.
void *fs_svc_scratch(fs_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->svc_scratch_bytes) return ctx->svc_scratch;
    fs_sync_all_streams(ctx);
    if (ctx->svc_scratch) (void)hipFree(ctx->svc_scratch);
    ctx->svc_scratch = nullptr;
    ctx->svc_scratch_bytes = 0;
    const size_t want = bytes < (1u << 16) ? (1u << 16) : bytes * 2;
    if (!fs_hip_ok(hipMalloc(&ctx->svc_scratch, want), "hipMalloc(service scratch)")) return nullptr;
    ctx->svc_scratch_bytes = want;
    return ctx->svc_scratch;
}

// grow-only device scratch of the manipulation loops (fs_advance): hipMalloc / hipFree cost ~0.1-0.3 ms each and hipFree
// synchronises the device, which a call per chunk of a few simulation steps cannot afford
void *fs_loop_scratch(fs_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->loop_scratch_bytes) return ctx->loop_scratch;
    fs_sync_all_streams(ctx);
    if (ctx->loop_scratch) (void)hipFree(ctx->loop_scratch);
    ctx->loop_scratch = nullptr;
    ctx->loop_scratch_bytes = 0;
    const size_t want = bytes < (1u << 18) ? (1u << 18) : bytes * 2;
    if (!fs_hip_ok(hipMalloc(&ctx->loop_scratch, want), "hipMalloc(loop scratch)")) return nullptr;
    ctx->loop_scratch_bytes = want;
    return ctx->loop_scratch;
}

extern "C" fs_ctx *fs_create(int device, int n_envs, int camera_width, int camera_height) {
    if (n_envs < 1 || n_envs > (1 << 20)) { fs_set_error("n_envs out of range"); return nullptr; }
    int count = 0;
    if (!fs_hip_ok(hipGetDeviceCount(&count), "hipGetDeviceCount") || count <= 0) {
        if (count <= 0) fs_set_error("no HIP device visible: libflingsim has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= count) { fs_set_error("device index out of range"); return nullptr; }
    if (!fs_hip_ok(hipSetDevice(device), "hipSetDevice")) return nullptr;
    {   // The library holds gfx950 code objects only, and the solver's reciprocal square root is that chip's v_rsq_f32 as data
        // (oracle/v_rsq_f32_gfx950.npz): another architecture gets a clear refusal here instead of "no kernel image" later.
        hipDeviceProp_t prop;
        if (!fs_hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return nullptr;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            fs_set_error("device " + std::to_string(device) + " is " + prop.gcnArchName +
                         ": libflingsim is built for gfx950 (MI355X) only");
            return nullptr;
        }
    }
    fs_ctx *ctx = new fs_ctx();
    ctx->device = device;
    ctx->n_envs = n_envs;
    ctx->cam_width = camera_width > 0 ? camera_width : 720;
    ctx->cam_height = camera_height > 0 ? camera_height : 720;
    ctx->envs.resize(n_envs);
    for (auto &e : ctx->envs) {
        memset(&e.dev, 0, sizeof(e.dev));
        memset(&e.shapes, 0, sizeof(e.shapes));
        e.cam.width = ctx->cam_width;
        e.cam.height = ctx->cam_height;
    }
    bool ok = fs_hip_ok(hipStreamCreateWithFlags(&ctx->main_stream, hipStreamNonBlocking), "hipStreamCreate") &&
              fs_hip_ok(hipMalloc((void **)&ctx->d_wait, sizeof(FsWaitDev) * n_envs), "hipMalloc(wait states)") &&
              fs_hip_ok(hipMemset(ctx->d_wait, 0, sizeof(FsWaitDev) * n_envs), "hipMemset") &&
              fs_hip_ok(hipMalloc((void **)&ctx->d_envs, sizeof(FsEnvDev) * n_envs), "hipMalloc(envs)") &&
              fs_hip_ok(hipMalloc((void **)&ctx->d_shapes, sizeof(FsShapesDev) * n_envs), "hipMalloc(shapes)") &&
              fs_hip_ok(hipMalloc((void **)&ctx->d_ids, sizeof(int) * n_envs), "hipMalloc(ids)") &&
              fs_hip_ok(hipMalloc((void **)&ctx->d_slot_envs, sizeof(FsEnvDev) * n_envs), "hipMalloc(slot table)") &&
              fs_hip_ok(hipMalloc((void **)&ctx->d_slot_sweeps, sizeof(FsSlotSweeps) * n_envs), "hipMalloc(slot sweeps)") &&
              fs_hip_ok(hipMemset(ctx->d_slot_sweeps, 0, sizeof(FsSlotSweeps) * n_envs), "hipMemset") &&
              fs_hip_ok(hipMalloc((void **)&ctx->d_coverage, sizeof(double) * n_envs), "hipMalloc(cov)") &&
              fs_hip_ok(hipHostMalloc((void **)&ctx->h_ids, sizeof(int) * n_envs, hipHostMallocDefault), "hipHostMalloc") &&
              fs_hip_ok(hipMemset(ctx->d_envs, 0, sizeof(FsEnvDev) * n_envs), "hipMemset") &&
              fs_hip_ok(hipMemset(ctx->d_shapes, 0, sizeof(FsShapesDev) * n_envs), "hipMemset");
    ctx->stream = ctx->main_stream;
    if (!ok) {
        delete ctx;
        return nullptr;
    }
    return ctx;
}

// The service lane.  While chunks queued by fs_advance_begin run on the main stream, the host keeps working for the episodes
// that are NOT part of them -- reductions, observations, resets -- and those calls must neither queue up behind the chunk
// nor wait for it.  Between fs_service_lane(ctx, 1) and fs_service_lane(ctx, 0) every entry point of this library runs on a
// second, high-priority stream instead.  Contract: on the service lane the caller only touches episodes that are not in a
// chunk in flight (or are in it as loops that have already ended).  Leaving the lane makes the main stream wait for what
// the lane queued, so whatever is launched next sees it.
extern "C" int fs_service_lane(fs_ctx *ctx, int on) {
    if (!ctx) return FS_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (on && !ctx->on_svc) {
        if (!ctx->svc_stream) {
            int lo = 0, hi = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));  // (hi is the numerically lowest = most urgent)
            HIP_TRY(hipStreamCreateWithPriority(&ctx->svc_stream, hipStreamNonBlocking, hi));
            HIP_TRY(hipEventCreateWithFlags(&ctx->svc_event, hipEventDisableTiming));
        }
        ctx->stream = ctx->svc_stream;
        ctx->on_svc = true;
    } else if (!on && ctx->on_svc) {
        HIP_TRY(hipEventRecord(ctx->svc_event, ctx->svc_stream));
        HIP_TRY(hipStreamWaitEvent(ctx->main_stream, ctx->svc_event, 0));
        ctx->stream = ctx->main_stream;
        ctx->on_svc = false;
    }
    return FS_OK;
}

extern "C" void fs_destroy(fs_ctx *ctx) { delete ctx; }
extern "C" int fs_n_envs(const fs_ctx *ctx) { return ctx ? ctx->n_envs : FS_ERR_ARG; }
extern "C" int fs_set_solver(fs_ctx *ctx, int solver) {
    if (!ctx || solver < 0 || solver > FS_SOLVER_COTENANT) { fs_set_error("bad solver id"); return FS_ERR_ARG; }
    ctx->force_merged_boundary = (solver == FS_SOLVER_STREAM_MERGED);
    ctx->force_coded_stream = (solver == FS_SOLVER_STREAM_CODED);
    ctx->force_split_boundary = (solver == FS_SOLVER_STREAM_SPLIT);
    ctx->force_generic_fused = (solver == FS_SOLVER_FUSED_GENERIC);
    ctx->force_coded_fused = (solver == FS_SOLVER_FUSED_CODED);
    ctx->force_ell_stream = (solver == FS_SOLVER_STREAM_ELL);
    ctx->solver = (solver == FS_SOLVER_FUSED_GENERIC || solver == FS_SOLVER_FUSED_CODED)
                      ? FS_SOLVER_FUSED
                      : ((solver == FS_SOLVER_STREAM_ELL || solver == FS_SOLVER_STREAM_CODED || solver == FS_SOLVER_STREAM_SPLIT ||
                         solver == FS_SOLVER_STREAM_MERGED)
                             ? FS_SOLVER_STREAM
                             : solver);
    return FS_OK;
}
extern "C" int fs_set_stream_groups(fs_ctx *ctx, int groups) {
    if (!ctx || groups < 0 || groups > FS_MAX_STREAM_GROUPS) { fs_set_error("fs_set_stream_groups: 0 (default) .. 4"); return FS_ERR_ARG; }
    ctx->stream_groups = groups;
    return FS_OK;
}
extern "C" int fs_last_stream_groups(const fs_ctx *ctx) { return ctx ? ctx->last_stream_groups : FS_ERR_ARG; }
extern "C" int fs_get_solver(const fs_ctx *ctx) { return ctx ? ctx->solver : FS_ERR_ARG; }
extern "C" int fs_last_kernel_form(const fs_ctx *ctx) { return ctx ? ctx->last_form : FS_ERR_ARG; }
extern "C" int fs_last_boundary_form(const fs_ctx *ctx) { return ctx ? ctx->last_boundary : FS_ERR_ARG; }
extern "C" void *fs_stream(fs_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

static FsEnv *get_env(fs_ctx *ctx, int env, bool need_scene = true) {
    if (!ctx) { fs_set_error("null context"); return nullptr; }
    if (env < 0 || env >= ctx->n_envs) { fs_set_error("env index out of range"); return nullptr; }
    FsEnv *e = &ctx->envs[env];
    if (need_scene && !e->has_scene) { fs_set_error("env has no scene: call fs_set_scene first"); return nullptr; }
    return e;
}

extern "C" int fs_device_key(fs_ctx *ctx, char *out, int n_chars) {
    if (!ctx || !out || n_chars < 16) return FS_ERR_ARG;
    HIP_TRY(hipDeviceGetPCIBusId(out, n_chars, ctx->device));   // e.g. "0000:05:00.0": the same for every process, whatever
    return FS_OK;                                              // HIP_VISIBLE_DEVICES made of the device's index
}

extern "C" int fs_fused_fits(fs_ctx *ctx, int env) {
    FsEnv *e = get_env(ctx, env);
    if (!e) return FS_ERR_ARG;
    return fs_fused_supported(ctx, *e) ? 1 : 0;
}

int fs_lane_guard(fs_ctx *ctx, int env) {
    if (!ctx || !ctx->on_svc || ctx->tickets_busy == 0) return FS_OK;
    for (const FsAdvTicket &t : ctx->tickets) {
        if (!t.busy) continue;
        for (int e : t.listed) {
            if (e != env) continue;
            // An entry that continues a wait / step loop (start = -1) which the host already knows to be over is dead: the
            // check kernel retires it before anything of the episode is read (fs_k_wait_check), so the episode is free.
            bool dead = false;
            for (size_t q = 0; q < t.w_env.size(); ++q)
                if (t.w_env[q] == env && t.w_start[q] < 0 && env < (int)ctx->wait_over.size() && ctx->wait_over[env]) dead = true;
            if (dead) continue;
            fs_set_error("service lane: episode " + std::to_string(env) + " is part of an fs_advance chunk in flight (fs_advance_end first)");
            return FS_ERR_STATE;
        }
    }
    return FS_OK;
}
// Stepping on the service lane while a chunk is in flight is refused outright: a launch sequence rebuilds the context's ONE
// slot table / sweep table and reuses its chain streams, fork / join events and loop scratch, which the chunk still running
// on the main stream reads -- whichever episodes are listed.  (The services the lane exists for -- reductions, observations,
// resets -- never step.)
int fs_step_guard(fs_ctx *ctx, const char *who) {
    if (!ctx || !ctx->on_svc || ctx->tickets_busy == 0) return FS_OK;
    fs_set_error(std::string(who) + ": the simulation cannot be stepped on the service lane while an fs_advance chunk is in flight "
                                    "(fs_advance_end / fs_service_lane(ctx, 0) first)");
    return FS_ERR_STATE;
}
#define LANE_GUARD(ctx, env)                                   \
    do {                                                       \
        const int guard_rc = fs_lane_guard((ctx), (env));      \
        if (guard_rc != FS_OK) return guard_rc;                \
    } while (0)

static uint64_t fnv(uint64_t h, const void *data, size_t bytes) {
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

static size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

// carve helper
struct Carver {
    char *base;
    size_t off = 0;
    template <typename T> T *take(size_t count) {
        T *p = base ? (T *)(base + off) : nullptr;
        off += align_up(sizeof(T) * count);
        return p;
    }
};

static std::shared_ptr<FsTopologyDev> make_topology(fs_ctx *ctx, const FsHostScene &s) {
    uint64_t key = 1469598103934665603ull;
    key = fnv(key, s.pos.data(), s.pos.size() * 4);
    key = fnv(key, s.springs.data(), s.springs.size() * 4);
    key = fnv(key, s.spring_len.data(), s.spring_len.size() * 4);
    key = fnv(key, s.spring_k.data(), s.spring_k.size() * 4);
    key = fnv(key, s.tris.data(), s.tris.size() * 4);
    for (auto it = ctx->topo_cache.begin(); it != ctx->topo_cache.end();) {
        auto sp = it->lock();
        if (!sp) { it = ctx->topo_cache.erase(it); continue; }
        if (sp->key == key && sp->n == s.n && sp->m == s.m && sp->t == s.t) return sp;
        ++it;
    }
    auto topo = std::make_shared<FsTopologyDev>();
    topo->key = key; topo->n = s.n; topo->m = s.m; topo->t = s.t; topo->max_deg = s.max_deg;
    const size_t n = s.n, m2 = size_t(2) * s.m, ell = size_t(s.max_deg) * s.n;
    auto carve = [&](char *base) {
        Carver c{base};
        topo->rest = c.take<FsVec4>(n);
        topo->adj_off = c.take<int>(n + 1);
        topo->adj_j = c.take<int>(m2 + 1);
        topo->adj_len = c.take<float>(m2 + 1);
        topo->adj_k = c.take<float>(m2 + 1);
        topo->ell_j = c.take<int>(ell + 1);
        topo->ell_len = c.take<float>(ell + 1);
        topo->ell_k = c.take<float>(ell + 1);
        topo->dict = c.take<float>(512);
        topo->code_w = c.take<uint32_t>(size_t(8) * n + 1);
        topo->nbr_w = c.take<uint32_t>(size_t(8) * n + 1);
        topo->restnear_w = c.take<uint32_t>(size_t(8) * n + 1);
        topo->sdict = c.take<FsVec4>(256);
        topo->scode = c.take<FsU32x4>(n + 1);
        topo->g64_L = c.take<float>(s.gp_L_ok ? size_t(FS_G64_SLOTS) * n : 1);
        topo->tris = c.take<int>(size_t(3) * s.t + 1);
        topo->vt_off = c.take<int>(n + 1);
        topo->vt_tri = c.take<int>(size_t(3) * s.t + 1);
        return c.off;
    };
    const size_t image_bytes = carve(nullptr);
    topo->owner = ctx;
    topo->slab = fs_pool_take(ctx, image_bytes, &topo->bytes);
    if (!topo->slab) return nullptr;
    // the image of the slab is assembled in the pinned staging buffer with the same carve and goes up in ONE copy (the
    // separate arrays used to be ~17 blocking copies from pageable memory)
    char *stage = (char *)fs_stage(ctx, image_bytes);
    if (!stage) return nullptr;
    bool ok = fs_hip_ok(hipStreamSynchronize(ctx->stream), "sync before staging");
    carve(stage);
    auto up = [&](void *dst, const void *src, size_t bytes) {
        if (bytes) memcpy(dst, src, bytes);
    };
    up(topo->rest, s.pos.data(), n * 16);
    up(topo->adj_off, s.adj_off.data(), (n + 1) * 4);
    up(topo->adj_j, s.adj_j.data(), m2 * 4);
    up(topo->adj_len, s.adj_len.data(), m2 * 4);
    up(topo->adj_k, s.adj_k.data(), m2 * 4);
    up(topo->ell_j, s.ell_j.data(), ell * 4);
    up(topo->ell_len, s.ell_len.data(), ell * 4);
    up(topo->ell_k, s.ell_k.data(), ell * 4);
    topo->dict_size = s.dict_size;
    up(topo->dict, s.dict.data(), 512 * 4);
    if (s.dict_size > 0) {
        up(topo->code_w, s.code_w.data(), size_t(8) * n * 4);
        up(topo->nbr_w, s.nbr_w.data(), size_t(8) * n * 4);
    }
    topo->sdict_size = s.sdict_size;
    up(topo->sdict, s.sdict.data(), 1024 * 4);
    if (s.sdict_size > 0) up(topo->scode, s.scode.data(), n * 16);
    topo->restnear_ok = s.restnear_ok;
    up(topo->restnear_w, s.restnear_w.data(), size_t(8) * n * 4);
    topo->g64_ok = s.g64_ok;
    topo->gp_L_ok = s.gp_L_ok;
    topo->gp_magic = s.gp_magic;
    topo->gp_halvable = s.gp_halvable;
    if (s.gp_L_ok) up(topo->g64_L, s.g64_L.data(), size_t(FS_G64_SLOTS) * n * 4);
    for (int q = 0; q < FS_G64_SLOTS; ++q) topo->g64_k[q] = s.g64_k[q];
    up(topo->tris, s.tris.data(), size_t(3) * s.t * 4);
    up(topo->vt_off, s.vt_off.data(), (n + 1) * 4);
    up(topo->vt_tri, s.vt_tri.data(), size_t(3) * s.t * 4);
    carve((char *)topo->slab);  // the members now point into the device slab
    ok = ok && fs_hip_ok(hipMemcpyAsync(topo->slab, stage, image_bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(topology)");
    ok = ok && fs_hip_ok(hipStreamSynchronize(ctx->stream), "hipMemcpy(topology) sync");
    if (!ok) return nullptr;
    ctx->topo_cache.push_back(topo);
    return topo;
}

// summary of an episode's phases for the streaming neighbour search (FsEnvDev::find_mode)
static int phase_find_mode(const int *phase, int n, int restnear_ok) {
    if (n <= 0) return 0;
    for (int i = 1; i < n; ++i)
        if (phase[i] != phase[0]) return 0;
    if (!(phase[0] & FS_PHASE_SELF_COLLIDE)) return 3;
    if (!(phase[0] & FS_PHASE_SELF_COLLIDE_FILTER)) return 2;
    return (restnear_ok == 2 && FS_STENCIL_FILTER) ? 4 : (restnear_ok ? 1 : 0);
}

static int push_env_desc(fs_ctx *ctx, int env) {
    ctx->desc_epoch++;  // the streaming back-end's launch table (fs_k_slot_table) is stale now
    HIP_TRY(hipMemcpyAsync(ctx->d_envs + env, &ctx->envs[env].dev, sizeof(FsEnvDev), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FS_OK;
}
static int push_shapes(fs_ctx *ctx, int env) {
    ctx->desc_epoch++;  // the launch table carries the spheres' per-substep sweeps (fs_k_slot_table): stale now
    HIP_TRY(hipMemcpyAsync(ctx->d_shapes + env, &ctx->envs[env].shapes, sizeof(FsShapesDev), hipMemcpyHostToDevice,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FS_OK;
}

// Init for episode `env` from a built host scene (main.cpp:613-1122): tears down the previous solver state, buffers and
// shapes, allocates the episode's slab (from the context's pool), uploads the initial state.
static int set_scene_impl(fs_ctx *ctx, int env, FsHostScene &&scene) {
    FsEnv *e = get_env(ctx, env, false);
    if (!e) return FS_ERR_ARG;
    LANE_GUARD(ctx, env);
    fs_sync_lane(ctx);  // nothing may still run on the episode's old slab (fs_sync_lane: what that means on either lane)
    if (e->slab) { fs_pool_give(ctx, e->slab, e->slab_bytes); e->slab = nullptr; e->slab_bytes = 0; }
    e->picker_ready = false;
    e->has_scene = false;
    e->topo.reset();
    auto topo = make_topology(ctx, scene);
    if (!topo) return FS_ERR_HIP;

    const size_t n = scene.n;
    if (e->d_saved_w && e->saved_w_n < (int)n) { (void)hipFree(e->d_saved_w); e->d_saved_w = nullptr; e->saved_w_n = 0; }  // (grow-only)
    FsEnvDev d;
    memset(&d, 0, sizeof(d));
    auto carve = [&](char *base) {
        Carver c{base};
        d.pos = c.take<FsVec4>(n);
        d.vel = c.take<FsVec4>(n);
        d.phase = c.take<int>(n);
        d.x0 = c.take<FsVec4>(n);
        d.v0 = c.take<FsVec4>(n);
        d.xa = c.take<FsVec4>(n);
        d.xb = c.take<FsVec4>(n);
        d.ncount = c.take<int>(n);
        d.nlist = c.take<int>(n * FS_MAX_NEIGHBORS);
        d.cell_count = c.take<int>(FS_GRID_BUCKETS + 1);
        d.cell_fill = c.take<int>(FS_GRID_BUCKETS + 1);
        d.cell_items = c.take<int>(n);
        return c.off;
    };
    const size_t bytes = carve(nullptr);
    size_t slab_bytes = 0;
    void *slab = fs_pool_take(ctx, bytes, &slab_bytes);
    if (!slab) return FS_ERR_HIP;
    carve((char *)slab);
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemsetAsync(slab, 0, bytes, st));
    d.n = scene.n; d.m = scene.m; d.max_deg = scene.max_deg; d.has_scene = 1;
    d.rest = topo->rest; d.adj_off = topo->adj_off; d.adj_j = topo->adj_j; d.adj_len = topo->adj_len; d.adj_k = topo->adj_k;
    d.ell_j = topo->ell_j; d.ell_len = topo->ell_len; d.ell_k = topo->ell_k;
    d.restnear_w = topo->restnear_w; d.restnear_ok = topo->restnear_ok;
    d.find_mode = phase_find_mode(scene.phase.data(), scene.n, topo->restnear_ok);
    d.dict_size = topo->dict_size; d.dict = topo->dict; d.code_w = topo->code_w; d.nbr_w = topo->nbr_w;
    d.sdict = topo->sdict; d.scode = topo->scode; d.sdict_size = topo->sdict_size; d.pad1 = 0;
    d.gp_count = scene.sdict_size > 0 ? scene.gp_count : 0;  // the pattern is used together with the spring codes
    d.gp_dimx = scene.gp_dimx; d.gp_dimz = scene.gp_dimz; d.gp_pad = 0;
    for (int q = 0; q < 16; ++q) { d.gp_dx[q] = scene.gp_dx[q]; d.gp_dz[q] = scene.gp_dz[q]; }
    d.g64_ok = topo->g64_ok; d.g64_L = topo->g64_L;
    d.gp_L_ok = topo->gp_L_ok; d.gp_magic = topo->gp_magic; d.gp_halvable = topo->gp_halvable;
    for (int q = 0; q < FS_G64_SLOTS; ++q) { d.g64_kh[q] = topo->g64_k[q] * 0.5f; d.gp_k[q] = topo->g64_k[q]; }
    d.p = scene.params;

    // uploads (main.cpp:1025-1085): positions and phases (velocities are the zeros of the memset) through the pinned staging
    // buffer, the two descriptors behind them; ONE wait at the end
    char *stage = (char *)fs_stage(ctx, n * 20);
    if (!stage) return FS_ERR_HIP;
    memcpy(stage, scene.pos.data(), n * 16);
    memcpy(stage + n * 16, scene.phase.data(), n * 4);
    HIP_TRY(hipMemcpyAsync(d.pos, stage, n * 16, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d.phase, stage + n * 16, n * 4, hipMemcpyHostToDevice, st));

    e->slab = slab; e->slab_bytes = slab_bytes; e->dev = d; e->topo = topo;
    e->host = std::move(scene);
    memset(&e->shapes, 0, sizeof(e->shapes));  // shapes wiped by Init (main.cpp:701-706)
    // CenterCamera (softgym_cloth.h:177-183)
    for (int k = 0; k < 3; ++k) { e->cam.pos[k] = e->host.cam_pos[k]; e->cam.angle[k] = e->host.cam_angle[k]; }
    e->cam.width = e->host.cam_width > 0 ? e->host.cam_width : ctx->cam_width;
    e->cam.height = e->host.cam_height > 0 ? e->host.cam_height : ctx->cam_height;
    e->has_scene = true;
    ctx->desc_epoch++;  // the streaming back-end's launch table (fs_k_slot_table) is stale now
    HIP_TRY(hipMemcpyAsync(ctx->d_envs + env, &e->dev, sizeof(FsEnvDev), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx->d_shapes + env, &e->shapes, sizeof(FsShapesDev), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return FS_OK;
}

extern "C" int fs_set_scene(fs_ctx *ctx, int env, const float *scene_params, int n_params, const float *verts,
                            int n_vert_floats, const int *stretch, int n_stretch_ints, const int *bend, int n_bend_ints,
                            const int *shear, int n_shear_ints, const int *faces, int n_face_ints) {
    if (!get_env(ctx, env, false)) return FS_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    FsHostScene scene;
    std::string err = fs_build_scene(scene, scene_params, n_params, verts, n_vert_floats, stretch, n_stretch_ints, bend,
                                     n_bend_ints, shear, n_shear_ints, faces, n_face_ints);
    if (!err.empty()) { fs_set_error(err); return FS_ERR_ARG; }
    return set_scene_impl(ctx, env, std::move(scene));
}

// fs_set_scene from a scene that fs_host_scene_build (host only: any thread, no GPU) made earlier -- the evaluation loop
// builds the NEXT tasks' scenes on a worker thread while the current ones simulate.  Takes the scene out of `scene`
// (which stays valid but empty; free it with fs_host_scene_free as usual).
struct fs_host_scene { FsHostScene s; };
extern "C" int fs_set_scene_prebuilt(fs_ctx *ctx, int env, fs_host_scene *scene) {
    if (!get_env(ctx, env, false)) return FS_ERR_ARG;
    if (!scene || scene->s.n <= 0) { fs_set_error("fs_set_scene_prebuilt: no scene (already consumed?)"); return FS_ERR_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    FsHostScene moved = std::move(scene->s);
    scene->s = FsHostScene();
    return set_scene_impl(ctx, env, std::move(moved));
}

extern "C" int fs_step(fs_ctx *ctx, int env, int n_steps) {
    if (!ctx) { fs_set_error("null context"); return FS_ERR_ARG; }
    if (n_steps < 0) { fs_set_error("n_steps < 0"); return FS_ERR_ARG; }
    if (const int guard_rc = fs_step_guard(ctx, "fs_step")) return guard_rc;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<int> ids;
    if (env == -1) {
        for (int i = 0; i < ctx->n_envs; ++i)
            if (ctx->envs[i].has_scene) ids.push_back(i);
    } else {
        if (!get_env(ctx, env)) return FS_ERR_ARG;
        ids.push_back(env);
    }
    if (ids.empty() || n_steps == 0) return FS_OK;
    return fs_step_ids(ctx, ids, n_steps, nullptr);
}

extern "C" int fs_step_list(fs_ctx *ctx, int n, const int *envs, int n_steps) {
    if (!ctx || !envs || n <= 0) { fs_set_error("fs_step_list: bad arguments"); return FS_ERR_ARG; }
    if (n_steps < 0) { fs_set_error("n_steps < 0"); return FS_ERR_ARG; }
    if (const int guard_rc = fs_step_guard(ctx, "fs_step_list")) return guard_rc;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<int> ids(envs, envs + n);
    for (int id : ids)
        if (!get_env(ctx, id)) return FS_ERR_ARG;
    if (n_steps == 0) return FS_OK;
    return fs_step_ids(ctx, ids, n_steps, nullptr);
}

int fs_step_ids(fs_ctx *ctx, const std::vector<int> &ids, int n_steps, const int *d_ids) {
    int solver = ctx->solver;
    if (solver == FS_SOLVER_COTENANT) {
        // a process that shares the device (include/flingsim.h): the fused kernel whenever the launch fits it, else as AUTO
        solver = FS_SOLVER_FUSED;
        for (int id : ids)
            if (!fs_fused_supported(ctx, ctx->envs[id])) solver = FS_SOLVER_AUTO;
    }
    if (solver == FS_SOLVER_AUTO) {
        // The fused kernel gives one CU to an episode for the whole frame: unbeatable once the launch fills the chip, but a
        // small launch leaves most CUs idle while the streaming kernels spread every stage over all of them.  Measured
        // crossover on the crumpled 64x64 bench scenario (scripts/solver_crossover.py): between 64 and 128 episodes.
        solver = FS_SOLVER_FUSED;
        size_t particles = 0;
        bool grid64 = true;
        for (int id : ids) {
            if (!fs_fused_supported(ctx, ctx->envs[id])) solver = FS_SOLVER_STREAM;
            particles += (size_t)ctx->envs[id].host.n;
            grid64 = grid64 && ctx->envs[id].dev.g64_ok;
        }
        // measured crossover on the crumpled 64x64 bench scenario (scripts/solver_crossover.py; round 4, hardware reciprocal
        // root): streaming 1.46 / 1.73 / 1.95 / 2.15 / 2.31 / 2.61 / 3.62 ms per step at 96 / 128 / 144 / 160 / 176 / 192 / 256
        // episodes against a flat 2.04-2.16 ms of the grid-64 fused kernel (~158 episodes) and 2.46-2.63 ms of the
        // dictionary-coded fused kernel (~190)
        // round 6, with the throughput form of the grid-L iteration (fs_k_iterate_gridl_tp) in the streaming back-end: streaming 1.70 /
        // 1.88 / 1.99 / 2.14 / 2.34 / 2.51 / 2.65 ms per step at 128 / 144 / 160 / 176 / 192 / 208 / 224 episodes against 2.10-2.16 ms
        // (grid-64) and 2.51-2.64 ms (dictionary-coded): the crossovers moved from ~158 / ~190 to ~176 / ~224 episodes
        if (particles < (size_t)(grid64 ? 176 : 224) * 4096) solver = FS_SOLVER_STREAM;
    } else if (solver == FS_SOLVER_FUSED) {
        for (int id : ids)
            if (!fs_fused_supported(ctx, ctx->envs[id])) {
                fs_set_error("FS_SOLVER_FUSED: episode does not fit the LDS-resident kernel");
                return FS_ERR_STATE;
            }
    }
    return solver == FS_SOLVER_FUSED ? fs_step_fused(ctx, ids, n_steps, d_ids) : fs_step_stream(ctx, ids, n_steps, d_ids);
}

extern "C" int fs_step_timed(fs_ctx *ctx, int env, int n_steps, float *elapsed_ms) {
    if (!ctx || !elapsed_ms) { fs_set_error("null argument"); return FS_ERR_ARG; }
    if (const int guard_rc = fs_step_guard(ctx, "fs_step_timed")) return guard_rc;
    HIP_TRY(hipSetDevice(ctx->device));
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    // ids upload etc. happen before the start event is reached on the stream only if we flush first
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipEventRecord(a, ctx->stream));
    int rc = fs_step(ctx, env, n_steps);
    HIP_TRY(hipEventRecord(b, ctx->stream));
    HIP_TRY(hipEventSynchronize(b));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return rc;
}

// HIP-event stopwatch on the context's stream: start records an event, stop records a second one, waits for it and
// returns the device time between them.
extern "C" int fs_timer_start(fs_ctx *ctx) {
    if (!ctx) return FS_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->ev_start) {
        HIP_TRY(hipEventCreate(&ctx->ev_start));
        HIP_TRY(hipEventCreate(&ctx->ev_stop));
    }
    HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    return FS_OK;
}
extern "C" int fs_timer_stop(fs_ctx *ctx, float *elapsed_ms) {
    if (!ctx || !elapsed_ms || !ctx->ev_start) { fs_set_error("fs_timer_stop without fs_timer_start"); return FS_ERR_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev_stop));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
    return FS_OK;
}

extern "C" int fs_sync(fs_ctx *ctx) {
    if (!ctx) return FS_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FS_OK;
}

extern "C" int fs_n_particles(const fs_ctx *ctx, int env) {
    FsEnv *e = get_env((fs_ctx *)ctx, env);
    return e ? e->host.n : FS_ERR_ARG;
}
extern "C" int fs_n_springs(const fs_ctx *ctx, int env) {
    FsEnv *e = get_env((fs_ctx *)ctx, env);
    return e ? e->host.m : FS_ERR_ARG;
}
extern "C" int fs_n_triangles(const fs_ctx *ctx, int env) {
    FsEnv *e = get_env((fs_ctx *)ctx, env);
    return e ? e->host.t : FS_ERR_ARG;
}
extern "C" int fs_n_shapes(const fs_ctx *ctx, int env) {
    FsEnv *e = get_env((fs_ctx *)ctx, env, false);
    return e ? e->shapes.count : FS_ERR_ARG;
}

// ---- device <-> host accessors through the pinned staging buffer
static int d2h(fs_ctx *ctx, void *dst, const void *src, size_t bytes) {
    void *st = fs_stage(ctx, bytes);
    if (!st) return FS_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(st, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(dst, st, bytes);
    return FS_OK;
}
static int h2d(fs_ctx *ctx, void *dst, const void *src, size_t bytes) {
    void *st = fs_stage(ctx, bytes);
    if (!st) return FS_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // staging buffer may still be in flight
    memcpy(st, src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FS_OK;
}
#define CHECK_LEN(have, need)                                                      \
    do {                                                                           \
        if ((have) < (need)) { fs_set_error("buffer too small"); return FS_ERR_ARG; } \
    } while (0)

extern "C" int fs_get_positions(fs_ctx *ctx, int env, float *out, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    CHECK_LEN(n_floats, 4 * e->host.n);
    HIP_TRY(hipSetDevice(ctx->device));
    return d2h(ctx, out, e->dev.pos, size_t(16) * e->host.n);
}
extern "C" int fs_set_positions(fs_ctx *ctx, int env, const float *in, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !in) return FS_ERR_ARG;
    LANE_GUARD(ctx, env);
    CHECK_LEN(n_floats, 4 * e->host.n);
    HIP_TRY(hipSetDevice(ctx->device));
    return h2d(ctx, e->dev.pos, in, size_t(16) * e->host.n);
}
extern "C" int fs_get_velocities(fs_ctx *ctx, int env, float *out, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    const int n = e->host.n;
    CHECK_LEN(n_floats, 3 * n);
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<float> tmp(size_t(4) * n);
    int rc = d2h(ctx, tmp.data(), e->dev.vel, size_t(16) * n);
    if (rc != FS_OK) return rc;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) out[3 * i + k] = tmp[4 * size_t(i) + k];
    return FS_OK;
}
extern "C" int fs_set_velocities(fs_ctx *ctx, int env, const float *in, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !in) return FS_ERR_ARG;
    LANE_GUARD(ctx, env);
    const int n = e->host.n;
    CHECK_LEN(n_floats, 3 * n);
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<float> tmp(size_t(4) * n, 0.0f);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) tmp[4 * size_t(i) + k] = in[3 * i + k];
    return h2d(ctx, e->dev.vel, tmp.data(), size_t(16) * n);
}
extern "C" int fs_get_phases(fs_ctx *ctx, int env, int *out, int n_ints) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    CHECK_LEN(n_ints, e->host.n);
    HIP_TRY(hipSetDevice(ctx->device));
    return d2h(ctx, out, e->dev.phase, size_t(4) * e->host.n);
}
extern "C" int fs_set_phases(fs_ctx *ctx, int env, const int *in, int n_ints) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !in) return FS_ERR_ARG;
    LANE_GUARD(ctx, env);
    CHECK_LEN(n_ints, e->host.n);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = h2d(ctx, e->dev.phase, in, size_t(4) * e->host.n);
    if (rc != FS_OK) return rc;
    const int mode = phase_find_mode(in, e->host.n, e->dev.restnear_ok);
    if (mode != e->dev.find_mode) {
        e->dev.find_mode = mode;
        rc = push_env_desc(ctx, env);
    }
    return rc;
}
extern "C" int fs_get_rest_positions(fs_ctx *ctx, int env, float *out, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    CHECK_LEN(n_floats, 4 * e->host.n);
    memcpy(out, e->host.pos.data(), size_t(16) * e->host.n);
    return FS_OK;
}
extern "C" int fs_get_normals(fs_ctx *ctx, int env, float *out, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    CHECK_LEN(n_floats, 4 * e->host.n);
    HIP_TRY(hipSetDevice(ctx->device));
    return fs_normals_env(ctx, env, out);
}
extern "C" int fs_get_edges(fs_ctx *ctx, int env, int *out, int n_ints) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    CHECK_LEN(n_ints, 2 * e->host.m);
    if (e->host.m > 0) memcpy(out, e->host.springs.data(), size_t(8) * e->host.m);  // (a 1 x 1 cloth has none: no null source)
    return FS_OK;
}
extern "C" int fs_get_faces(fs_ctx *ctx, int env, int *out, int n_ints) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    CHECK_LEN(n_ints, 3 * e->host.t);
    if (e->host.t > 0) memcpy(out, e->host.tris.data(), size_t(12) * e->host.t);  // (a 1 x 1 cloth has none: no null source)
    return FS_OK;
}
extern "C" int fs_get_spring_lengths(fs_ctx *ctx, int env, float *out, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    CHECK_LEN(n_floats, e->host.m);
    if (e->host.m > 0) memcpy(out, e->host.spring_len.data(), size_t(4) * e->host.m);  // (a 1 x 1 cloth has none: no null source)
    return FS_OK;
}
extern "C" int fs_get_spring_stiffness(fs_ctx *ctx, int env, float *out, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !out) return FS_ERR_ARG;
    CHECK_LEN(n_floats, e->host.m);
    if (e->host.m > 0) memcpy(out, e->host.spring_k.data(), size_t(4) * e->host.m);  // (a 1 x 1 cloth has none: no null source)
    return FS_OK;
}
extern "C" int fs_get_params(fs_ctx *ctx, int env, float *o, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !o) return FS_ERR_ARG;
    CHECK_LEN(n_floats, 32);
    const FsParams &p = e->dev.p;
    memset(o, 0, sizeof(float) * 32);
    o[0] = (float)p.numIterations; o[1] = (float)p.numSubsteps; o[2] = p.dt;
    o[3] = p.gravity[0]; o[4] = p.gravity[1]; o[5] = p.gravity[2];
    o[6] = p.radius; o[7] = p.solidRestDistance; o[8] = p.collisionDistance; o[9] = p.shapeCollisionMargin;
    o[10] = p.particleCollisionMargin; o[11] = p.dynamicFriction; o[12] = p.staticFriction; o[13] = p.particleFriction;
    o[14] = p.damping; o[15] = p.sleepThreshold; o[16] = p.relaxationFactor; o[17] = p.maxAcceleration;
    o[18] = p.maxSpeed; o[19] = p.restitution; o[20] = p.adhesion; o[21] = p.dissipation;
    o[22] = (float)p.numPlanes; o[23] = p.planes[0][0]; o[24] = p.planes[0][1]; o[25] = p.planes[0][2];
    o[26] = p.planes[0][3]; o[27] = (float)p.maxNeighbors; o[28] = (float)p.maxContacts; o[29] = (float)p.relaxationMode;
    return FS_OK;
}
extern "C" int fs_set_params(fs_ctx *ctx, int env, const float *o, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !o) return FS_ERR_ARG;
    LANE_GUARD(ctx, env);
    CHECK_LEN(n_floats, 32);
    HIP_TRY(hipSetDevice(ctx->device));
    FsParams &p = e->dev.p;
    if (o[0] < 0 || o[0] > 1000 || o[1] < 1 || o[1] > 100 || !(o[2] > 0.0f)) { fs_set_error("bad iteration / substep / dt values"); return FS_ERR_ARG; }
    if (o[6] != p.radius || o[10] != p.particleCollisionMargin) {
        fs_set_error("fs_set_params: radius / particleCollisionMargin are baked into the topology (rest-pose filter)");
        return FS_ERR_ARG;
    }
    p.numIterations = (int)o[0]; p.numSubsteps = (int)o[1]; p.dt = o[2];
    p.gravity[0] = o[3]; p.gravity[1] = o[4]; p.gravity[2] = o[5];
    p.radius = o[6]; p.solidRestDistance = o[7]; p.collisionDistance = o[8]; p.shapeCollisionMargin = o[9];
    p.particleCollisionMargin = o[10]; p.dynamicFriction = o[11]; p.staticFriction = o[12]; p.particleFriction = o[13];
    p.damping = o[14]; p.sleepThreshold = o[15]; p.relaxationFactor = o[16]; p.maxAcceleration = o[17];
    p.maxSpeed = o[18]; p.restitution = o[19]; p.adhesion = o[20]; p.dissipation = o[21];
    p.planes[0][0] = o[23]; p.planes[0][1] = o[24]; p.planes[0][2] = o[25]; p.planes[0][3] = o[26];
    e->host.params = p;
    return push_env_desc(ctx, env);
}
extern "C" int fs_get_scene_bounds(fs_ctx *ctx, int env, float *lower3, float *upper3) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !lower3 || !upper3) return FS_ERR_ARG;
    memcpy(lower3, e->host.scene_lower, 12);
    memcpy(upper3, e->host.scene_upper, 12);
    return FS_OK;
}

// ---- shapes
extern "C" int fs_add_sphere(fs_ctx *ctx, int env, float radius, const float *pos3, const float *quat4) {
    FsEnv *e = get_env(ctx, env, false);
    if (!e || !pos3 || !quat4) return FS_ERR_ARG;
    LANE_GUARD(ctx, env);
    if (e->shapes.count >= FS_MAX_SHAPES) { fs_set_error("too many shapes"); return FS_ERR_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    int q = e->shapes.count++;
    e->shapes.pos[q] = FsVec4{pos3[0], pos3[1], pos3[2], radius};
    e->shapes.prev[q] = FsVec4{pos3[0], pos3[1], pos3[2], radius};  // prev := current (helpers.h:493-494)
    for (int k = 0; k < 4; ++k) { e->shape_rot[q][k] = quat4[k]; e->shape_prev_rot[q][k] = quat4[k]; }
    return push_shapes(ctx, env);
}
extern "C" int fs_clear_shapes(fs_ctx *ctx, int env) {
    FsEnv *e = get_env(ctx, env, false);
    if (!e) return FS_ERR_ARG;
    LANE_GUARD(ctx, env);
    HIP_TRY(hipSetDevice(ctx->device));
    e->shapes.count = 0;
    return push_shapes(ctx, env);
}
extern "C" int fs_get_shape_states(fs_ctx *ctx, int env, float *out, int n_floats) {
    FsEnv *e = get_env(ctx, env, false);
    if (!e) return FS_ERR_ARG;
    CHECK_LEN(n_floats, 14 * e->shapes.count);
    for (int q = 0; q < e->shapes.count; ++q) {
        float *o = out + 14 * q;
        o[0] = e->shapes.pos[q].x; o[1] = e->shapes.pos[q].y; o[2] = e->shapes.pos[q].z;
        o[3] = e->shapes.prev[q].x; o[4] = e->shapes.prev[q].y; o[5] = e->shapes.prev[q].z;
        for (int k = 0; k < 4; ++k) { o[6 + k] = e->shape_rot[q][k]; o[10 + k] = e->shape_prev_rot[q][k]; }
    }
    return FS_OK;
}
extern "C" int fs_set_shape_states(fs_ctx *ctx, int env, const float *in, int n_floats) {
    FsEnv *e = get_env(ctx, env, false);
    if (!e) return FS_ERR_ARG;
    LANE_GUARD(ctx, env);
    if (e->shapes.count == 0) return FS_OK;  // loops over the internal count (pyflex.cpp:839)
    if (!in) return FS_ERR_ARG;
    CHECK_LEN(n_floats, 14 * e->shapes.count);
    HIP_TRY(hipSetDevice(ctx->device));
    for (int q = 0; q < e->shapes.count; ++q) {
        const float *o = in + 14 * q;
        e->shapes.pos[q].x = o[0]; e->shapes.pos[q].y = o[1]; e->shapes.pos[q].z = o[2];
        e->shapes.prev[q].x = o[3]; e->shapes.prev[q].y = o[4]; e->shapes.prev[q].z = o[5];
        for (int k = 0; k < 4; ++k) { e->shape_rot[q][k] = o[6 + k]; e->shape_prev_rot[q][k] = o[10 + k]; }
    }
    return push_shapes(ctx, env);
}

// ---- camera (asymmetric layouts are the reference's: pyflex.cpp:891-922)
extern "C" int fs_get_camera_params(fs_ctx *ctx, int env, float *o) {
    FsEnv *e = get_env(ctx, env, false);
    if (!e || !o) return FS_ERR_ARG;
    o[0] = (float)e->cam.width; o[1] = (float)e->cam.height;
    for (int k = 0; k < 3; ++k) { o[2 + k] = e->cam.pos[k]; o[5 + k] = e->cam.angle[k]; }
    return FS_OK;
}
extern "C" int fs_set_camera_params(fs_ctx *ctx, int env, const float *in) {
    FsEnv *e = get_env(ctx, env, false);
    if (!e || !in) return FS_ERR_ARG;
    for (int k = 0; k < 3; ++k) { e->cam.pos[k] = in[k]; e->cam.angle[k] = in[3 + k]; }
    e->cam.width = (int)in[6];
    e->cam.height = (int)in[7];
    return FS_OK;
}

extern "C" int fs_render(fs_ctx *ctx, int env, unsigned char *rgba, int n_bytes, float *depth, int n_floats) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !rgba || !depth) return FS_ERR_ARG;
    const int px = e->cam.width * e->cam.height;
    CHECK_LEN(n_bytes, 4 * px);
    CHECK_LEN(n_floats, px);
    HIP_TRY(hipSetDevice(ctx->device));
    return fs_render_env(ctx, env, rgba, depth);
}

extern "C" int fs_get_sphere_mesh(fs_ctx *ctx, int env, float *verts, float *normals, int n_floats, int *tris, int n_ints) {
    FsEnv *e = get_env(ctx, env);
    if (!e) return FS_ERR_ARG;
    const int s = e->shapes.count;
    if (verts || normals) CHECK_LEN(n_floats, 4 * 441 * s);
    if (tris) CHECK_LEN(n_ints, 3 * 800 * s);
    HIP_TRY(hipSetDevice(ctx->device));
    return fs_sphere_mesh_env(ctx, env, verts, normals, tris);
}

extern "C" int fs_coverage(fs_ctx *ctx, double *out, int n_doubles) {
    if (!ctx || !out) return FS_ERR_ARG;
    CHECK_LEN(n_doubles, ctx->n_envs);
    HIP_TRY(hipSetDevice(ctx->device));
    return fs_coverage_all(ctx, out);
}

extern "C" int fs_get_last_neighbors(fs_ctx *ctx, int env, int *counts, int *lists) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !counts || !lists) return FS_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = e->host.n;
    int rc = d2h(ctx, counts, e->dev.ncount, size_t(4) * n);
    if (rc != FS_OK) return rc;
    for (int i = 0; i < n; ++i) counts[i] &= 0xff;  // (bits 8-31 of the word: the particle's shape candidates, below)
    std::vector<int> slotmajor(size_t(n) * FS_MAX_NEIGHBORS);
    rc = d2h(ctx, slotmajor.data(), e->dev.nlist, slotmajor.size() * 4);
    if (rc != FS_OK) return rc;
    for (int i = 0; i < n; ++i)
        for (int s = 0; s < FS_MAX_NEIGHBORS; ++s)
#ifdef FS_BLOCK_CLOCKS  // developer build: the instrumented kernels leave their clocks in rows 94 / 95 (scripts/block_clocks.py, stream_clocks.py)
            lists[size_t(i) * FS_MAX_NEIGHBORS + s] = (s < counts[i] || s >= 94) ? slotmajor[size_t(s) * n + i] : -1;
#else
            lists[size_t(i) * FS_MAX_NEIGHBORS + s] = s < counts[i] ? slotmajor[size_t(s) * n + i] : -1;
#endif
    return FS_OK;
}

extern "C" int fs_get_last_shape_candidates(fs_ctx *ctx, int env, unsigned *masks) {
    FsEnv *e = get_env(ctx, env);
    if (!e || !masks) return FS_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = e->host.n;
    int rc = d2h(ctx, masks, e->dev.ncount, size_t(4) * n);
    if (rc != FS_OK) return rc;
    for (int i = 0; i < n; ++i) masks[i] >>= 8;
    return FS_OK;
}

extern "C" void *fs_device_positions(fs_ctx *ctx, int env) {
    FsEnv *e = get_env(ctx, env);
    return e ? (void *)e->dev.pos : nullptr;
}
