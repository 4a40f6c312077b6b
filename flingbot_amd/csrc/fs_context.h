// fs_context.h -- host runtime of libflingsim: a context owning n_envs cloth episodes on one HIP device.
//
// Replaces the reference's process-global demo runtime (PyFlex/bindings/main.cpp: g_solver / g_buffers / Init /
// UpdateFrame) with a per-context object; the device arrays are the source of truth, host mirrors are materialised
// only at the accessor boundary (pyflex.cpp:326-922 semantics).
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "fs_scene.h"
#include "fs_types.h"

struct fs_ctx;

struct FsTopologyDev {  // immutable, shared by episodes with the same cloth
    fs_ctx *owner = nullptr;  // the slab goes back to this context's buffer pool
    void *slab = nullptr;
    size_t bytes = 0;
    uint64_t key = 0;
    int n = 0, m = 0, max_deg = 0;
    FsVec4 *rest = nullptr;
    int *adj_off = nullptr, *adj_j = nullptr;
    float *adj_len = nullptr, *adj_k = nullptr;
    int *ell_j = nullptr;
    float *ell_len = nullptr, *ell_k = nullptr;
    int dict_size = 0;
    float *dict = nullptr;
    uint32_t *code_w = nullptr, *nbr_w = nullptr, *restnear_w = nullptr;
    int sdict_size = 0;
    FsVec4 *sdict = nullptr;
    FsU32x4 *scode = nullptr;
    int restnear_ok = 0;
    int g64_ok = 0, gp_L_ok = 0, gp_halvable = 0;
    uint32_t gp_magic = 0;
    float *g64_L = nullptr;  // [12][n] canonical-slot rest lengths of the grid-64 fused kernel
    float g64_k[FS_G64_SLOTS] = {0};
    int *tris = nullptr;  // 3t
    int *vt_off = nullptr, *vt_tri = nullptr;  // vertex -> triangles CSR
    int t = 0;
    ~FsTopologyDev();
};

struct FsCamera {
    float pos[3] = {0, 2, 0};
    float angle[3] = {0, 0, 0};
    int width = 720, height = 720;
};

struct FsEnv {
    bool has_scene = false;
    FsHostScene host;  // topology + initial state (positions here are the INITIAL ones)
    std::shared_ptr<FsTopologyDev> topo;
    void *slab = nullptr;  // dynamic state + scratch
    size_t slab_bytes = 0;
    FsEnvDev dev;          // host copy of the descriptor
    FsShapesDev shapes;    // host copy
    float shape_rot[FS_MAX_SHAPES][4], shape_prev_rot[FS_MAX_SHAPES][4];
    // on-device picker state (fs_picker.hip): picked particle per shape (-1 none), inverse masses saved at reset
    int *d_picked = nullptr;   // [FS_MAX_SHAPES]
    float *d_saved_w = nullptr;  // [saved_w_n >= n], grow-only
    int saved_w_n = 0;
    FsVec4 *d_snapshot = nullptr;  // [snapshot_n] positions kept by fs_snapshot_positions (SimEnv.preaction)
    int snapshot_n = 0, snapshot_cap = 0;  // particles of the snapshot taken / capacity of the buffer
    double picker_threshold = 0.005, particle_radius = 0.00625;
    double picker_radius = -1.0;  // < 0: use the float32 radius of shape 0 (fs_picker_set_radius)
    bool picker_ready = false;
    FsCamera cam;
};

#define FS_MAX_STREAM_GROUPS 4
#define FS_ADV_TICKETS 4  // fs_advance_begin calls that may be in flight at once

struct FsPoolBuf { void *ptr; size_t bytes; };

// wait_until_stable / plain-step loop state of an episode on the device, kept ACROSS fs_advance calls so that the host may
// queue the next chunk of the loop before it has seen the result of the previous one (fs_picker.hip)
struct FsWaitDev { int steps, stable, over, pad; };

// one fs_advance_begin call whose launches may still be running
struct FsAdvTicket {
    bool busy = false;
    hipEvent_t done = nullptr;
    void *d_tab = nullptr;  // device tables of the call
    size_t d_tab_bytes = 0;
    void *h_tab = nullptr;  // pinned image of the tables (upload)
    size_t h_tab_bytes = 0;
    FsWaitDev *h_wait = nullptr;  // pinned [n_envs]: the wait states after the call's last launch
    int n = 0;
    std::vector<int> listed;  // the call's episodes (fs_lane_guard)
    std::vector<int> w_arg, w_env, w_kind, w_limit, w_start;  // the call's waiters: index in the caller's arrays, episode, ...
    std::vector<int> w_gen;                                    // ... and the generation of the loop the entry belongs to (fs_ctx::wait_gen)
    std::vector<char> w_skip;                                  // loop budget already used up when the call was made
    size_t n_seq = 0;
    double wall_begin_ms = 0.0;
};

struct fs_ctx {
    int device = 0;
    int n_envs = 0;
    int solver = 0;
    bool force_ell_stream = false;     // FS_SOLVER_STREAM_ELL: streaming kernels with the uncompressed ELL adjacency
    bool force_generic_fused = false;  // FS_SOLVER_FUSED_GENERIC: fused kernel with the streamed ELL adjacency
    bool force_coded_stream = false;   // FS_SOLVER_STREAM_CODED: never the grid-L form (dictionary-coded / latency / grid forms by size)
    bool force_coded_fused = false;    // FS_SOLVER_FUSED_CODED: never the grid-64 form (dictionary-coded adjacency kernel)
    bool force_split_boundary = false; // FS_SOLVER_STREAM_SPLIT: separate finalize / predict / scan / scatter launches (the form for cloths > 16384 particles)
    bool force_merged_boundary = false; // FS_SOLVER_STREAM_MERGED: fs_k_boundary at every launch size (AUTO / STREAM: from 16 episodes on)
    bool bound_attr_set = false;       // fs_k_boundary's dynamic LDS attribute applied on this context's device
    bool fused_attr_set = false;       // hipFuncAttributeMaxDynamicSharedMemorySize applied on this context's device
    int last_form = 0;                 // FS_FORM_* of the most recent solver launch (fs_last_kernel_form)
    long long last_movep_steps = 0;    // simulation steps of the most recent fs_movep* call, all episodes (fs_last_movep_steps)
    hipStream_t stream = nullptr;       // the stream of the CURRENT lane: main_stream, or svc_stream between fs_service_lane(1) / (0)
    hipStream_t main_stream = nullptr;  // solver launches, fs_advance chunks
    hipStream_t svc_stream = nullptr;   // high priority: reductions / observation / resets of episodes that are NOT part of a chunk in flight
    hipEvent_t svc_event = nullptr;
    bool on_svc = false;
    int last_boundary = 0;  // substep-boundary form of the most recent streaming launch: 0 four kernels, 1 fs_k_boundary
    // concurrent chains of the streaming back-end (fs_solver.hip): streams / join events per group, the fork event
    hipStream_t aux_streams[FS_MAX_STREAM_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t aux_events[FS_MAX_STREAM_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_event = nullptr;
    int stream_groups = 0;        // > 0: forced number of chains (fs_set_stream_groups); 0: the default of fs_solver.hip
    int last_stream_groups = 1;   // chains of the most recent streaming launch
    std::vector<FsEnv> envs;
    FsEnvDev *d_envs = nullptr;       // [n_envs]
    FsShapesDev *d_shapes = nullptr;  // [n_envs]
    FsEnvDev *d_slot_envs = nullptr;  // [n_envs] launch table of the streaming kernels (fs_k_slot_table)
    FsSlotSweeps *d_slot_sweeps = nullptr;  // [n_envs] the slots' sphere sweeps per substep, built with the table
    int *d_ids = nullptr;             // [n_envs] launch list
    int *h_ids = nullptr;             // pinned
    std::vector<int> uploaded_ids;    // what d_ids holds (upload_ids skips an identical list)
    std::vector<int> table_ids;       // the list d_slot_envs was built for
    bool d_ids_valid = false;
    unsigned long long desc_epoch = 0;   // bumped whenever a descriptor in d_envs is rewritten
    unsigned long long table_epoch = ~0ull;  // desc_epoch the launch table in d_slot_envs was built at, for uploaded_ids (~0: none)
    void *h_stage = nullptr;          // pinned staging for accessors
    size_t h_stage_bytes = 0;
    std::vector<std::weak_ptr<FsTopologyDev>> topo_cache;
    int cam_width = 720, cam_height = 720;
    // renderer scratch (fs_render.hip)
    void *render_scratch = nullptr;
    size_t render_scratch_bytes = 0;
    void *loop_scratch = nullptr;     // device tables of fs_movep_batch (fs_picker.hip), grow-only
    size_t loop_scratch_bytes = 0;
    void *svc_scratch = nullptr;      // device scratch of the small reductions (fs_loops.hip), grow-only
    size_t svc_scratch_bytes = 0;
    // device buffers kept for reuse: hipFree synchronises the whole device, which an episode reset (fs_set_scene) in the
    // middle of a running evaluation loop cannot afford; buffers return here and are handed out again by size
    std::vector<FsPoolBuf> pool;
    size_t pool_bytes = 0;
    FsWaitDev *d_wait = nullptr;      // [n_envs]
    FsAdvTicket tickets[FS_ADV_TICKETS];
    std::vector<char> wait_over;  // [n_envs] host's knowledge: the episode's wait / step loop has ended (fs_advance_end said so)
    std::vector<int> wait_gen;    // [n_envs] counts the wait / step loops an episode has started (start >= 0): a ticket's report
                                  // only ends the loop it was queued for, never a newer one a later ticket has started
                                  // and no new loop was started since -- its entries in chunks queued ahead retire unstepped
    int tickets_busy = 0;
    double *d_coverage = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;  // fs_timer_start / fs_timer_stop
    // fs_advance's own stopwatch (fs_advance_timing): device time between its first and last launch, wall time of the call
    hipEvent_t adv_ev0 = nullptr, adv_ev1 = nullptr;
    double adv_gpu_ms = 0.0, adv_wall_ms = 0.0, adv_prep_ms = 0.0;
    long long adv_calls = 0, adv_sequences = 0;

    ~fs_ctx();
};

void fs_set_error(const std::string &msg);
bool fs_hip_ok(hipError_t e, const char *what);
void *fs_stage(fs_ctx *ctx, size_t bytes);
void fs_sync_all_streams(fs_ctx *ctx);  // both lanes' streams and the launch chains' streams
void fs_sync_lane(fs_ctx *ctx);         // service lane: its stream; main lane: the main stream and the chains'
void *fs_svc_scratch(fs_ctx *ctx, size_t bytes);
// The service lane's contract, checked: FS_ERR_STATE when a call on the service lane is about to rewrite an episode that is
// part of an fs_advance chunk still in flight (nothing waits for the chunk there, so the write would race with its launches).
int fs_lane_guard(fs_ctx *ctx, int env);
int fs_step_guard(fs_ctx *ctx, const char *who);  // FS_ERR_STATE for a stepping call on the service lane while a chunk is in flight
void *fs_pool_take(fs_ctx *ctx, size_t bytes, size_t *got_bytes);
void fs_pool_give(fs_ctx *ctx, void *ptr, size_t bytes);
void *fs_loop_scratch(fs_ctx *ctx, size_t bytes);

// solver back-ends
// d_ids: optional device copy of `ids` already resident (skips the upload); nullptr = upload ids to ctx->d_ids
int fs_step_stream(fs_ctx *ctx, const std::vector<int> &ids, int n_steps, const int *d_ids = nullptr);
int fs_step_fused(fs_ctx *ctx, const std::vector<int> &ids, int n_steps, const int *d_ids = nullptr);
// picks the back-end like fs_step does and launches n_steps frames for `ids`
int fs_step_ids(fs_ctx *ctx, const std::vector<int> &ids, int n_steps, const int *d_ids = nullptr);
bool fs_fused_supported(const fs_ctx *ctx, const FsEnv &env);
// renderer / coverage
int fs_render_env(fs_ctx *ctx, int env, unsigned char *rgba, float *depth);
int fs_render_device(fs_ctx *ctx, int env, unsigned char **d_rgba_out, float **d_depth_out);
int fs_normals_env(fs_ctx *ctx, int env, float *out4n);
int fs_sphere_mesh_env(fs_ctx *ctx, int env, float *verts4, float *nrms4, int *tris);
int fs_coverage_all(fs_ctx *ctx, double *out);
