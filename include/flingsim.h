/*
 * flingsim.h -- C-ABI of the MI355X-native FlingBot cloth hot path (libflingsim.so).
 *
 * This is the drop-in boundary: every entry point replaces one function of the reference's `pyflex` pybind11
 * module (PyFlex/bindings/pyflex.cpp, cited per function) and takes only plain pointers and sizes, so the
 * reference's own binding layer (pybind11), ctypes, cffi or any other FFI can bind it (INTEGRATION.md shows the stubs).
 *
 * Differences from the reference boundary, all additive:
 *   - the reference is a process-global singleton (one g_solver, main.cpp:163); here a context owns `n_envs`
 *     independent cloth episodes that step in ONE batched launch.  `pyflex.*` == env 0 of a 1-env context.
 *   - errors are return codes (0 = ok, <0 = error) + fs_last_error(), instead of printf/exit (pyflex.cpp:103-107).
 *   - the caller owns every host buffer; `n_*` arguments are ELEMENT counts (floats / ints), and are checked.
 *
 * There is no CPU fallback: fs_create fails if no HIP device is usable.
 */
#ifndef FLINGSIM_H
#define FLINGSIM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_ctx fs_ctx;

#define FS_OK 0
#define FS_ERR_ARG (-1)
#define FS_ERR_HIP (-2)
#define FS_ERR_STATE (-3)
#define FS_ERR_LIMIT (-4) /* fs_movep: step limit reached (MoveJointsException, environment/exceptions.py) */

/* solver back-ends (fs_set_solver) */
#define FS_SOLVER_AUTO 0    /* fused LDS-resident kernel when the episode fits one CU's LDS, else streaming */
#define FS_SOLVER_STREAM 1  /* per-stage kernels over HBM-resident SoA state (any particle count) */
#define FS_SOLVER_FUSED 2   /* one workgroup per episode, particle state resident in LDS for the whole step */
#define FS_SOLVER_FUSED_GENERIC 3 /* FUSED, but spring adjacency streamed from L2 instead of register-resident */
#define FS_SOLVER_STREAM_ELL 4    /* STREAM, but with the uncompressed (index, length, stiffness) adjacency arrays */
#define FS_SOLVER_FUSED_CODED 5   /* FUSED, but never the grid-64 form: the dictionary-coded adjacency kernel */
#define FS_SOLVER_STREAM_CODED 6  /* STREAM, but never the grid-L form: coded / latency / grid forms chosen by launch size */
#define FS_SOLVER_STREAM_SPLIT 7  /* STREAM, but the substep boundary as separate finalize / predict / scan / scatter launches
                                     (the form cloths above 16384 particles and launches of fewer than 16 episodes take)
                                     instead of fs_k_boundary */
#define FS_SOLVER_STREAM_MERGED 8 /* STREAM, with fs_k_boundary at every launch size */
#define FS_SOLVER_COTENANT 9      /* for a process that SHARES the device with others (fs_tenants_*): every launch whose episodes
                                     all fit the fused kernel takes it whatever the launch size (one launch per frame on one
                                     compute unit each -- co-tenants' frames run side by side), every other launch is AUTO's */

const char *fs_last_error(void);
int fs_version(void);

/* pyflex.init (pyflex.cpp:15-124): create the solver/renderer context on HIP device `device` for `n_envs` episodes.
   camera_width/height are the render target size (g_screenWidth/Height). */
fs_ctx *fs_create(int device, int n_envs, int camera_width, int camera_height);
/* pyflex.clean (pyflex.cpp:126-160) */
void fs_destroy(fs_ctx *ctx);
int fs_n_envs(const fs_ctx *ctx);
int fs_set_solver(fs_ctx *ctx, int solver);
int fs_get_solver(const fs_ctx *ctx);
/* 1 when FS_SOLVER_FUSED can step this episode (<= 4096 particles, one collision plane at most), 0 when it cannot (the
   caller then keeps FS_SOLVER_AUTO): lets a front end choose the back-end when a scene is set instead of finding out from a
   failed fs_step */
int fs_fused_fits(fs_ctx *ctx, int env);

/* Co-tenant table (csrc/fs_tenants.cpp): the reference runs one PyFleX per Ray worker on one GPU (`--num_processes 16`,
   README.md:147-148, utils.py:144-157); the processes of one user that drive the same physical device through this library find
   each other in a small mmap-ed file per (user, device) under /dev/shm (FLINGSIM_TENANT_DIR overrides), so that the `pyflex`
   module can pick FS_SOLVER_COTENANT by itself.  `device_key` names the physical device: fs_device_key() = its PCI bus id.
   fs_tenants_register: add this process, returns the number of live tenants including it; fs_tenants_count: live tenants
   (prune > 0: also clear the entries of processes that died without unregistering; prune < 0: the number of occupied slots as
   they stand, no liveness check, no system call); fs_tenants_unregister: remove this process.
   Host-only (no HIP call): usable, and tested, without a GPU. */
int fs_device_key(fs_ctx *ctx, char *out, int n_chars);
int fs_tenants_register(const char *device_key);
int fs_tenants_count(const char *device_key, int prune);
int fs_tenants_unregister(const char *device_key);

/* pyflex.set_scene (pyflex.cpp:229-244 -> main.cpp:613 Init -> softgym_cloth.h:33 Initialize).
   scene_params[19] layout: flex_utils.py:332-342.  Empty `verts` selects the grid path (helpers.h:838). */
int fs_set_scene(fs_ctx *ctx, int env, const float *scene_params, int n_params, const float *verts, int n_vert_floats,
                 const int *stretch, int n_stretch_ints, const int *bend, int n_bend_ints, const int *shear,
                 int n_shear_ints, const int *faces, int n_face_ints);

/* pyflex.step (pyflex.cpp:213-222 -> main.cpp:2120 UpdateFrame -> NvFlexUpdateSolver(dt, substeps)).
   Advances env (or every env with a scene when env == -1) by n_steps frames.  Asynchronous on the context's HIP
   stream; every getter synchronises. */
int fs_step(fs_ctx *ctx, int env, int n_steps);
/* the same for a list of episodes (each listed once): one batched launch sequence for exactly those */
int fs_step_list(fs_ctx *ctx, int n, const int *envs, int n_steps);
int fs_sync(fs_ctx *ctx);
/* fs_step bracketed by HIP events recorded on the context's stream; returns the elapsed device time of the launch(es)
   in milliseconds.  Used by bench.py for the per-launch kernel duration (roofline). */
int fs_step_timed(fs_ctx *ctx, int env, int n_steps, float *elapsed_ms);
/* HIP-event stopwatch on the context's stream (non-blocking start; stop waits for the stream to reach it). */
int fs_timer_start(fs_ctx *ctx);
int fs_timer_stop(fs_ctx *ctx, float *elapsed_ms);
/* the HIP stream (hipStream_t) the context launches on, for callers that time with HIP events */
void *fs_stream(fs_ctx *ctx);

/* counts: pyflex.get_n_particles :326, get_n_shapes :334; springs / triangles implied by get_edges / get_faces */
int fs_n_particles(const fs_ctx *ctx, int env);
int fs_n_springs(const fs_ctx *ctx, int env);
int fs_n_triangles(const fs_ctx *ctx, int env);
int fs_n_shapes(const fs_ctx *ctx, int env);

/* particle mirrors.  get_positions :414 / set_positions :464 (float[4N] xyz + invMass),
   get_velocities :753 / set_velocities :772 (float[3N]), get_phases :378 / set_phases :395 (int[N]),
   get_restPositions (float[4N]), get_edges :433 (int[2M]), get_faces :449 (int[3T]). */
int fs_get_positions(fs_ctx *ctx, int env, float *out, int n_floats);
int fs_set_positions(fs_ctx *ctx, int env, const float *in, int n_floats);
int fs_get_velocities(fs_ctx *ctx, int env, float *out, int n_floats);
int fs_set_velocities(fs_ctx *ctx, int env, const float *in, int n_floats);
int fs_get_phases(fs_ctx *ctx, int env, int *out, int n_ints);
int fs_set_phases(fs_ctx *ctx, int env, const int *in, int n_ints);
int fs_get_rest_positions(fs_ctx *ctx, int env, float *out, int n_floats);
int fs_get_normals(fs_ctx *ctx, int env, float *out, int n_floats); /* NvFlexGetNormals, main.cpp:2286 */
int fs_get_edges(fs_ctx *ctx, int env, int *out, int n_ints);
int fs_get_faces(fs_ctx *ctx, int env, int *out, int n_ints);
int fs_get_spring_lengths(fs_ctx *ctx, int env, float *out, int n_floats);
int fs_get_spring_stiffness(fs_ctx *ctx, int env, float *out, int n_floats);
/* effective NvFlexParams of the env as a packed float[32] (layout: DESIGN.md "parameter table") */
int fs_get_params(fs_ctx *ctx, int env, float *out, int n_floats);
/* overwrite the packed parameter table (same layout as fs_get_params); used for experiments -- FlingBot never changes
   the solver parameters after set_scene */
int fs_set_params(fs_ctx *ctx, int env, const float *in, int n_floats);
/* pyflex.get_scene_lower / get_scene_upper (pyflex.cpp:865-889) */
int fs_get_scene_bounds(fs_ctx *ctx, int env, float *lower3, float *upper3);

/* shapes.  add_sphere :311 (helpers.h:484), clear_shapes (helpers.h:1677),
   get_shape_states :789 / set_shape_states :832: float[14S] = pos3, prevPos3, quat4, prevQuat4 */
int fs_add_sphere(fs_ctx *ctx, int env, float radius, const float *pos3, const float *quat4);
int fs_clear_shapes(fs_ctx *ctx, int env);
int fs_get_shape_states(fs_ctx *ctx, int env, float *out, int n_floats);
int fs_set_shape_states(fs_ctx *ctx, int env, const float *in, int n_floats);

/* camera.  get_camera_params :891 -> [w,h,px,py,pz,ax,ay,az]; set_camera_params :908 <- [px,py,pz,ax,ay,az,w,h] */
int fs_get_camera_params(fs_ctx *ctx, int env, float *out8);
int fs_set_camera_params(fs_ctx *ctx, int env, const float *in8);

/* pyflex.render (pyflex.cpp:924-1133): RGBA8 bottom-up [h*w*4] and linear depth [h*w] of env. */
int fs_render(fs_ctx *ctx, int env, unsigned char *rgba, int n_bytes, float *depth, int n_floats);

/* white-box access for tests: the triangle meshes fs_render rasterises for env's kinematic spheres = what the reference
   draws for them (main.cpp:1739-1751: CreateSphere(20, 20, r) core/mesh.cpp:858-902, transformed by the shape's PREVIOUS
   position and rotation).  verts / normals: float[4 * 441 * S], tris: int[3 * 800 * S]; any of them may be null. */
int fs_get_sphere_mesh(fs_ctx *ctx, int env, float *verts, float *normals, int n_floats, int *tris, int n_ints);

/* coverage reward of every env (flex_utils.py:358-395 get_current_covered_area with pos=None, particle radius
   0.00625), out[n_envs] in float64 like the reference's return value; envs without a scene report 0. */
int fs_coverage(fs_ctx *ctx, double *out, int n_doubles);

/* white-box access for tests: particle-contact candidate lists of the last substep, counts[N], lists[N*96] */
int fs_get_last_neighbors(fs_ctx *ctx, int env, int *counts, int *lists);
/* white-box access for tests: the collideShapes stage of the last substep (NvFlex.h:205; candidates within collisionDistance +
   shapeCollisionMargin of the predicted position, NvFlex.h:145-147, at most maxContactsPerParticle, NvFlex.h:361), masks[N]:
   bit q = plane q, bit 8 + q = kinematic sphere q */
int fs_get_last_shape_candidates(fs_ctx *ctx, int env, unsigned *masks);
/* white-box access for tests: which kernel form the most recent fs_step* launch of this context ran (0 = none yet).
   The launch size selects the form (fs_solver.hip), so a parity test asserts that it compared the form it meant to. */
#define FS_FORM_FUSED_12 1       /* fs_k_fused_step<12>: register-resident coded adjacency, <= 12 springs per particle */
#define FS_FORM_FUSED_16 2       /* fs_k_fused_step<16>: the same with <= 16 springs per particle */
#define FS_FORM_FUSED_GENERIC 3  /* fs_k_fused_step<0>: adjacency streamed from the ELL arrays */
#define FS_FORM_STREAM_EAGER 4   /* fs_k_iterate_eager<false>: latency form of small launches */
#define FS_FORM_STREAM_CODED 5   /* fs_k_iterate<true>: throughput form, one-byte spring codes */
#define FS_FORM_STREAM_ELL 6     /* fs_k_iterate<false>: throughput form, uncompressed adjacency */
#define FS_FORM_STREAM_GRID 7    /* fs_k_iterate_grid: neighbour ids from the grid coordinates (large launches) */
#define FS_FORM_STREAM_GRIDL 9   /* fs_k_iterate_gridl: canonical grid cloths, rest lengths from the per-particle table */
#define FS_FORM_FUSED_GRID64 8   /* fs_k_fused_grid64: 64-wide grid cloths, packed two-particle springs, no adjacency */
#define FS_FORM_STREAM_GRIDL_TP 10 /* fs_k_iterate_gridl_tp: the grid-L iteration's throughput form -- launches above 262 144 particles of
                                      tether-free cloths (the evaluation loop at its real sizes): springs in two halves of six,
                                      equal-mass fast path */
int fs_last_kernel_form(const fs_ctx *ctx);
/* white box: how the most recent streaming launch did its substep boundaries (finalize + predict + bucket sort): 0 = four
   kernels (launches of fewer than 16 episodes, cloths above 16384 particles), 1 = fs_k_boundary (one launch per boundary) */
int fs_last_boundary_form(const fs_ctx *ctx);
/* Streaming back-end: a launch list that cannot fill the chip is split into `groups` slot ranges whose launch chains run
   concurrently on streams of their own (episodes are independent; results do not change).  0 = the library's measured
   default (1 below ~24 x 4096 particles, 2 from there), 1..4 = forced.  fs_last_stream_groups: the
   number of chains of the most recent streaming launch (white box for the tests). */
int fs_set_stream_groups(fs_ctx *ctx, int groups);
int fs_last_stream_groups(const fs_ctx *ctx);
/* white-box access for tests: y[i] = the reciprocal square root the constraint kernels use (csrc/fs_constraints.h fs_rsqrt),
   evaluated on the device for n host values -- what pins the checker's restatement of it to this chip */
int fs_eval_rsqrt(fs_ctx *ctx, const float *x, float *y, int n);
/* device pointer of env's position array (float4[N]) for zero-copy consumers (torch) */
void *fs_device_positions(fs_ctx *ctx, int env);

/* ---- on-device picker + movep executor (additive) -------------------------------------------
   Native counterpart of the host loop the reference runs around pyflex.step(): SimEnv.movep (simEnv.py:739-769) ->
   PickerPickPlace.step (flex_utils.py:223-252) -> Picker.step (flex_utils.py:121-205).  The pickers are the
   episode's kinematic spheres (fs_add_sphere); results are identical to driving fs_step through those Python classes. */
/* Picker.reset bookkeeping (flex_utils.py:85,99-101): nothing held; remember every particle's inverse mass. */
int fs_picker_reset(fs_ctx *ctx, int env, double picker_threshold, double particle_radius);
/* Picker.picker_radius (flex_utils.py:57) as the python float the reference adds into the grasp threshold
   (flex_utils.py:154-155); without this call the threshold uses the float32 radius pyflex.add_sphere stored for shape 0.
   fs_picker_reset clears it. */
int fs_picker_set_radius(fs_ctx *ctx, int env, double picker_radius);
/* simulation steps, summed over the episodes, that the most recent fs_movep / fs_movep_batch* call -- or the movep
   entries of the most recent fs_advance call -- executed (movep iterations that find the pickers on their targets do not
   step the simulation, flex_utils.py:231-233); 0 after a call that returned an error before launching anything */
long long fs_last_movep_steps(const fs_ctx *ctx);
/* One chunk (at most `cap` simulation steps) for episodes that are in different phases of their primitives, so that all of
   them share every launch sequence; the chunk ends with the first movep that completes, but takes at least cap_min steps
   (one host round trip per call) when somebody has that many left: kind[a] = 0: SimEnv.movep (simEnv.py:739-769) towards targets[a] ([S][3], float64; f32[a]
   != 0: the caller's targets were a float32 array, see fs_movep_batch_f32) with grasp[a][S], speed[a], iteration limit[a],
   min_steps[a] (< 0: None), resumed at loop iteration start[a]; kind[a] = 1: flex_utils.wait_until_stable
   (flex_utils.py:430-441) with max_steps = limit[a], of which start[a] steps are already taken, tolerance tolerance[a];
   kind[a] = 2: limit[a] plain pyflex.step() calls, start[a] of them taken (status 1 when done).
   Out: progress[a] = loop iteration / step count reached (pass it back as start[a]), status[a] = 0 continue with another
   call, 1 finished (targets reached / stable), 2 finished at the limit (movep: MoveJointsException; wait: not stable),
   steps[a] = simulation steps this call took for the episode.  Results are identical to the uninterrupted loops. */
int fs_advance(fs_ctx *ctx, int n, const int *envs, const int *kind, const double *targets, const int *grasp,
               const double *speed, const int *limit, const int *min_steps, const int *f32, const int *start, double eps,
               const double *tolerance, int cap_min, int cap, int *progress_out, int *status_out, int *steps_out);
/* fs_advance in two halves, so that the host can work while a chunk runs (flingbot_amd/schedule.py pipelines them: the
   next chunk is queued, and the requests of the episodes that just finished something are served, while the device is
   still busy with the previous one).
   fs_advance_begin: same arguments; queues the chunk's launches on the context's main stream and returns a ticket (>= 0; at
   most 4 may be open) or an error code (< 0).  The movep entries' outputs are final when it returns -- their trajectories
   are planned on the host -- and so are those of a wait / step loop whose steps were already used up; every other wait /
   step entry has status -1 until fs_advance_end.  start[a] = -1 for a wait / step entry CONTINUES the loop from the state
   the device keeps per episode: the host may queue the next chunk of a wait before it knows whether the previous chunk
   ended it (if it did, the episode's entries retire at once and nothing is stepped).
   fs_advance_end: waits for the ticket's launches and fills in the wait / step entries (same index a as in the begin
   call; progress = steps of the loop taken so far, steps = progress - start, or -1 for start = -1).
   fs_advance_in_flight: open tickets. */
int fs_advance_begin(fs_ctx *ctx, int n, const int *envs, const int *kind, const double *targets, const int *grasp,
                     const double *speed, const int *limit, const int *min_steps, const int *f32, const int *start, double eps,
                     const double *tolerance, int cap_min, int cap, int *progress_out, int *status_out, int *steps_out);
int fs_advance_end(fs_ctx *ctx, int ticket, int *progress_out, int *status_out, int *steps_out);
int fs_advance_in_flight(const fs_ctx *ctx);
/* The service lane: between fs_service_lane(ctx, 1) and fs_service_lane(ctx, 0) every entry point of the library works on
   a second, high-priority stream, so that reductions, observations and resets for episodes that are NOT part of a chunk
   in flight neither queue up behind the chunk nor wait for it.  Contract: on the lane the caller touches only such
   episodes (or ones whose wait loop in the chunk has already ended).  Leaving the lane orders the main stream behind it.
   The calls that REWRITE an episode (fs_set_scene*, fs_set_positions / velocities / phases / params / shape_states /
   particles, fs_add_sphere, fs_clear_shapes, fs_picker_reset) check the contract: FS_ERR_STATE for an episode that an open
   ticket still steps or moves (an entry that continues a wait loop fs_advance_end has already reported over does not count).
   The calls that STEP the simulation (fs_step, fs_step_list, fs_step_timed, fs_wait_until_stable, fs_movep*, fs_advance*)
   return FS_ERR_STATE on the lane while any ticket is open, whichever episodes they list: a launch sequence uses per-context
   tables and streams the chunk in flight is still reading. */
int fs_service_lane(fs_ctx *ctx, int on);
/* fs_advance's stopwatch since fs_create: out5 = calls, launch sequences, wall ms inside the calls, device ms between a
   call's first and last launch, wall ms the calls spent before their first launch (planning, tables, upload) */
int fs_advance_timing(const fs_ctx *ctx, double *out5);
/* The context's buffer pool (episode slabs, topology images, ticket tables: recycled by size instead of hipFree / hipMalloc,
   which synchronise the device).  out[0] = idle bytes held, out[1] = idle buffers, out[2] = the idle limit in bytes: what
   lies beyond it goes back to the driver, oldest first (4 GiB; FLINGSIM_POOL_IDLE_MB overrides it when the library loads). */
int fs_pool_stats(const fs_ctx *ctx, long long *out3);
/* picked particle index per picker (-1 = none) */
int fs_picker_get_picked(fs_ctx *ctx, int env, int *out, int n_ints);
/* SimEnv.movep: move picker k toward targets[3k..3k+2] by `speed` per simulation step with grasp flag grasp[k], until all
   are within eps (and more than min_steps iterations ran; min_steps < 0 = None).  Runs every simulation step on the
   device without host round trips.  iterations_out = loop iterations (what movep's `step` counts).
   Returns FS_ERR_LIMIT when `limit` iterations were not enough. */
int fs_movep(fs_ctx *ctx, int env, const double *targets, const int *grasp, double speed, int limit, int min_steps,
             double eps, int *iterations_out);
/* the same for n episodes at once: targets double[n][S][3], grasp int[n][S], iterations_out int[n] */
int fs_movep_batch(fs_ctx *ctx, int n, const int *envs, const double *targets, const int *grasp, double speed, int limit,
                   int min_steps, double eps, int *iterations_out);
/* the same when the reference's `pos` argument is a float32 numpy array (stretch_cloth builds its targets from the
   float32 picker positions, simEnv.py:146-156,180-182): movep's own arithmetic then runs in float32 */
int fs_movep_batch_f32(fs_ctx *ctx, int n, const int *envs, const float *targets, const int *grasp, double speed,
                       int limit, int min_steps, double eps, int *iterations_out);

/* ---- device-side feedback loops and reductions -------------------------------------------------
   The reference's primitives download whole particle arrays to take one number from them, every simulation step or
   every loop trip; these entry points compute the same numbers on the device.                                        */
/* flex_utils.py:430-441 wait_until_stable for n episodes at once: before EVERY step the episode's max |velocity
   component| (float32, compared in double like `np.abs(v).max() < tolerance`) is tested; an episode that passes stops
   stepping.  At most max_steps steps.  steps_out[k] = simulation steps taken, stable_out[k] = 1 when the test passed
   (the function's return value), 0 when max_steps ran out.  No host round trip per step. */
int fs_wait_until_stable(fs_ctx *ctx, int n, const int *envs, int max_steps, double tolerance, int *steps_out,
                         int *stable_out);
/* out[3k..3k+2] = { min height y, max height y, max |velocity component| } of episode envs[k]
   (simEnv.py:186-200 lift_cloth `heights.min()`, :809-813 is_cloth_grasped `heights.max()`, flex_utils.py:435) */
int fs_cloth_stats(fs_ctx *ctx, int n, const int *envs, float *out, int n_floats);
/* stretch_cloth's probe (simEnv.py:155-168) for episode envs[k]:
     single_grasp_out[k] = 1 when every particle with y > height_thr[k] has x < 0, or every one has x > 0 (also when there
                           is none), evaluated in float32 like the numpy expression;
     nearest_out[3k..]   = position of the particle whose float32 distance to midpoint_xz[2k..2k+1] in the x-z plane,
                           sqrt(dx*dx + dz*dz), is smallest (lowest index on ties = Python's stable sort). */
int fs_stretch_probe(fs_ctx *ctx, int n, const int *envs, const float *midpoint_xz, const float *height_thr,
                     int *single_grasp_out, float *nearest_out);
/* SimEnv.preaction / postaction (simEnv.py:464-475): keep the current positions of the listed episodes on the device,
   and later report out[k] = max_i || |pos_i - kept_i| ||_2 in float32 -- np.linalg.norm(np.abs(post - pre), axis=1).max() --
   which the reference compares with 5e-2 to end an episode whose action did not move the cloth. */
int fs_snapshot_positions(fs_ctx *ctx, int n, const int *envs);
int fs_max_displacement(fs_ctx *ctx, int n, const int *envs, float *out, int n_floats);
/* Write ONE particle per listed episode: position + inverse mass from pos4[4k..4k+3], velocity zeroed when zero_velocity != 0.
   The reference's task generator (environment/tasks.py:177-224) moves a pinned pick point by reading and rewriting the
   whole position and velocity arrays through pyflex every simulation step. */
int fs_set_particles(fs_ctx *ctx, int n, const int *envs, const int *particle_ids, const float *pos4, int zero_velocity);

/* ---- observation transforms on the device ---------------------------------------------------
   learning/nets.py:155-193 prepare_image: n_transforms rotated / scaled / resized copies of one observation.
     d_img   device float32 [channels][size][size]                  (the tensor preprocess_obs returns)
     matrix  host double [n][4], offset host double [n][2]          rotation matrix rows and offset exactly as
             scipy.ndimage.rotate(reshape=False) derives them from the angle (c, s = cosdg, sindg; [[c, s], [-s, c]];
             offset = centre - matrix @ centre) -- computed by the caller so special angles stay exact
     scale   host double [n]                                        <1 centre crop, >1 replicate pad to int(scale*size)
     d_out   device float32 [n][channels][dim][dim]
     d_work  device scratch of fs_prepare_image_work_bytes() bytes; stream: hipStream_t the work is enqueued on
   Cubic-spline rotation (mode='nearest', float64 like scipy) evaluated only at the pixels the nearest-neighbour resize
   keeps. */
size_t fs_prepare_image_work_bytes(int channels, int size, int n_transforms);
int fs_prepare_image(const float *d_img, int channels, int size, int n_transforms, const double *matrix,
                     const double *offset, const double *scale, int dim, float *d_out, void *d_work, void *stream);

/* ---- action selection on the device ----------------------------------------------------------
   SimEnv.get_max_value_valid_action (environment/simEnv.py:560-661): the best-valued VALID action over all primitives,
   transforms and pixels.  A candidate (primitive p, transform t, pixel (y, z)) is valid when its two reach points lie
   in the obs_dim image (get_action_params :517-537), map into the pretransform image (pixels_to_3d_positions,
   environment/utils.py:237-276, with `transform_mats[t]` = get_transform_matrix(depth_dim, obs_dim, -rotation, scale)
   row-major), unproject through `d_depth` (pixel_to_3d :214-234) and satisfy check_action_reachability (:539-558;
   stretchdrag also at the end of its drag :624-642).
     d_values  device float32 [n_primitives][n_transforms][obs_dim][obs_dim] (stacked value maps)
     best_index_out  flattened index into values[:, :, g:-g, g:-g] (g = pix_grasp_dist) of the winner -- the entry the
                     reference's descending walk stops at (ties: lowest flattened index) -- or -1 when nothing is valid
     d_work    device scratch of fs_select_action_work_bytes(n_transforms) bytes; stream: hipStream_t */
#define FS_ACTION_FLING 0
#define FS_ACTION_STRETCHDRAG 1
#define FS_ACTION_DRAG 2
#define FS_ACTION_PLACE 3
#define FS_ACTION_MAX_PRIMITIVES 8
size_t fs_select_action_work_bytes(int n_transforms);
int fs_select_action(const float *d_values, int n_primitives, const int *primitive_kinds, int n_transforms, int obs_dim,
                     int pix_grasp_dist, int pix_drag_dist, int pix_place_dist, const double *transform_mats,
                     const float *d_depth, int depth_dim, double focal_length, const double *pose_matrix,
                     const double *left_arm_base, const double *right_arm_base, double reach_distance_limit,
                     double stretchdrag_dist, double grasp_height, long long *best_index_out, float *best_value_out,
                     void *d_work, void *stream);

/* ---- observation stage on the device ---------------------------------------------------------
   Everything the reference does on the host between pyflex.render() and prepare_image, without downloading the frame:
   get_image (environment/flex_utils.py:418-427: flip rows, drop alpha, cv2.resize INTER_LINEAR to image_dim),
   SimEnv.get_cloth_mask (environment/simEnv.py:699-708: RGB2HSV, inRange((0,0,0),(100,100,100)) == 0, largest connected
   component, environment/utils.py:585-601), the bounding box SimEnv.get_obs derives its adaptive scale from
   (simEnv.py:717-731) and preprocess_obs (environment/utils.py:579-582).
     d_obs   device float32 [4][image_dim][image_dim]: rgb / 255 and depth, rows top-down
     d_mask  device uint8 [image_dim][image_dim] (1 = pixel of the largest cloth component) or NULL
     bbox    host int[5]: x.min, x.max, y.min, y.max of np.where(mask) (x = row) and the component's pixel count; -1 / 0
             when no pixel passes the colour test (the reference's get_obs then leaves the scale factors alone)
     d_work  device scratch of fs_observe_work_bytes(image_dim) bytes
   Renders with the episode's camera (fs_set_camera_params), runs on the context's stream and returns when the results are
   complete.  cv2 / skimage conventions are restated in oracle/observe.py (parity with the reference's own cv2 build is
   unpinned: cv2 is absent from the build image). */
size_t fs_observe_work_bytes(int image_dim);
int fs_observe(fs_ctx *ctx, int env, int image_dim, float *d_obs, unsigned char *d_mask, int *bbox, void *d_work);
/* fs_observe for n episodes with the host round trips of one call (the labelling rounds and the results of all
   episodes come back together).  d_obs [n][4][S][S], d_mask [n][S][S] or NULL, bbox [n][5],
   d_work: n * fs_observe_work_bytes(image_dim) bytes.  Every result equals the single call's. */
int fs_observe_batch(fs_ctx *ctx, int n, const int *envs, int image_dim, float *d_obs, unsigned char *d_mask, int *bbox,
                     void *d_work);

/* ---- value network forward ------------------------------------------------------------------
   SpatialValueNet.forward (learning/nets.py:81-141) in eval mode for size x size = 64 x 64 observations (the
   reference's obs_dim): normalise, Conv3x3(C->16)+BN+LeakyReLU, 8 residual blocks of two Conv3x3(16->16)+BN, Conv3x3(16->1).
   The caller folds each BatchNorm into the preceding convolution (w' = w g / sqrt(var + eps),
   b' = beta - running_mean g / sqrt(var + eps)) and packs the result once:
     fs_value_net_pack   host -> host.  in_channels 1 / 3 / 4; mean, std float32[in_channels] (preprocess_obs :131-137);
                         w_first [16][in_channels][3][3], b_first [16]; w_blocks [16 convs][16][16][3][3] in network order
                         (block 0 conv1, block 0 conv2, block 1 conv1, ...), b_blocks [16][16]; w_last [1][16][3][3];
                         packed float32[fs_value_net_param_floats()]
     fs_value_net_forward  d_params: the packed block copied to the device; d_obs device float32
                         [batch][obs_channels][64][64], of which channels channel_offset .. channel_offset+in_channels-1
                         feed the network (rgb_only: 0..2, depth_only: 3); d_out device float32 [batch][64][64];
                         d_work device scratch of fs_value_net_work_bytes(batch, 64) bytes; stream: hipStream_t.
   fp32 throughout (the 16->16 convolutions on v_mfma_f32_16x16x4_f32); any other size returns FS_ERR_ARG. */
size_t fs_value_net_param_floats(void);
size_t fs_value_net_work_bytes(int batch, int size);
int fs_value_net_pack(int in_channels, const float *mean, const float *std, const float *w_first, const float *b_first,
                      const float *w_blocks, const float *b_blocks, const float *w_last, float *packed);
int fs_value_net_forward(const float *d_params, const float *d_obs, int obs_channels, int channel_offset,
                         int in_channels, int batch, int size, float *d_out, void *d_work, void *stream);

/* ---- host-only entry points (no HIP device needed) ------------------------------------------------------------
   Scene builder exposed on its own so host logic can be checked without a GPU: same arguments as fs_set_scene. */
typedef struct fs_host_scene fs_host_scene;
#define FS_SCENE_POSITIONS 0        /* float[4N] */
#define FS_SCENE_VELOCITIES 1       /* float[3N] */
#define FS_SCENE_PHASES 2           /* int[N] */
#define FS_SCENE_SPRINGS 3          /* int[2M]   (pyflex.get_edges) */
#define FS_SCENE_SPRING_LENGTHS 4   /* float[M] */
#define FS_SCENE_SPRING_STIFFNESS 5 /* float[M] */
#define FS_SCENE_TRIANGLES 6        /* int[3T]   (pyflex.get_faces) */
#define FS_SCENE_TRI_NORMALS 7      /* float[3T] */
#define FS_SCENE_ADJ_OFFSETS 8      /* int[N+1]  particle -> spring CSR */
#define FS_SCENE_ADJ_NEIGHBORS 9    /* int[2M] */
#define FS_SCENE_BOUNDS 10          /* float[6] lower, upper */
#define FS_SCENE_PARAMS 11          /* float[32] packed parameter table */
/* derived tables of the kernels (white box for the CPU tests; empty when the cloth does not have them) */
#define FS_SCENE_RESTNEAR 12        /* uint32[8][n]: per particle up to 16 ids (16 bits each, 0xffff = none) of the particles closer
                                       than the collision radius in the rest pose (SelfCollideFilter, NvFlex.h:166) */
#define FS_SCENE_STREAM_CODES 13    /* uint32[n][4]: one byte per spring slot of the particle, 255 = none */
#define FS_SCENE_STREAM_DICT 14     /* float[entries][4]: bits(j - i), rest length, stiffness, 0 */
#define FS_SCENE_FLAGS 15           /* int[4]: restnear_ok (0 none, 1 packed ids, 2 = the sets are the 8 grid neighbours), g64_ok,
                                       gp_L_ok, gp_halvable */
fs_host_scene *fs_host_scene_build(const float *scene_params, int n_params, const float *verts, int n_vert_floats,
                                   const int *stretch, int n_stretch_ints, const int *bend, int n_bend_ints,
                                   const int *shear, int n_shear_ints, const int *faces, int n_face_ints);
void fs_host_scene_free(fs_host_scene *h);
int fs_host_scene_counts(const fs_host_scene *h, int *n, int *m, int *t, int *max_deg);
int fs_host_scene_copy(const fs_host_scene *h, int what, void *out, int n_elems);
/* fs_set_scene from a scene fs_host_scene_build made earlier (on any thread, without the GPU): the scene is taken out of
   `scene`, which stays valid but empty (free it with fs_host_scene_free).  Same result as fs_set_scene. */
int fs_set_scene_prebuilt(fs_ctx *ctx, int env, fs_host_scene *scene);
/* host-only: the mesh of ONE kinematic sphere exactly as fs_render rasterises it (see fs_get_sphere_mesh) */
int fs_host_sphere_mesh(float radius, const float *prev_pos3, const float *prev_quat4, float *verts, float *normals,
                        int *tris);
/* RenderScene camera / light set-up (main.cpp:1411-1438; core/maths.h:507-598): out[0:16] view, [16:32] proj,
   [32:48] lightTransform (row-major, column vectors), [48:51] lightPos, [51:54] lightDir. */
int fs_camera_matrices(const float *cam_pos3, const float *cam_angle3, int width, int height,
                       const float *scene_lower3, const float *scene_upper3, float *out54);

#ifdef __cplusplus
}
#endif
#endif
