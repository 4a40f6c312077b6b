"""ctypes binding of libflingsim.so (include/flingsim.h) -- the batched, multi-episode face of the hot path.

`FlingSim` owns a context of `n_envs` cloth episodes on one MI355X; `FlingSim.env(i)` returns a view with the
reference's `pyflex` method names (PyFlex/bindings/pyflex.cpp:1135-1208) so code written against `pyflex` reads the same.
There is no CPU fallback: construction raises if the HIP library is missing or no GPU is visible.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

FS_SOLVER_AUTO, FS_SOLVER_STREAM, FS_SOLVER_FUSED = 0, 1, 2
FS_SOLVER_FUSED_GENERIC, FS_SOLVER_STREAM_ELL, FS_SOLVER_FUSED_CODED, FS_SOLVER_STREAM_CODED = 3, 4, 5, 6  # test / comparison variants (include/flingsim.h)
FS_SOLVER_STREAM_SPLIT, FS_SOLVER_STREAM_MERGED, FS_SOLVER_COTENANT = 7, 8, 9
# fs_last_kernel_form (white box): which kernel form the last solver launch ran
(FS_FORM_FUSED_12, FS_FORM_FUSED_16, FS_FORM_FUSED_GENERIC, FS_FORM_STREAM_EAGER, FS_FORM_STREAM_CODED, FS_FORM_STREAM_ELL,
 FS_FORM_STREAM_GRID, FS_FORM_FUSED_GRID64, FS_FORM_STREAM_GRIDL, FS_FORM_STREAM_GRIDL_TP) = range(1, 11)

_lib = None


class FlingSimError(RuntimeError):
    pass


class MoveLimitError(FlingSimError):
    """fs_movep ran into its step limit (environment/exceptions.py MoveJointsException in the reference)."""


def load_library(build_if_missing=True):
    """dlopen libflingsim.so and declare every prototype of include/flingsim.h."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch first, when it is installed: this image's torch wheel bundles its own ROCm runtime, and a process in which
    # the system runtime (linked by libflingsim) initialised the GPU before torch did cannot use torch.cuda afterwards
    # ("No HIP GPUs are available").  The device-resident stages (observe, prepare_image, value network, action selection)
    # hand torch tensors to this library, so both runtimes end up in one process.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("FLINGSIM_LIB", _build.LIB)  # development override: an alternative build of the library
    if not os.path.exists(path):
        if not build_if_missing:
            raise FlingSimError(f"{path} is missing: run `python -m flingbot_amd.build`")
        _build.build_lib()
    lib = C.CDLL(path)
    fp, ip, vp, ci, cf = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_float
    u8p = C.POINTER(C.c_ubyte)
    proto = {
        "fs_last_error": (C.c_char_p, []),
        "fs_version": (ci, []),
        "fs_create": (vp, [ci, ci, ci, ci]),
        "fs_destroy": (None, [vp]),
        "fs_n_envs": (ci, [vp]),
        "fs_set_solver": (ci, [vp, ci]),
        "fs_get_solver": (ci, [vp]),
        "fs_fused_fits": (ci, [vp, ci]),
        "fs_last_kernel_form": (ci, [vp]),
        "fs_last_boundary_form": (ci, [vp]),
        "fs_set_stream_groups": (ci, [vp, ci]),
        "fs_last_stream_groups": (ci, [vp]),
        "fs_set_scene": (ci, [vp, ci, fp, ci, fp, ci, ip, ci, ip, ci, ip, ci, ip, ci]),
        "fs_step": (ci, [vp, ci, ci]),
        "fs_step_list": (ci, [vp, ci, ip, ci]),
        "fs_sync": (ci, [vp]),
        "fs_stream": (vp, [vp]),
        "fs_n_particles": (ci, [vp, ci]),
        "fs_n_springs": (ci, [vp, ci]),
        "fs_n_triangles": (ci, [vp, ci]),
        "fs_n_shapes": (ci, [vp, ci]),
        "fs_get_positions": (ci, [vp, ci, fp, ci]),
        "fs_set_positions": (ci, [vp, ci, fp, ci]),
        "fs_get_velocities": (ci, [vp, ci, fp, ci]),
        "fs_set_velocities": (ci, [vp, ci, fp, ci]),
        "fs_get_phases": (ci, [vp, ci, ip, ci]),
        "fs_set_phases": (ci, [vp, ci, ip, ci]),
        "fs_get_rest_positions": (ci, [vp, ci, fp, ci]),
        "fs_get_normals": (ci, [vp, ci, fp, ci]),
        "fs_get_edges": (ci, [vp, ci, ip, ci]),
        "fs_get_faces": (ci, [vp, ci, ip, ci]),
        "fs_get_spring_lengths": (ci, [vp, ci, fp, ci]),
        "fs_get_spring_stiffness": (ci, [vp, ci, fp, ci]),
        "fs_get_params": (ci, [vp, ci, fp, ci]),
        "fs_set_params": (ci, [vp, ci, fp, ci]),
        "fs_get_scene_bounds": (ci, [vp, ci, fp, fp]),
        "fs_add_sphere": (ci, [vp, ci, cf, fp, fp]),
        "fs_clear_shapes": (ci, [vp, ci]),
        "fs_get_shape_states": (ci, [vp, ci, fp, ci]),
        "fs_set_shape_states": (ci, [vp, ci, fp, ci]),
        "fs_get_camera_params": (ci, [vp, ci, fp]),
        "fs_set_camera_params": (ci, [vp, ci, fp]),
        "fs_render": (ci, [vp, ci, u8p, ci, fp, ci]),
        "fs_get_sphere_mesh": (ci, [vp, ci, fp, fp, ci, ip, ci]),
        "fs_coverage": (ci, [vp, C.POINTER(C.c_double), ci]),
        "fs_step_timed": (ci, [vp, ci, ci, fp]),
        "fs_picker_reset": (ci, [vp, ci, C.c_double, C.c_double]),
        "fs_picker_set_radius": (ci, [vp, ci, C.c_double]),
        "fs_last_movep_steps": (C.c_longlong, [vp]),
        "fs_advance_timing": (ci, [vp, C.POINTER(C.c_double)]),
        "fs_pool_stats": (ci, [vp, C.POINTER(C.c_longlong)]),
        "fs_advance_begin": (ci, [vp, ci, ip, ip, C.POINTER(C.c_double), ip, C.POINTER(C.c_double), ip, ip, ip, ip, C.c_double,
                                  C.POINTER(C.c_double), ci, ci, ip, ip, ip]),
        "fs_advance_end": (ci, [vp, ci, ip, ip, ip]),
        "fs_advance_in_flight": (ci, [vp]),
        "fs_service_lane": (ci, [vp, ci]),
        "fs_set_scene_prebuilt": (ci, [vp, ci, vp]),
        "fs_picker_get_picked": (ci, [vp, ci, ip, ci]),
        "fs_movep": (ci, [vp, ci, C.POINTER(C.c_double), ip, C.c_double, ci, ci, C.c_double, ip]),
        "fs_movep_batch": (ci, [vp, ci, ip, C.POINTER(C.c_double), ip, C.c_double, ci, ci, C.c_double, ip]),
        "fs_movep_batch_f32": (ci, [vp, ci, ip, fp, ip, C.c_double, ci, ci, C.c_double, ip]),
        "fs_advance": (ci, [vp, ci, ip, ip, C.POINTER(C.c_double), ip, C.POINTER(C.c_double), ip, ip, ip, ip, C.c_double,
                            C.POINTER(C.c_double), ci, ci, ip, ip, ip]),
        "fs_wait_until_stable": (ci, [vp, ci, ip, ci, C.c_double, ip, ip]),
        "fs_cloth_stats": (ci, [vp, ci, ip, fp, ci]),
        "fs_stretch_probe": (ci, [vp, ci, ip, fp, fp, ip, fp]),
        "fs_snapshot_positions": (ci, [vp, ci, ip]),
        "fs_max_displacement": (ci, [vp, ci, ip, fp, ci]),
        "fs_set_particles": (ci, [vp, ci, ip, ip, fp, ci]),
        "fs_prepare_image_work_bytes": (C.c_size_t, [ci, ci, ci]),
        "fs_prepare_image": (ci, [vp, ci, ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), ci,
                                  vp, vp, vp]),
        "fs_select_action_work_bytes": (C.c_size_t, [ci]),
        "fs_select_action": (ci, [vp, ci, ip, ci, ci, ci, ci, ci, C.POINTER(C.c_double), vp, ci, C.c_double,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double,
                                  C.c_double, C.POINTER(C.c_longlong), fp, vp, vp]),
        "fs_observe_work_bytes": (C.c_size_t, [ci]),
        "fs_observe": (ci, [vp, ci, ci, vp, vp, ip, vp]),
        "fs_observe_batch": (ci, [vp, ci, ip, ci, vp, vp, ip, vp]),
        "fs_value_net_param_floats": (C.c_size_t, []),
        "fs_value_net_work_bytes": (C.c_size_t, [ci, ci]),
        "fs_value_net_pack": (ci, [ci, fp, fp, fp, fp, fp, fp, fp, fp]),
        "fs_value_net_forward": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]),
        "fs_eval_rsqrt": (ci, [vp, fp, fp, ci]),
        "fs_timer_start": (ci, [vp]),
        "fs_timer_stop": (ci, [vp, fp]),
        "fs_host_scene_build": (vp, [fp, ci, fp, ci, ip, ci, ip, ci, ip, ci, ip, ci]),
        "fs_host_scene_free": (None, [vp]),
        "fs_host_scene_counts": (ci, [vp, ip, ip, ip, ip]),
        "fs_host_scene_copy": (ci, [vp, ci, vp, ci]),
        "fs_host_sphere_mesh": (ci, [cf, fp, fp, fp, fp, ip]),
        "fs_camera_matrices": (ci, [fp, fp, ci, ci, fp, fp, fp]),
        "fs_get_last_neighbors": (ci, [vp, ci, ip, ip]),
        "fs_get_last_shape_candidates": (ci, [vp, ci, ip]),
        "fs_device_positions": (vp, [vp, ci]),
        "fs_device_key": (ci, [vp, C.c_char_p, ci]),
        "fs_tenants_register": (ci, [C.c_char_p]),
        "fs_tenants_count": (ci, [C.c_char_p, ci]),
        "fs_tenants_unregister": (ci, [C.c_char_p]),
    }
    unbound = []
    for name, (res, args) in proto.items():
        if "FLINGSIM_LIB" in os.environ and not hasattr(lib, name):
            unbound.append(name)   # development override with an older build of the library (A/B timing): entry points it
            continue               # predates stay unbound -- and are named below, so a stale or mistyped override is seen
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if unbound:
        import warnings
        warnings.warn(f"FLINGSIM_LIB={os.environ['FLINGSIM_LIB']}: this build lacks {len(unbound)} C-ABI entries of "
                      f"include/flingsim.h ({', '.join(unbound)}); calls to them will fail, and its results may differ from "
                      f"this tree's oracle", RuntimeWarning, stacklevel=2)
    lib._fs_symbols = tuple(proto)
    _lib = lib
    return lib


def _f(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).ravel())


def _i(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel())


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class FlingSim:
    """A context of `n_envs` independent cloth episodes stepped in one batched launch."""

    def __init__(self, n_envs=1, device=0, camera_width=720, camera_height=720, solver=FS_SOLVER_AUTO):
        self.lib = load_library()
        self.h = self.lib.fs_create(int(device), int(n_envs), int(camera_width), int(camera_height))
        if not self.h:
            raise FlingSimError("fs_create failed: " + self.lib.fs_last_error().decode())
        self.n_envs = int(n_envs)
        self.device = int(device)
        self.set_solver(solver)

    def close(self):
        if getattr(self, "h", None):
            self.lib.fs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise FlingSimError(self.lib.fs_last_error().decode())
        return rc

    def set_solver(self, solver):
        self._ck(self.lib.fs_set_solver(self.h, int(solver)))

    def device_key(self):
        """fs_device_key: the PCI bus id of the context's device -- the same string in every process that drives this GPU."""
        buf = C.create_string_buffer(64)
        self._ck(self.lib.fs_device_key(self.h, buf, 64))
        return buf.value.decode()

    def last_kernel_form(self):
        """FS_FORM_* of the most recent solver launch (white box for the parity tests)."""
        return self._ck(self.lib.fs_last_kernel_form(self.h))

    def last_boundary_form(self):
        """0 four kernels / 1 fs_k_boundary in the most recent streaming launch (white box)."""
        return self._ck(self.lib.fs_last_boundary_form(self.h))

    def set_stream_groups(self, groups):
        """Concurrent launch chains of the streaming back-end: 0 = measured default, 1..4 = forced (fs_set_stream_groups)."""
        self._ck(self.lib.fs_set_stream_groups(self.h, int(groups)))

    def last_stream_groups(self):
        return self._ck(self.lib.fs_last_stream_groups(self.h))

    def env(self, i=0):
        return EnvView(self, i)

    # ---- batched
    def step(self, n_steps=1, env=-1):
        self._ck(self.lib.fs_step(self.h, int(env), int(n_steps)))

    def step_list(self, envs, n_steps=1):
        """Advance exactly the listed episodes (one batched launch sequence)."""
        ids = _i(envs)
        if ids.size:
            self._ck(self.lib.fs_step_list(self.h, ids.size, _ip(ids), int(n_steps)))

    def sync(self):
        self._ck(self.lib.fs_sync(self.h))

    def stream(self):
        return self.lib.fs_stream(self.h)

    def coverage(self):
        out = np.empty(self.n_envs, np.float64)
        self._ck(self.lib.fs_coverage(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
        return out

    def step_timed(self, n_steps=1, env=-1):
        """fs_step bracketed by HIP events on the context's stream -> elapsed device milliseconds."""
        ms = C.c_float(0.0)
        self._ck(self.lib.fs_step_timed(self.h, int(env), int(n_steps), C.byref(ms)))
        return float(ms.value)

    # ---- on-device picker / movep (include/flingsim.h)
    def picker_reset(self, env, picker_threshold=0.005, particle_radius=0.00625, picker_radius=None):
        """Picker.reset's bookkeeping.  picker_radius: the reference's python-float Picker.picker_radius, which enters the
        grasp threshold as a double (flex_utils.py:154-155); None keeps the float32 radius of shape 0."""
        self._ck(self.lib.fs_picker_reset(self.h, int(env), float(picker_threshold), float(particle_radius)))
        if picker_radius is not None:
            self._ck(self.lib.fs_picker_set_radius(self.h, int(env), float(picker_radius)))

    def picked(self, env):
        out = np.full(self.n_shapes(env), -1, np.int32)
        self._ck(self.lib.fs_picker_get_picked(self.h, int(env), _ip(out), out.size))
        return out

    def movep(self, envs, targets, grasp, speed=0.1, limit=1000, min_steps=None, eps=1e-4):
        """SimEnv.movep for one episode (envs = int) or a batch (envs = sequence).  targets [n,S,3], grasp [n,S].
        A float32 `targets` array keeps movep's arithmetic in float32, exactly as numpy does in the reference when
        stretch_cloth passes float32 picker positions; anything else is float64 (python lists of floats).
        Returns the iteration counts; raises MoveLimitError like the reference's MoveJointsException."""
        single = np.isscalar(envs)
        ids = _i([envs] if single else envs)
        arr = np.asarray(targets)
        gr = _i(np.asarray(grasp).astype(np.int32).reshape(ids.size, -1))
        iters = np.zeros(ids.size, np.int32)
        ms = -1 if min_steps is None else int(min_steps)
        if arr.dtype == np.float32:
            tg = np.ascontiguousarray(arr.reshape(ids.size, -1, 3))
            rc = self.lib.fs_movep_batch_f32(self.h, ids.size, _ip(ids), _fp(tg), _ip(gr), float(speed), int(limit), ms,
                                             float(eps), _ip(iters))
        else:
            tg = np.ascontiguousarray(arr.astype(np.float64).reshape(ids.size, -1, 3))
            rc = self.lib.fs_movep_batch(self.h, ids.size, _ip(ids), tg.ctypes.data_as(C.POINTER(C.c_double)), _ip(gr),
                                         float(speed), int(limit), ms, float(eps), _ip(iters))
        self.last_movep_steps = int(self.lib.fs_last_movep_steps(self.h))  # simulation steps (<= iterations), all episodes
        if rc == -4:
            raise MoveLimitError(self.lib.fs_last_error().decode())
        self._ck(rc)
        return int(iters[0]) if single else iters

    def wait_until_stable(self, envs, max_steps=300, tolerance=1e-2):
        """flex_utils.wait_until_stable (flex_utils.py:430-441) for one episode or a batch, looped on the device.
        Returns (stable, steps): the reference's return value and the number of simulation steps taken, per episode."""
        single = np.isscalar(envs)
        ids = _i([envs] if single else envs)
        steps, stable = np.zeros(ids.size, np.int32), np.zeros(ids.size, np.int32)
        self._ck(self.lib.fs_wait_until_stable(self.h, ids.size, _ip(ids), int(max_steps), float(tolerance), _ip(steps),
                                               _ip(stable)))
        return (bool(stable[0]), int(steps[0])) if single else (stable.astype(bool), steps)

    def advance(self, envs, kind, targets, grasp, speed, limit, min_steps, f32, start, cap_min=8, cap=64, eps=1e-4,
                tolerance=1e-2):
        """One chunk of simulation for episodes in DIFFERENT phases of their primitives (fs_advance): kind[a] 0 = movep
        towards targets[a] ([S,3]) with grasp[a], speed[a], iteration limit[a], min_steps[a] (None -> -1), f32[a] (the
        caller's targets were float32), resumed at loop iteration start[a]; 1 = wait_until_stable with max_steps =
        limit[a], start[a] steps already taken, tolerance (scalar or per entry); 2 = limit[a] plain steps.  Returns
        (progress, status, steps) int32 arrays: status 0 = call again with start = progress, 1 = finished, 2 = finished
        at the limit."""
        ids, kd = _i(envs), _i(kind)
        n = ids.size
        tg = np.ascontiguousarray(np.asarray(targets, np.float64).reshape(n, -1))
        gr = _i(np.asarray(grasp).astype(np.int32).reshape(n, -1))
        sp = np.ascontiguousarray(np.asarray(speed, np.float64).reshape(n))
        lim, ms, f3, st = _i(limit), _i(min_steps), _i(f32), _i(start)
        prog, status, steps = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        tol = np.ascontiguousarray(np.broadcast_to(np.asarray(tolerance, np.float64), (n,)))
        dp = C.POINTER(C.c_double)
        self._ck(self.lib.fs_advance(self.h, n, _ip(ids), _ip(kd), tg.ctypes.data_as(dp), _ip(gr), sp.ctypes.data_as(dp),
                                     _ip(lim), _ip(ms), _ip(f3), _ip(st), float(eps), tol.ctypes.data_as(dp), int(cap_min), int(cap),
                                     _ip(prog), _ip(status), _ip(steps)))
        return prog, status, steps

    def advance_begin(self, envs, kind, targets, grasp, speed, limit, min_steps, f32, start, cap_min=8, cap=64, eps=1e-4,
                      tolerance=1e-2):
        """fs_advance_begin: queue the chunk and return at once -> (ticket, progress, status, steps).  The movep entries'
        results are final; a wait / step entry has status -1 until advance_end(ticket) and may pass start = -1 ("continue
        from the loop state the device keeps") so that the next chunk can be queued before the previous one has reported."""
        ids, kd = _i(envs), _i(kind)
        n = ids.size
        tg = np.ascontiguousarray(np.asarray(targets, np.float64).reshape(n, -1))
        gr = _i(np.asarray(grasp).astype(np.int32).reshape(n, -1))
        sp = np.ascontiguousarray(np.asarray(speed, np.float64).reshape(n))
        lim, ms, f3, st = _i(limit), _i(min_steps), _i(f32), _i(start)
        prog, status, steps = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        tol = np.ascontiguousarray(np.broadcast_to(np.asarray(tolerance, np.float64), (n,)))
        dp = C.POINTER(C.c_double)
        ticket = self._ck(self.lib.fs_advance_begin(self.h, n, _ip(ids), _ip(kd), tg.ctypes.data_as(dp), _ip(gr), sp.ctypes.data_as(dp),
                                                    _ip(lim), _ip(ms), _ip(f3), _ip(st), float(eps), tol.ctypes.data_as(dp),
                                                    int(cap_min), int(cap), _ip(prog), _ip(status), _ip(steps)))
        return ticket, prog, status, steps

    def advance_end(self, ticket, prog, status, steps):
        """fs_advance_end: wait for the ticket's launches; fills the wait / step entries of the arrays advance_begin returned."""
        self._ck(self.lib.fs_advance_end(self.h, int(ticket), _ip(prog), _ip(status), _ip(steps)))
        return prog, status, steps

    def advance_in_flight(self):
        return self._ck(self.lib.fs_advance_in_flight(self.h))

    def service_lane(self, on):
        """fs_service_lane: True -> every call of this context runs on the high-priority service stream (for episodes that
        are not part of a chunk in flight) until service_lane(False)."""
        self._ck(self.lib.fs_service_lane(self.h, 1 if on else 0))

    def last_movep_steps_raw(self):
        """fs_last_movep_steps: simulation steps of the movep entries of the most recent fs_movep* / fs_advance* call."""
        return int(self.lib.fs_last_movep_steps(self.h))

    def advance_timing(self):
        """fs_advance's stopwatch since the context was created: dict(calls, sequences, wall_ms, gpu_ms, prep_ms)."""
        out = np.zeros(5, np.float64)
        self._ck(self.lib.fs_advance_timing(self.h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return dict(calls=int(out[0]), sequences=int(out[1]), wall_ms=float(out[2]), gpu_ms=float(out[3]), prep_ms=float(out[4]))

    def pool_stats(self):
        """fs_pool_stats: dict(idle_bytes, idle_buffers, idle_limit) of the context's buffer pool."""
        out = (C.c_longlong * 3)()
        self._ck(self.lib.fs_pool_stats(self.h, out))
        return dict(idle_bytes=int(out[0]), idle_buffers=int(out[1]), idle_limit=int(out[2]))

    def cloth_stats(self, envs):
        """[n,3] float32: min height, max height, max |velocity component| per episode (device reductions)."""
        ids = _i([envs] if np.isscalar(envs) else envs)
        out = np.empty((ids.size, 3), np.float32)
        self._ck(self.lib.fs_cloth_stats(self.h, ids.size, _ip(ids), _fp(out), out.size))
        return out

    def stretch_probe(self, envs, midpoints_xz, height_thr):
        """stretch_cloth's probe (simEnv.py:155-168) per episode: (single_grasp bool[n], nearest float32[n,3])."""
        ids = _i([envs] if np.isscalar(envs) else envs)
        mid = np.ascontiguousarray(np.asarray(midpoints_xz, np.float32).reshape(ids.size, 2))
        thr = np.ascontiguousarray(np.asarray(height_thr, np.float32).reshape(ids.size))
        single, near = np.zeros(ids.size, np.int32), np.empty((ids.size, 3), np.float32)
        self._ck(self.lib.fs_stretch_probe(self.h, ids.size, _ip(ids), _fp(mid), _fp(thr), _ip(single), _fp(near)))
        return single.astype(bool), near

    def snapshot_positions(self, envs):
        """SimEnv.preaction (simEnv.py:464-465): keep the current positions on the device."""
        ids = _i([envs] if np.isscalar(envs) else envs)
        self._ck(self.lib.fs_snapshot_positions(self.h, ids.size, _ip(ids)))

    def max_displacement(self, envs):
        """float32[n]: largest particle displacement since snapshot_positions (simEnv.py:470-472)."""
        ids = _i([envs] if np.isscalar(envs) else envs)
        out = np.empty(ids.size, np.float32)
        self._ck(self.lib.fs_max_displacement(self.h, ids.size, _ip(ids), _fp(out), out.size))
        return out

    def set_particles(self, envs, particle_ids, pos4, zero_velocity=True):
        """One particle per episode: position + inverse mass (float32 [n,4]), velocity optionally zeroed."""
        ids = _i([envs] if np.isscalar(envs) else envs)
        pids = _i(np.asarray(particle_ids).reshape(-1))
        p4 = np.ascontiguousarray(np.asarray(pos4, np.float32).reshape(ids.size, 4))
        self._ck(self.lib.fs_set_particles(self.h, ids.size, _ip(ids), _ip(pids), _fp(p4), 1 if zero_velocity else 0))

    def timer_start(self):
        self._ck(self.lib.fs_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float(0.0)
        self._ck(self.lib.fs_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    # ---- per env, pyflex-shaped
    def set_scene(self, env, scene_params, vertices=(), stretch_edges=(), bend_edges=(), shear_edges=(), faces=()):
        sp = _f(scene_params)
        v, st, be, sh, fa = _f(vertices), _i(stretch_edges), _i(bend_edges), _i(shear_edges), _i(faces)
        self._ck(self.lib.fs_set_scene(self.h, env, _fp(sp), sp.size, _fp(v), v.size, _ip(st), st.size, _ip(be),
                                       be.size, _ip(sh), sh.size, _ip(fa), fa.size))

    def set_scene_prebuilt(self, env, scene):
        """fs_set_scene from a PrebuiltScene (host part done earlier, possibly on another thread); consumes it."""
        self._ck(self.lib.fs_set_scene_prebuilt(self.h, env, scene.take()))
        scene.free()

    def n_particles(self, env=0):
        return self._ck(self.lib.fs_n_particles(self.h, env))

    def n_springs(self, env=0):
        return self._ck(self.lib.fs_n_springs(self.h, env))

    def n_triangles(self, env=0):
        return self._ck(self.lib.fs_n_triangles(self.h, env))

    def n_shapes(self, env=0):
        return self._ck(self.lib.fs_n_shapes(self.h, env))

    def _getf(self, fn, env, size):
        out = np.empty(size, np.float32)
        self._ck(getattr(self.lib, fn)(self.h, env, _fp(out), out.size))
        return out

    def _geti(self, fn, env, size):
        out = np.empty(size, np.int32)
        self._ck(getattr(self.lib, fn)(self.h, env, _ip(out), out.size))
        return out

    def get_positions(self, env=0):
        return self._getf("fs_get_positions", env, 4 * self.n_particles(env))

    def set_positions(self, env, p):
        p = _f(p)
        self._ck(self.lib.fs_set_positions(self.h, env, _fp(p), p.size))

    def get_velocities(self, env=0):
        return self._getf("fs_get_velocities", env, 3 * self.n_particles(env))

    def set_velocities(self, env, v):
        v = _f(v)
        self._ck(self.lib.fs_set_velocities(self.h, env, _fp(v), v.size))

    def get_phases(self, env=0):
        return self._geti("fs_get_phases", env, self.n_particles(env))

    def set_phases(self, env, ph):
        ph = _i(ph)
        self._ck(self.lib.fs_set_phases(self.h, env, _ip(ph), ph.size))

    def get_restPositions(self, env=0):
        return self._getf("fs_get_rest_positions", env, 4 * self.n_particles(env))

    def get_normals(self, env=0):
        return self._getf("fs_get_normals", env, 4 * self.n_particles(env))

    def get_edges(self, env=0):
        return self._geti("fs_get_edges", env, 2 * self.n_springs(env))

    def get_faces(self, env=0):
        return self._geti("fs_get_faces", env, 3 * self.n_triangles(env))

    def get_spring_lengths(self, env=0):
        return self._getf("fs_get_spring_lengths", env, self.n_springs(env))

    def get_spring_stiffness(self, env=0):
        return self._getf("fs_get_spring_stiffness", env, self.n_springs(env))

    def get_params(self, env=0):
        return self._getf("fs_get_params", env, 32)

    def set_params(self, env, table):
        table = _f(table)
        self._ck(self.lib.fs_set_params(self.h, env, _fp(table), table.size))

    def get_scene_bounds(self, env=0):
        lo, up = np.empty(3, np.float32), np.empty(3, np.float32)
        self._ck(self.lib.fs_get_scene_bounds(self.h, env, _fp(lo), _fp(up)))
        return lo, up

    def add_sphere(self, env, radius, pos, quat):
        pos, quat = _f(pos), _f(quat)
        self._ck(self.lib.fs_add_sphere(self.h, env, C.c_float(radius), _fp(pos), _fp(quat)))

    def clear_shapes(self, env=0):
        self._ck(self.lib.fs_clear_shapes(self.h, env))

    def get_shape_states(self, env=0):
        return self._getf("fs_get_shape_states", env, 14 * self.n_shapes(env))

    def set_shape_states(self, env, s):
        s = _f(s)
        self._ck(self.lib.fs_set_shape_states(self.h, env, _fp(s), s.size))

    def get_camera_params(self, env=0):
        out = np.empty(8, np.float32)
        self._ck(self.lib.fs_get_camera_params(self.h, env, _fp(out)))
        return out

    def set_camera_params(self, env, p):
        p = _f(p)
        assert p.size >= 8
        self._ck(self.lib.fs_set_camera_params(self.h, env, _fp(p)))

    def render(self, env=0):
        w, h = self.get_camera_params(env)[:2].astype(int)
        rgba = np.empty(int(w) * int(h) * 4, np.uint8)
        depth = np.empty(int(w) * int(h), np.float32)
        self._ck(self.lib.fs_render(self.h, env, rgba.ctypes.data_as(C.POINTER(C.c_ubyte)), rgba.size, _fp(depth),
                                    depth.size))
        return rgba, depth

    def sphere_mesh(self, env=0):
        """(verts float32[441 S, 4], normals float32[441 S, 4], tris int32[800 S, 3]): the picker meshes `render`
        rasterises (white-box access for tests)."""
        s_ = self.n_shapes(env)
        verts, nrms = np.empty((441 * s_, 4), np.float32), np.empty((441 * s_, 4), np.float32)
        tris = np.empty((800 * s_, 3), np.int32)
        self._ck(self.lib.fs_get_sphere_mesh(self.h, env, _fp(verts), _fp(nrms), verts.size, _ip(tris), tris.size))
        return verts, nrms, tris

    def observe(self, env, image_dim, want_mask=False):
        """get_image + get_cloth_mask + preprocess_obs on the device (fs_observe, csrc/fs_observe.hip): renders episode
        `env` with its camera and returns (obs float32 CUDA tensor [4, S, S], bbox int[5] = x.min, x.max, y.min, y.max and
        pixel count of the largest cloth component (-1 / 0 when there is none)[, mask uint8 CUDA tensor [S, S]])."""
        import torch
        s_ = int(image_dim)
        dev = torch.device("cuda", self.device)
        obs = torch.empty((4, s_, s_), dtype=torch.float32, device=dev)
        mask = torch.empty((s_, s_), dtype=torch.uint8, device=dev) if want_mask else None
        nbytes = int(self.lib.fs_observe_work_bytes(s_))
        work = getattr(self, "_observe_work", None)
        if work is None or work.numel() < nbytes:
            work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._observe_work = work
        torch.cuda.current_stream(dev).synchronize()  # the buffers above may still be in use on torch's stream
        bbox = np.zeros(5, np.int32)
        self._ck(self.lib.fs_observe(self.h, int(env), s_, C.c_void_p(obs.data_ptr()),
                                     C.c_void_p(mask.data_ptr()) if want_mask else None, _ip(bbox),
                                     C.c_void_p(work.data_ptr())))
        return (obs, bbox, mask) if want_mask else (obs, bbox)

    def observe_batch(self, envs, image_dim, want_mask=False):
        """observe() for several episodes with the host round trips of one call (fs_observe_batch): returns
        (obs float32 CUDA [n, 4, S, S], bbox int32 [n, 5][, mask uint8 CUDA [n, S, S]])."""
        import torch
        ids = _i(envs)
        n, s_ = int(ids.size), int(image_dim)
        dev = torch.device("cuda", self.device)
        obs = torch.empty((n, 4, s_, s_), dtype=torch.float32, device=dev)
        mask = torch.empty((n, s_, s_), dtype=torch.uint8, device=dev) if want_mask else None
        bbox = np.zeros((n, 5), np.int32)
        if n == 0:
            return (obs, bbox, mask) if want_mask else (obs, bbox)
        nbytes = int(self.lib.fs_observe_work_bytes(s_)) * n
        work = getattr(self, "_observe_work", None)
        if work is None or work.numel() < nbytes:
            work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._observe_work = work
        torch.cuda.current_stream(dev).synchronize()  # the buffers above may still be in use on torch's stream
        self._ck(self.lib.fs_observe_batch(self.h, n, _ip(ids), s_, C.c_void_p(obs.data_ptr()),
                                           C.c_void_p(mask.data_ptr()) if want_mask else None, _ip(bbox),
                                           C.c_void_p(work.data_ptr())))
        return (obs, bbox, mask) if want_mask else (obs, bbox)

    def eval_rsqrt(self, x):
        """The constraint kernels' reciprocal square root on the device for a float32 array (white box: fs_eval_rsqrt)."""
        x = np.ascontiguousarray(np.asarray(x, np.float32).ravel())
        y = np.empty_like(x)
        self._ck(self.lib.fs_eval_rsqrt(self.h, _fp(x), _fp(y), x.size))
        return y

    def get_last_neighbors(self, env=0):
        n = self.n_particles(env)
        counts = np.empty(n, np.int32)
        lists = np.empty(n * 96, np.int32)
        self._ck(self.lib.fs_get_last_neighbors(self.h, env, _ip(counts), _ip(lists)))
        return counts, lists.reshape(n, 96)

    def get_last_shape_candidates(self, env=0):
        """collideShapes of the last substep: per particle, bit q = plane q, bit 8 + q = kinematic sphere q."""
        masks = np.empty(self.n_particles(env), np.int32)
        self._ck(self.lib.fs_get_last_shape_candidates(self.h, env, _ip(masks)))
        return masks.view(np.uint32)


class EnvView:
    """One episode of a FlingSim with the reference `pyflex` call names (no env argument)."""

    def __init__(self, sim, env):
        self.sim, self.e = sim, int(env)

    def set_scene(self, scene_params, vertices=(), stretch_edges=(), bend_edges=(), shear_edges=(), faces=()):
        self.sim.set_scene(self.e, scene_params, vertices, stretch_edges, bend_edges, shear_edges, faces)

    def step(self, n=1):
        self.sim.step(n, env=self.e)

    @property
    def n(self):
        return self.sim.n_particles(self.e)

    def get_n_particles(self):
        return self.sim.n_particles(self.e)

    def get_n_shapes(self):
        return self.sim.n_shapes(self.e)

    def __getattr__(self, name):
        fn = getattr(self.sim, name)
        if name.startswith("get_") or name in ("clear_shapes", "render"):
            return lambda: fn(self.e)
        return lambda *a: fn(self.e, *a)


# ---- host-only helpers (no GPU needed): scene builder and camera set-up of the C-ABI
SCENE_ARRAYS = {"positions": (0, np.float32), "velocities": (1, np.float32), "phases": (2, np.int32),
                "springs": (3, np.int32), "spring_lengths": (4, np.float32), "spring_stiffness": (5, np.float32),
                "triangles": (6, np.int32), "tri_normals": (7, np.float32), "adj_offsets": (8, np.int32),
                "adj_neighbors": (9, np.int32), "bounds": (10, np.float32), "params": (11, np.float32)}


class PrebuiltScene:
    """The host half of fs_set_scene (topology, adjacency tables, spring codes: fs_host_scene_build), built without the GPU --
    on any thread: ctypes releases the GIL for the call -- and handed to FlingSim.set_scene_prebuilt later."""

    def __init__(self, scene_params, vertices=(), stretch_edges=(), bend_edges=(), shear_edges=(), faces=()):
        self.lib = load_library()
        sp = _f(scene_params)
        v, st, be, sh, fa = _f(vertices), _i(stretch_edges), _i(bend_edges), _i(shear_edges), _i(faces)
        self.h = self.lib.fs_host_scene_build(_fp(sp), sp.size, _fp(v), v.size, _ip(st), st.size, _ip(be), be.size, _ip(sh),
                                              sh.size, _ip(fa), fa.size)
        if not self.h:
            raise FlingSimError("fs_host_scene_build failed: " + self.lib.fs_last_error().decode())

    def take(self):
        if not self.h:
            raise FlingSimError("PrebuiltScene: already consumed")
        return self.h

    def free(self):
        if getattr(self, "h", None):
            self.lib.fs_host_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def host_sphere_mesh(radius, prev_pos, prev_quat):
    """The mesh `render` rasterises for one kinematic sphere, from the library's host code (no GPU needed)."""
    lib = load_library()
    verts, nrms, tris = np.empty((441, 4), np.float32), np.empty((441, 4), np.float32), np.empty((800, 3), np.int32)
    rc = lib.fs_host_sphere_mesh(C.c_float(radius), _fp(_f(prev_pos)), _fp(_f(prev_quat)), _fp(verts), _fp(nrms), _ip(tris))
    assert rc == 0
    return verts, nrms, tris


def host_scene(scene_params, vertices=(), stretch_edges=(), bend_edges=(), shear_edges=(), faces=()):
    """Run the product's host scene builder (fs_host_scene_build) and return every array as numpy."""
    lib = load_library()
    sp = _f(scene_params)
    v, st, be, sh, fa = _f(vertices), _i(stretch_edges), _i(bend_edges), _i(shear_edges), _i(faces)
    h = lib.fs_host_scene_build(_fp(sp), sp.size, _fp(v), v.size, _ip(st), st.size, _ip(be), be.size, _ip(sh), sh.size,
                                _ip(fa), fa.size)
    if not h:
        raise FlingSimError(lib.fs_last_error().decode())
    try:
        cnt = [C.c_int(0) for _ in range(4)]
        lib.fs_host_scene_counts(h, *[C.byref(c) for c in cnt])
        n, m, t, deg = [c.value for c in cnt]
        sizes = {"positions": 4 * n, "velocities": 3 * n, "phases": n, "springs": 2 * m, "spring_lengths": m,
                 "spring_stiffness": m, "triangles": 3 * t, "tri_normals": 3 * t, "adj_offsets": n + 1,
                 "adj_neighbors": 2 * m, "bounds": 6, "params": 32}
        out = {"n": n, "m": m, "t": t, "max_deg": deg}
        for name, code, size in (("restnear", 12, 8 * n), ("stream_codes", 13, 4 * n), ("stream_dict", 14, 4 * 256)):
            a = np.full(max(size, 1), 0xffffffff, np.uint32)  # derived tables of the kernels (empty when the cloth has none)
            if lib.fs_host_scene_copy(h, code, a.ctypes.data_as(C.c_void_p), a.size) < 0:
                raise FlingSimError(lib.fs_last_error().decode())
            out[name] = a[:size]
        flags = np.zeros(4, np.int32)  # FS_SCENE_FLAGS: what the host decided about the derived tables
        if lib.fs_host_scene_copy(h, 15, flags.ctypes.data_as(C.c_void_p), flags.size) < 0:
            raise FlingSimError(lib.fs_last_error().decode())
        out["flags"] = dict(restnear_ok=int(flags[0]), g64_ok=int(flags[1]), gp_L_ok=int(flags[2]), gp_halvable=int(flags[3]))
        for name, (code, dt) in SCENE_ARRAYS.items():
            a = np.zeros(max(sizes[name], 1), dt)
            rc = lib.fs_host_scene_copy(h, code, a.ctypes.data_as(C.c_void_p), a.size)
            if rc < 0:
                raise FlingSimError(lib.fs_last_error().decode())
            out[name] = a[:sizes[name]]
        return out
    finally:
        lib.fs_host_scene_free(h)


def camera_matrices(cam_pos, cam_angle, width, height, scene_lower, scene_upper):
    """fs_camera_matrices -> dict(view[4,4], proj[4,4], light[4,4], lightpos[3], lightdir[3]) (row-major)."""
    lib = load_library()
    a = [_f(x) for x in (cam_pos, cam_angle, scene_lower, scene_upper)]
    out = np.zeros(54, np.float32)
    rc = lib.fs_camera_matrices(_fp(a[0]), _fp(a[1]), int(width), int(height), _fp(a[2]), _fp(a[3]), _fp(out))
    if rc < 0:
        raise FlingSimError("fs_camera_matrices failed")
    return {"view": out[0:16].reshape(4, 4), "proj": out[16:32].reshape(4, 4), "light": out[32:48].reshape(4, 4),
            "lightpos": out[48:51], "lightdir": out[51:54]}
