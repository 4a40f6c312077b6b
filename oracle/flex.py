"""ctypes front-end of the C oracle (oracle/flex_oracle.c).  TEST INFRASTRUCTURE, not product code.

Mirrors the subset of the ``pyflex`` module surface (PyFlex/bindings/pyflex.cpp:1135-1208) the hot path uses so the
parity tests can drive the oracle and the HIP path with the same calls.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def oracle_lib_path():
    """liboracle_fma.so (fmaf inlined as the hardware instruction) when this CPU has FMA, else the libm-fmaf build.
    Both give the same bits.  FLINGSIM_ORACLE_LIB overrides (scripts/host_sanitizers.sh: an ASan / UBSan build of the same source)."""
    return os.environ.get("FLINGSIM_ORACLE_LIB") or os.path.join(_HERE, "liboracle_fma.so" if _cpu_has_fma() else "liboracle.so")


# bounding builds of the same step (oracle/Makefile): plain IEEE arithmetic, and each arithmetic choice alone.  Tests use
# them to measure how far the default oracle's spelled-out approximations move a step; they are never the parity oracle.
VARIANTS = ("exact", "exact_rsqrt", "nofma", "newton", "host")  # host: exact_rsqrt at -O3 -mfma = bench.py's CPU baseline
# sensitivity builds of the model-level [I] choices (flex_oracle.c "MODEL switches"): one alternative reading each, built on
# first use (liboracle_alt_<name>.so).  They measure how far another reading of the closed solver would move a trajectory
# (PARITY.md); they are never the parity oracle either.
MODEL_ALTERNATIVES = ("friction_post", "neighbors_by_distance", "shape_end_pose", "sleep_velocity_only", "sleep_at_predict",
                      "no_sleep", "apply_per_type", "damping_mult", "stiffness_iter", "shape_every_iteration", "contact_planes",
                      "count_candidates", "neighbors_at_start", "no_maxaccel", "maxaccel_per_frame", "maxaccel_position", "kinematic_velocity_kept")


def build_oracle(force=False):
    """Compile liboracle.so (gcc) and, when /root/reference is present, oracle/_ref."""
    srcs = [os.path.join(_HERE, f) for f in ("flex_oracle.c", "flex_oracle.h", "raster_oracle.c", "Makefile")]
    lib = oracle_lib_path()
    stale = force or not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs)
    stale = stale or any(not os.path.exists(os.path.join(_HERE, f"liboracle_{v}.so")) for v in VARIANTS)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so", "liboracle_fma.so"] +
                              [f"liboracle_{v}.so" for v in VARIANTS])
    if os.path.isdir("/root/reference/PyFlex/core") and os.path.exists(os.path.join(_HERE, "ref_camera_probe.cpp")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_ref/camera_ref"])
        if os.path.exists(os.path.join(_HERE, "ref_sphere_probe.cpp")):
            subprocess.check_call(["make", "-s", "-C", _HERE, "_ref/sphere_ref"])
    return lib


_libs = {}
_rsq_table = None


def rsqrt_table():
    """The chip's v_rsq_f32 as data: 2^24 two-bit fields (ulps from float32(1 / sqrt(float64(x))) + 2), four per byte, dumped
    on an MI355X by tests/golden/make_rsq_table.py.  Loaded once; every oracle library points at this buffer."""
    global _rsq_table
    if _rsq_table is None:
        with np.load(os.path.join(_HERE, "v_rsq_f32_gfx950.npz")) as z:
            assert str(z["arch"]) == "gfx950", "the table holds ONE chip's v_rsq_f32: it must be the architecture the kernels are built for"
            _rsq_table = np.ascontiguousarray(z["delta2bit"], np.uint8)
        assert _rsq_table.size == 1 << 22
    return _rsq_table


def eval_rsqrt(x):
    """The oracle's reciprocal square root (= the product's: v_rsq_f32(max(x, FLT_MIN)) of gfx950) for a float32 array."""
    x = np.ascontiguousarray(np.asarray(x, np.float32).ravel())
    y = np.empty_like(x)
    assert _load().orc_eval_rsqrt(_fp(x), _fp(y), x.size) == 0
    return y


def _load(variant=None):
    if variant in _libs:
        return _libs[variant]
    assert variant is None or variant in VARIANTS or (
        variant.startswith("alt_") and all(a in MODEL_ALTERNATIVES for a in variant[4:].split("+"))), variant   # "alt_a+b": both
    if variant == "host" and not _cpu_has_fma():
        variant = "exact_rsqrt"  # the same arithmetic through libm's fmaf
    path = oracle_lib_path() if variant is None else os.path.join(_HERE, f"liboracle_{variant}.so")
    if variant is not None and variant.startswith("alt_"):
        srcs = [os.path.join(_HERE, f) for f in ("flex_oracle.c", "flex_oracle.h", "Makefile")]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", _HERE, "-B", os.path.basename(path)])
    elif not os.path.exists(path):
        build_oracle()
    lib = C.CDLL(path)
    lib.orc_set_rsqrt_table.argtypes = [C.c_void_p]
    lib.orc_set_rsqrt_table(rsqrt_table().ctypes.data)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
    lib.orc_eval_rsqrt.argtypes = [fp, fp, C.c_int]
    lib.orc_create.restype = vp
    lib.orc_destroy.argtypes = [vp]
    lib.orc_set_scene.argtypes = [vp, fp, fp, C.c_int, ip, C.c_int, ip, C.c_int, ip, C.c_int, ip, C.c_int]
    lib.orc_step.argtypes = [vp, C.c_int]
    for name in ("orc_n_particles", "orc_n_springs", "orc_n_triangles", "orc_n_shapes"):
        getattr(lib, name).argtypes = [vp]
    for name in ("orc_get_positions", "orc_set_positions", "orc_get_velocities", "orc_set_velocities",
                 "orc_get_rest_positions", "orc_get_normals", "orc_get_spring_lengths", "orc_get_spring_stiffness",
                 "orc_get_params", "orc_get_shape_states", "orc_set_shape_states"):
        getattr(lib, name).argtypes = [vp, fp]
    for name in ("orc_get_phases", "orc_set_phases", "orc_get_edges", "orc_get_faces"):
        getattr(lib, name).argtypes = [vp, ip]
    lib.orc_get_scene_bounds.argtypes = [vp, fp, fp]
    lib.orc_add_sphere.argtypes = [vp, C.c_float, fp, fp]
    lib.orc_clear_shapes.argtypes = [vp]
    lib.orc_get_last_neighbors.argtypes = [vp, ip, ip]
    lib.orc_max_neighbor_list.argtypes = [vp]
    lib.orc_get_last_shape_candidates.argtypes = [vp, ip]
    lib.orc_missed_shape_contacts.argtypes = [vp]
    for name in ("orc_accel_clamps", "orc_degenerate_normals"):
        getattr(lib, name).argtypes, getattr(lib, name).restype = [vp], C.c_long
    _libs[variant] = lib
    return lib


def _f(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).ravel())


def _i(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel())


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class OracleSim:
    """One cloth episode on the CPU oracle, pyflex-shaped methods."""

    def __init__(self, variant=None):
        """variant: None = THE oracle; "exact" / "exact_rsqrt" / "nofma" = the arithmetic bounding builds (VARIANTS);
        "alt_<name>" = one model-level alternative (MODEL_ALTERNATIVES)."""
        self.lib = _load(variant)
        self.h = self.lib.orc_create()

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass

    # -- pyflex.set_scene (pyflex.cpp:229)
    def set_scene(self, scene_params, vertices=(), stretch_edges=(), bend_edges=(), shear_edges=(), faces=()):
        sp = _f(scene_params)
        assert sp.size >= 19
        v, st, be, sh, fa = _f(vertices), _i(stretch_edges), _i(bend_edges), _i(shear_edges), _i(faces)
        rc = self.lib.orc_set_scene(self.h, _fp(sp), _fp(v), v.size, _ip(st), st.size, _ip(be), be.size,
                                    _ip(sh), sh.size, _ip(fa), fa.size)
        assert rc == 0

    def step(self, n=1):
        assert self.lib.orc_step(self.h, int(n)) == 0

    @property
    def n(self):
        return self.lib.orc_n_particles(self.h)

    @property
    def m(self):
        return self.lib.orc_n_springs(self.h)

    @property
    def t(self):
        return self.lib.orc_n_triangles(self.h)

    def get_n_particles(self):
        return self.n

    def get_n_shapes(self):
        return self.lib.orc_n_shapes(self.h)

    def _getf(self, fn, size):
        out = np.empty(size, dtype=np.float32)
        getattr(self.lib, fn)(self.h, _fp(out))
        return out

    def _geti(self, fn, size):
        out = np.empty(size, dtype=np.int32)
        getattr(self.lib, fn)(self.h, _ip(out))
        return out

    def get_positions(self):
        return self._getf("orc_get_positions", 4 * self.n)

    def set_positions(self, p):
        p = _f(p)
        assert p.size >= 4 * self.n
        self.lib.orc_set_positions(self.h, _fp(p))

    def get_velocities(self):
        return self._getf("orc_get_velocities", 3 * self.n)

    def set_velocities(self, v):
        v = _f(v)
        assert v.size >= 3 * self.n
        self.lib.orc_set_velocities(self.h, _fp(v))

    def get_phases(self):
        return self._geti("orc_get_phases", self.n)

    def set_phases(self, ph):
        ph = _i(ph)
        self.lib.orc_set_phases(self.h, _ip(ph))

    def get_restPositions(self):
        return self._getf("orc_get_rest_positions", 4 * self.n)

    def get_normals(self):
        return self._getf("orc_get_normals", 4 * self.n)

    def get_edges(self):
        return self._geti("orc_get_edges", 2 * self.m)

    def get_faces(self):
        return self._geti("orc_get_faces", 3 * self.t)

    def get_spring_lengths(self):
        return self._getf("orc_get_spring_lengths", self.m)

    def get_spring_stiffness(self):
        return self._getf("orc_get_spring_stiffness", self.m)

    def get_params(self):
        return self._getf("orc_get_params", 32)

    def get_scene_bounds(self):
        lo, up = np.empty(3, np.float32), np.empty(3, np.float32)
        self.lib.orc_get_scene_bounds(self.h, _fp(lo), _fp(up))
        return lo, up

    def add_sphere(self, radius, pos, quat):
        pos, quat = _f(pos), _f(quat)
        assert self.lib.orc_add_sphere(self.h, C.c_float(radius), _fp(pos), _fp(quat)) == 0

    def clear_shapes(self):
        self.lib.orc_clear_shapes(self.h)

    def get_shape_states(self):
        return self._getf("orc_get_shape_states", 14 * self.get_n_shapes())

    def set_shape_states(self, s):
        s = _f(s)
        assert s.size >= 14 * self.get_n_shapes()
        self.lib.orc_set_shape_states(self.h, _fp(s))

    def max_neighbor_list(self):
        """Longest particle-contact candidate list any particle has had since set_scene (before truncation to 96)."""
        return self.lib.orc_max_neighbor_list(self.h)

    def get_last_neighbors(self):
        counts = np.empty(self.n, np.int32)
        lists = np.empty(self.n * 96, np.int32)
        self.lib.orc_get_last_neighbors(self.h, _ip(counts), _ip(lists))
        return counts, lists.reshape(self.n, 96)

    def get_last_shape_candidates(self):
        """collideShapes of the last substep: per particle, bit q = plane q, bit 8 + q = sphere q."""
        return self._geti("orc_get_last_shape_candidates", self.n).view(np.uint32)

    def missed_shape_contacts(self):
        """How often an iteration found a plane / sphere violated that collideShapes had not listed (since set_scene)."""
        return self.lib.orc_missed_shape_contacts(self.h)

    def accel_clamps(self):
        """Particle-substeps in which finalize's maxAcceleration clamp (NvFlex.h:112-113) changed a velocity (since set_scene)."""
        return int(self.lib.orc_accel_clamps(self.h))

    def degenerate_normals(self):
        """Contacts whose normal fell back to (0,1,0) because the pair was coincident (since set_scene)."""
        return int(self.lib.orc_degenerate_normals(self.h))
