"""The pybind11 `pyflex` drop-in module: surface (CPU) and behaviour through the reference's call pattern (GPU)."""
import os
import sys

import numpy as np
import pytest

from conftest import cloth_params

# the 40 m.def names of the reference module (PyFlex/bindings/pyflex.cpp:1137-1207)
REFERENCE_NAMES = """main init set_scene clean step render get_camera_params set_camera_params add_box add_sphere
add_capsule pop_box get_n_particles get_n_shapes get_n_rigids get_n_rigidPositions get_phases set_phases get_groups
set_groups get_positions set_positions get_edges get_faces get_restPositions get_rigidOffsets get_rigidIndices
get_rigidLocalPositions get_rigidGlobalPositions get_rigidRotations get_rigidTranslations get_velocities set_velocities
get_shape_states set_shape_states clear_shapes get_scene_upper get_scene_lower add_rigid_body set_shape_color""".split()


def _import_pyflex():
    import torch  # noqa: F401  -- load order: torch's ROCm runtime first, as in every other test of this suite
    from flingbot_amd import build

    path = build.build_pyflex()
    d = os.path.dirname(path)
    if d not in sys.path:
        sys.path.insert(0, d)
    import pyflex

    return pyflex


def test_module_exports_the_reference_surface():
    pyflex = _import_pyflex()
    assert len(REFERENCE_NAMES) == 40
    for name in REFERENCE_NAMES:
        assert callable(getattr(pyflex, name, None)), f"pyflex.{name} missing"
    for name in ("picker_reset", "movep", "step_n", "wait_until_stable", "_tenants"):  # additive device-side loops
        assert callable(getattr(pyflex, name, None)), f"pyflex.{name} missing"
    # init takes four REQUIRED positionals like the reference (m.def without py::arg, pyflex.cpp:1138)
    with pytest.raises(TypeError):
        pyflex.init()
    # calls before init raise instead of dereferencing a null solver
    with pytest.raises(RuntimeError):
        pyflex.get_positions()


_TENANT_CHILD = r"""
import ctypes, os, sys, time
lib = ctypes.CDLL(sys.argv[1])
lib.fs_tenants_register.argtypes = [ctypes.c_char_p]
lib.fs_tenants_count.argtypes = [ctypes.c_char_p, ctypes.c_int]
lib.fs_tenants_unregister.argtypes = [ctypes.c_char_p]
key = sys.argv[2].encode()
print(lib.fs_tenants_register(key), flush=True)       # live tenants including this one
line = sys.stdin.readline()                            # parent: "count" / "leave"
while line:
    if line.strip() == "count":
        print(lib.fs_tenants_count(key, 1), flush=True)
    elif line.strip() == "leave":
        lib.fs_tenants_unregister(key)
        print("left", flush=True)
        break
    line = sys.stdin.readline()
"""


def test_tenant_table_counts_live_processes_and_forgets_dead_ones(tmp_path, monkeypatch):
    """csrc/fs_tenants.cpp: the table through which the processes that share a GPU find each other (the reference's
    `--num_processes 16`, README.md:147-148).  Real processes: two children register next to the parent (1 -> 2 -> 3), one
    unregisters (2), the other is KILLED without a goodbye -- the next pruning count forgets it (1).  Different device keys do
    not see each other; registering twice does not count twice."""
    import ctypes
    import signal
    import subprocess

    from flingbot_amd import build

    lib_path = build.build_lib()
    monkeypatch.setenv("FLINGSIM_TENANT_DIR", str(tmp_path))
    lib = ctypes.CDLL(lib_path)
    lib.fs_tenants_register.argtypes = [ctypes.c_char_p]
    lib.fs_tenants_count.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.fs_tenants_unregister.argtypes = [ctypes.c_char_p]
    key, other = b"0000:05:00.0", b"0000:06:00.0"
    assert lib.fs_tenants_count(key, 1) == 0
    assert lib.fs_tenants_register(key) == 1 and lib.fs_tenants_register(key) == 1     # idempotent per process
    assert lib.fs_tenants_count(other, 1) == 0                                          # another device: another table
    env = dict(os.environ, FLINGSIM_TENANT_DIR=str(tmp_path))

    def child():
        p = subprocess.Popen([sys.executable, "-c", _TENANT_CHILD, lib_path, key.decode()], stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, text=True, env=env)
        return p, int(p.stdout.readline())

    a, seen_a = child()
    assert seen_a == 2 and lib.fs_tenants_count(key, 0) == 2
    b, seen_b = child()
    assert seen_b == 3 and lib.fs_tenants_count(key, 1) == 3
    a.stdin.write("count\n"); a.stdin.flush()
    assert int(a.stdout.readline()) == 3                                                # every tenant sees the same table
    a.stdin.write("leave\n"); a.stdin.flush()
    assert a.stdout.readline().strip() == "left" and a.wait(timeout=10) == 0
    assert lib.fs_tenants_count(key, 0) == 2
    b.send_signal(signal.SIGKILL)                                                       # no goodbye
    b.wait(timeout=10)
    assert lib.fs_tenants_count(key, -1) == 2                                           # occupied slots as they stand: the dead one too
    assert lib.fs_tenants_count(key, 1) == 1                                            # pruned: the pid is gone
    assert lib.fs_tenants_count(key, 0) == 1 and lib.fs_tenants_count(key, -1) == 1
    # self-heal: somebody wipes the table while this process is a tenant -- its next pruning count lists it again
    path = [os.path.join(tmp_path, f) for f in os.listdir(tmp_path) if f.endswith(key.decode().replace(":", "_"))][0]
    with open(path, "r+b") as fh:
        fh.seek(16)
        fh.write(bytes(16 * 62))
    assert lib.fs_tenants_count(key, -1) == 0 and lib.fs_tenants_count(key, 1) == 1 and lib.fs_tenants_count(key, -1) == 1
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith("flingsim-tenants-"))
    assert len(files) == 2 and all(os.stat(os.path.join(tmp_path, f)).st_mode & 0o077 == 0 for f in files)   # per user, private
    assert lib.fs_tenants_unregister(key) == 0 and lib.fs_tenants_count(key, 1) == 0


def test_tenant_table_drops_a_reused_pid(tmp_path, monkeypatch):
    """A slot whose pid is alive but belongs to ANOTHER process now (start time differs from the registrant's) is dead: written
    by hand into the table file -- this test process's pid with a wrong start time -- and gone after the next pruning count."""
    import ctypes
    import struct

    from flingbot_amd import build

    monkeypatch.setenv("FLINGSIM_TENANT_DIR", str(tmp_path))
    lib = ctypes.CDLL(build.build_lib())
    lib.fs_tenants_count.argtypes = [ctypes.c_char_p, ctypes.c_int]
    key = b"reused"
    assert lib.fs_tenants_count(key, 1) == 0             # creates and maps the table
    path = [os.path.join(tmp_path, f) for f in os.listdir(tmp_path) if f.endswith("-reused")][0]
    with open(path, "r+b") as fh:                        # header 16 B, then slots of (int pid, int, uint64 start)
        fh.seek(16)
        fh.write(struct.pack("<iiQ", os.getpid(), 0, 12345))
    assert lib.fs_tenants_count(key, 0) == 1             # alive by pid alone ...
    assert lib.fs_tenants_count(key, 1) == 0             # ... but not the process that registered


_COTENANT_CHILD = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np
import pyflex, scenarios as sc
pyflex.init(True, False, 720, 720)
e_f, e_i = np.zeros(0, np.float32), np.zeros(0, np.int32)
pyflex.set_scene(0, sc.survey_params(16), e_f, e_i, e_i, e_i, e_i, 0)
print("ready", pyflex._tenants()[0], pyflex._tenants()[1], flush=True)
sys.stdin.readline()
pyflex.step()
first = pyflex._tenants()[1]                  # a neighbour that registered meanwhile is noticed at the very next step
for _ in range(69):
    pyflex.step()
print("stepped", pyflex._tenants()[0], pyflex._tenants()[1], first, flush=True)
print(np.asarray(pyflex.get_positions()).view(np.uint32).sum(dtype=np.uint64), flush=True)
sys.stdin.readline()
"""


@pytest.mark.gpu
def test_module_selects_the_cotenant_backend_by_itself(gpu_required, tmp_path):
    """Two unmodified `pyflex` processes on one GPU, FLINGSIM_SHARED_GPU unset: the first starts as a lone tenant (AUTO), the
    second sees two and takes the co-tenant back-end at its first set_scene, the first follows at its next step -- and both
    produce the same bits (the back-ends are interchangeable).  FLINGSIM_SHARED_GPU=0 / 1 overrides the detection."""
    import subprocess

    from flingbot_amd import build

    mod_dir = os.path.dirname(build.build_pyflex())
    here = os.path.dirname(os.path.abspath(__file__))

    def child(**extra):
        env = {k: v for k, v in os.environ.items() if k != "FLINGSIM_SHARED_GPU"}
        env.update(FLINGSIM_TENANT_DIR=str(tmp_path), **extra)
        p = subprocess.Popen([sys.executable, "-c", _COTENANT_CHILD, mod_dir, here], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                             text=True, env=env)
        return p, p.stdout.readline().split()

    a, ra = child()
    assert ra == ["ready", "1", "auto"], ra
    b, rb = child()
    assert rb == ["ready", "2", "cotenant"], rb
    c, rc = child(FLINGSIM_SHARED_GPU="0")
    assert rc == ["ready", "3", "auto"], rc                         # the caller's word wins
    sums = []
    for p in (a, b, c):
        p.stdin.write("go\n"); p.stdin.flush()
    for p, want in ((a, "cotenant"), (b, "cotenant"), (c, "auto")):
        assert p.stdout.readline().split() == ["stepped", "3", want, want]   # a lone starter follows at its FIRST step
        sums.append(int(p.stdout.readline()))
    assert sums[0] == sums[1] == sums[2]
    for p in (a, b, c):
        p.stdin.write("bye\n"); p.stdin.flush()
        assert p.wait(timeout=30) == 0
    d, rd = child(FLINGSIM_SHARED_GPU="1")
    assert rd == ["ready", "1", "cotenant"], rd                     # (and the three above have left the table)
    d.stdin.write("go\n"); d.stdin.flush()
    assert d.stdout.readline().split() == ["stepped", "1", "cotenant", "cotenant"]
    assert int(d.stdout.readline()) == sums[0]
    d.stdin.write("bye\n"); d.stdin.flush()
    assert d.wait(timeout=30) == 0


@pytest.mark.gpu
def test_reference_call_pattern_matches_oracle(gpu_required):
    """flex_utils.set_scene / set_state / Picker.reset / Picker.step call order (flex_utils.py:74-119,304-355),
    float64 inputs and keyword arguments included, through the real module; positions equal the oracle's bit for bit."""
    from oracle import OracleSim

    pyflex = _import_pyflex()
    pyflex.init(True, True, 720, 720)
    orc = OracleSim()
    sp = cloth_params(24, 20, pos=(0.0, 2.0, 0.0))  # Task default: grid created below the ground (quirk 19)
    pyflex.set_scene(scene_idx=0, scene_params=sp, vertices=np.zeros(0), stretch_edges=np.zeros(0, int),
                     bend_edges=np.zeros(0, int), shear_edges=np.zeros(0, int), faces=np.zeros(0, int), thread_idx=0)
    orc.set_scene(sp)
    pyflex.step()
    orc.step()
    n = pyflex.get_n_particles()
    assert n == 480 and pyflex.get_n_shapes() == 0
    # set_state (float64 arrays, as the reference passes them)
    rng = np.random.RandomState(0)
    pos = pyflex.get_positions().reshape(-1, 4).astype(np.float64)
    pos[:, 1] = 0.05 + 0.01 * rng.rand(n)
    pos[:, [0, 2]] -= pos[:, [0, 2]].mean(0)
    vel = np.zeros(3 * n)
    pyflex.set_positions(pos.flatten())
    pyflex.set_velocities(vel)
    pyflex.set_shape_states(np.zeros(0))  # no shapes yet: a no-op like the reference
    pyflex.set_phases(pyflex.get_phases())
    pyflex.set_camera_params(np.array([0, 2, 0, np.pi / 2, -np.pi / 2, 0, 720, 720]))
    orc.set_positions(pos.flatten())
    orc.set_velocities(vel)
    # Picker.reset
    for c in ([0.04, 0.3, 0.0], [-0.04, 0.3, 0.0]):
        pyflex.add_sphere(0.02, c, [1, 0, 0, 0])
        orc.add_sphere(0.02, c, [1, 0, 0, 0])
    pyflex.set_shape_states(pyflex.get_shape_states())
    assert pyflex.get_n_shapes() == 2
    # a few Picker.step-like updates: prev := current, move, pin one particle
    for k in range(10):
        st = np.array(pyflex.get_shape_states()).reshape(-1, 14)
        st[:, 3:6] = st[:, :3]
        st[:, 1] -= 0.02
        p = np.array(pyflex.get_positions()).reshape(-1, 4)
        p[5, 3] = 0.0
        p[5, 1] += 0.002
        pyflex.set_shape_states(st)
        pyflex.set_positions(p)
        orc.set_shape_states(st)
        orc.set_positions(p)
        pyflex.step()
        orc.step()
    assert np.array_equal(pyflex.get_positions().view(np.uint32), orc.get_positions().view(np.uint32))
    assert np.array_equal(pyflex.get_velocities().view(np.uint32), orc.get_velocities().view(np.uint32))
    assert np.array_equal(pyflex.get_edges(), orc.get_edges())
    assert np.array_equal(pyflex.get_faces(), orc.get_faces())
    assert np.array_equal(pyflex.get_groups(), np.zeros(n, np.int32))
    cam = pyflex.get_camera_params()
    assert cam[0] == 720 and cam[1] == 720 and cam[3] == 2.0  # [w,h,px,py,pz,ax,ay,az] (pyflex.cpp:891)
    rgb, depth = pyflex.render()
    assert rgb.shape == (720 * 720 * 4,) and depth.shape == (720 * 720,)
    assert rgb.dtype == np.uint8 and depth.dtype == np.float32
    assert pyflex.get_rigidOffsets().size == 0 and pyflex.get_n_rigids() == 0
    with pytest.raises(RuntimeError):
        pyflex.add_box(np.ones(3), np.zeros(3), np.array([1, 0, 0, 0.0]), 0)
    # additive device-side loops: movep + wait_until_stable equal the numpy restatement driving the oracle
    from oracle.picker import OraclePicker

    tool = OraclePicker(orc)
    tool.particle_inv_mass = orc.get_positions().reshape(-1, 4)[:, 3].copy()
    tool.particle_inv_mass[5] = pos[5, 3]  # particle 5 was pinned by hand above; its saved mass is the original one
    p = np.array(pyflex.get_positions()).reshape(-1, 4)
    p[5, 3] = pos[5, 3]
    pyflex.set_positions(p)
    orc.set_positions(p.astype(np.float32).ravel())
    pyflex.picker_reset()
    cur = np.array(pyflex.get_shape_states()).reshape(-1, 14)[:, :3].astype(np.float64)
    targets = cur + [[0.01, -0.03, 0.02], [-0.02, -0.03, 0.0]]
    it = pyflex.movep(targets, [1, 0], speed=4e-3)
    assert it == tool.movep(targets, [True, False], speed=4e-3)
    # the same entry point under the name SURVEY 8(f) gives it: step_n(targets, speed, grasp, max_steps)
    back = cur + [[0.0, -0.03, 0.0], [0.0, -0.03, 0.0]]
    assert pyflex.step_n(back, 4e-3, [1, 0], 500) == tool.movep(back, [True, False], speed=4e-3, limit=500)
    stable, steps = pyflex.wait_until_stable(max_steps=25, tolerance=1e-2)
    done, ok = 0, False
    for _ in range(25):
        if np.abs(orc.get_velocities()).max() < 1e-2:
            ok = True
            break
        orc.step()
        done += 1
    assert (stable, steps) == (ok, done)
    assert np.array_equal(pyflex.get_positions().view(np.uint32), orc.get_positions().view(np.uint32))
    assert np.array_equal(pyflex.get_shape_states().view(np.uint32), orc.get_shape_states().view(np.uint32))
    pyflex.clean()
