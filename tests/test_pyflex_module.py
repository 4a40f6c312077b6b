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
    for name in ("picker_reset", "movep", "step_n", "wait_until_stable"):  # additive device-side loops
        assert callable(getattr(pyflex, name, None)), f"pyflex.{name} missing"
    # init takes four REQUIRED positionals like the reference (m.def without py::arg, pyflex.cpp:1138)
    with pytest.raises(TypeError):
        pyflex.init()
    # calls before init raise instead of dereferencing a null solver
    with pytest.raises(RuntimeError):
        pyflex.get_positions()


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
