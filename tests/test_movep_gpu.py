"""On-device picker + movep executor (fs_movep) against (a) the golden trajectory recorded from the
REFERENCE's Picker / PickerPickPlace classes and (b) the restated picker driving the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import cloth_params

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _picker_centres():
    r = np.sqrt(2 - 1) * 0.02 * 2.
    return [[0.0 + np.cos(2 * np.pi * i / 2) * r, 0.1, 0.0 + np.sin(2 * np.pi * i / 2) * r] for i in range(2)]


def _setup(env, g):
    env.set_scene(g["scene_params"])
    env.step(1)
    env.set_positions(g["init_pos"].ravel())
    env.set_velocities(np.zeros(3 * g["init_pos"].shape[0], np.float32))
    for c in _picker_centres():  # Picker.reset: add spheres, then shape states := centres (flex_utils.py:82-97)
        env.add_sphere(0.02, c, [1, 0, 0, 0])
    st = np.array(env.get_shape_states()).reshape(-1, 14)
    for i, c in enumerate(_picker_centres()):
        st[i] = np.hstack([c, c, [1, 0, 0, 0], [1, 0, 0, 0]])
    env.set_shape_states(st)


@pytest.mark.parametrize("solver", [1, 2])
def test_device_movep_matches_reference_picker_golden(gpu_required, solver):
    """92 simulation steps of approach / grasp+lift / stretch (min_steps) / fling / single release, executed entirely on
    the device, end in exactly the state the reference's Python classes produce on the oracle."""
    from flingbot_amd import sim as fsim

    g = np.load(os.path.join(GOLD, "picker_golden.npz"))
    ctx = fsim.FlingSim(n_envs=1, solver=solver)
    env = ctx.env(0)
    _setup(env, g)
    ctx.picker_reset(0)
    iters = []
    for target, speed, ms, gs in zip(g["targets"], g["speeds"], g["min_steps"], g["grasp"]):
        iters.append(ctx.movep(0, target, gs, speed=float(speed), min_steps=None if ms < 0 else int(ms)))
    assert iters == g["iters"].tolist()
    assert ctx.picked(0).tolist() == g["picked"][-1].tolist() == [0, -1]
    assert np.array_equal(env.get_shape_states().view(np.uint32), g["shapes"][-1].view(np.uint32))
    assert np.array_equal(env.get_positions().view(np.uint32), g["pos_last"].view(np.uint32))


def test_device_movep_batch_matches_oracle_picker(gpu_required):
    """Batched movep: episodes with different targets finish after different step counts; each equals its own CPU run
    (restated picker + oracle).  Also: limit -> MoveLimitError, like MoveJointsException."""
    from flingbot_amd import sim as fsim
    from oracle import OracleSim
    from oracle.picker import OraclePicker

    g = np.load(os.path.join(GOLD, "picker_golden.npz"))
    n_envs = 3
    ctx = fsim.FlingSim(n_envs=n_envs, solver=0)
    orcs, tools = [], []
    for e in range(n_envs):
        _setup(ctx.env(e), g)
        ctx.picker_reset(e)
        o = OracleSim()
        o.set_scene(g["scene_params"])
        o.step(1)
        o.set_positions(g["init_pos"].ravel())
        o.set_velocities(np.zeros(3 * g["init_pos"].shape[0], np.float32))
        t = OraclePicker(o)
        t.reset(_picker_centres())
        orcs.append(o)
        tools.append(t)
    p = g["init_pos"]
    p0, p1 = p[0, :3].astype(np.float64) + [0, 0.02, 0], p[23, :3].astype(np.float64) + [0, 0.02, 0]
    approach = np.array([[p0, p1]] * n_envs)
    lift = np.array([[p0 + [0, 0.05 + 0.03 * e, 0.01 * e], p1 + [0, 0.05 + 0.03 * e, -0.01 * e]] for e in range(n_envs)])
    it_a = ctx.movep(range(n_envs), approach, [[0, 0]] * n_envs, speed=0.1)
    it_l = ctx.movep(range(n_envs), lift, [[1, 1], [1, 0], [1, 1]], speed=5e-3)
    for e in range(n_envs):
        assert it_a[e] == tools[e].movep(approach[e], [False, False], speed=0.1)
        assert it_l[e] == tools[e].movep(lift[e], [[True, True], [True, False], [True, True]][e], speed=5e-3)
        assert ctx.picked(e).tolist() == [-1 if q is None else q for q in tools[e].picked_particles]
        assert np.array_equal(ctx.get_positions(e).view(np.uint32), orcs[e].get_positions().view(np.uint32)), e
        assert np.array_equal(ctx.get_shape_states(e).view(np.uint32), orcs[e].get_shape_states().view(np.uint32)), e
    assert len(set(it_l.tolist())) == 3  # the episodes really ran for different numbers of steps
    with pytest.raises(fsim.MoveLimitError):
        ctx.movep(0, lift[0] + [0, 0.3, 0], [1, 1], speed=5e-3, limit=10)
