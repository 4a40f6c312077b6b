"""Batched fling primitive + device-side feedback loops against the golden trajectories recorded
from the REFERENCE's SimEnv.pick_and_fling_primitive / stretch_cloth / lift_cloth / fling_primitive / movep and
flex_utils.wait_until_stable (tests/golden/make_golden.py fling), bit for bit."""
import numpy as np
import pytest

from fling_helpers import load_fling_golden, picker_centres

pytestmark = pytest.mark.gpu


def _make(g, n):
    from flingbot_amd import sim as fsim

    ctx = fsim.FlingSim(n_envs=n, solver=0)
    for e in range(n):
        env = ctx.env(e)
        env.set_scene(g["scene_params"])
        env.step(1)
        env.set_positions(g["init_pos"].ravel())
        env.set_velocities(np.zeros(3 * g["init_pos"].shape[0], np.float32))
        for c in picker_centres():
            env.add_sphere(0.02, c, [1, 0, 0, 0])
        st = np.array(env.get_shape_states()).reshape(-1, 14)
        for i, c in enumerate(picker_centres()):
            st[i] = np.hstack([c, c, [1, 0, 0, 0], [1, 0, 0, 0]])
        env.set_shape_states(st)
        ctx.picker_reset(e)
    return ctx


def test_batched_pick_and_fling_matches_reference_golden(gpu_required):
    from flingbot_amd.primitives import FlingPrimitives

    g = load_fling_golden()
    n = len(g["terminate"])
    ctx = _make(g, n)
    prim = FlingPrimitives(ctx, range(n))
    out = prim.pick_and_fling(g["p1"], g["p2"], g["g1"], g["g2"])
    for e in range(n):
        assert out[e]["terminated"] == bool(g["terminate"][e]), e
        if np.isnan(g["stretch_ret"][e]):
            assert out[e]["dist"] is None
        else:
            assert out[e]["dist"] == g["stretch_ret"][e] and out[e]["fling_height"] == g["lift_ret"][e], e
        assert np.array_equal(ctx.get_positions(e).view(np.uint32), g["pos_fling"][e].view(np.uint32)), e
    assert out[3]["skipped"] and prim.sim_steps > 0

    # wait_until_stable on the device: lifted + moving cloths that settle after different numbers of steps
    for e in range(n):
        lifted = ctx.get_positions(e).reshape(-1, 4).copy()
        lifted[:, 1] += np.float32(0.25)
        ctx.set_positions(e, lifted.ravel())
        vel = np.zeros((lifted.shape[0], 3), np.float32)
        vel[:, 1] = -0.5
        ctx.set_velocities(e, vel.ravel())
    stable, steps = ctx.wait_until_stable(range(n), max_steps=200, tolerance=2e-2)
    assert steps.tolist() == g["steps_drop"].tolist() and stable.tolist() == g["stable_drop"].tolist()
    assert len(set(steps.tolist())) >= n - 1  # the episodes really stopped at different steps
    for e in range(n):
        assert np.array_equal(ctx.get_positions(e).view(np.uint32), g["pos_final"][e].view(np.uint32)), e
        assert np.array_equal(ctx.get_shape_states(e).view(np.uint32), g["shapes_final"][e].view(np.uint32)), e


def test_wait_until_stable_budget_and_reductions_match_oracle(gpu_required):
    """max_steps runs out -> (False, max_steps); cloth_stats / stretch_probe equal the numpy expressions on the oracle."""
    from fling_helpers import OracleBatch

    g = load_fling_golden()
    ctx = _make(g, 2)
    orc = OracleBatch(2, g["scene_params"], g["init_pos"])
    for e in range(2):
        p = g["init_pos"].copy()
        p[:, 1] += np.float32(0.3 + 0.1 * e)
        p[:, 0] += np.float32(0.013 * e)
        v = np.zeros((p.shape[0], 3), np.float32)
        v[:, 1] = -0.3
        for s in (ctx, orc):
            (s.set_positions(e, p.ravel()), s.set_velocities(e, v.ravel())) if s is ctx else \
                (s.sims[e].set_positions(p.ravel()), s.sims[e].set_velocities(v.ravel()))
    stable, steps = ctx.wait_until_stable([0, 1], max_steps=7, tolerance=1e-2)
    o_stable, o_steps = orc.wait_until_stable([0, 1], max_steps=7, tolerance=1e-2)
    assert stable.tolist() == o_stable.tolist() == [False, False] and steps.tolist() == o_steps.tolist() == [7, 7]
    assert np.array_equal(ctx.cloth_stats([0, 1]).view(np.uint32), orc.cloth_stats([0, 1]).view(np.uint32))
    mids = np.array([[0.01, -0.02], [0.05, 0.03]], np.float32)
    thr = np.array([0.2, 0.35], np.float32)
    s_gpu, n_gpu = ctx.stretch_probe([0, 1], mids, thr)
    s_cpu, n_cpu = orc.stretch_probe([0, 1], mids, thr)
    assert s_gpu.tolist() == s_cpu.tolist()
    assert np.array_equal(n_gpu.view(np.uint32), n_cpu.view(np.uint32))
    for e in range(2):
        assert np.array_equal(ctx.get_positions(e).view(np.uint32), orc.get_positions(e).view(np.uint32))
    # SimEnv.preaction / postaction's displacement test (simEnv.py:464-475), float32 like the numpy expression
    ctx.snapshot_positions([0, 1])
    pre = [ctx.get_positions(e).reshape(-1, 4)[:, :3].copy() for e in range(2)]
    ctx.step(9)
    got = ctx.max_displacement([0, 1])
    for e in range(2):
        post = ctx.get_positions(e).reshape(-1, 4)[:, :3]
        want = np.linalg.norm(np.abs(post - pre[e]), axis=1).max()
        assert want.dtype == np.float32 and got[e] == want and want > 0, (e, got[e], want)


def test_batched_drag_place_stretchdrag_match_reference_golden(gpu_required):
    """The other manipulation primitives of SimEnv (simEnv.py:320-428) through the device-side movep, against
    tests/golden/primitives_golden.npz."""
    from fling_helpers import load_primitives_golden, run_primitives_golden

    g = load_primitives_golden()
    run_primitives_golden(lambda n: _make(g, n), lambda sim, k: sim.get_positions(k), lambda sim, k: sim.get_shape_states(k))


def test_reductions_propagate_nan_like_numpy(gpu_required):
    """np.abs(v).max() with a NaN velocity is NaN and never below the tolerance: wait_until_stable keeps stepping until the
    budget runs out; cloth_stats reports NaN like numpy's min / max."""
    g = load_fling_golden()
    ctx = _make(g, 1)
    v = np.zeros((g["init_pos"].shape[0], 3), np.float32)
    v[7, 1] = np.nan
    ctx.set_velocities(0, v.ravel())
    assert np.isnan(ctx.cloth_stats([0])[0, 2])
    stable, steps = ctx.wait_until_stable(0, max_steps=3, tolerance=1e-2)
    assert (stable, steps) == (False, 3)
