"""Batched task generation on the device (flingbot_amd/tasks.py, SURVEY.md 8f row f4 generator half) against the tasks the
REFERENCE's generate_randomization produced on the oracle (tests/golden/task_golden.npz), bit for bit."""
import pytest

pytestmark = pytest.mark.gpu


def test_device_task_generator_matches_reference_golden(gpu_required):
    from fling_helpers import check_tasks_against_golden
    from flingbot_amd import sim as fsim

    check_tasks_against_golden(lambda n: fsim.FlingSim(n_envs=n, solver=0))


def test_generated_tasks_load_and_fling(gpu_required):
    """generate -> load into a fresh context (set_scene + set_state, flex_utils.py:320-355) -> SimEnv.reset's picker set-up
    -> one batched fling: the loaded state is the generated one, bit for bit, and the primitives run on it."""
    import random

    import numpy as np

    from flingbot_amd import sim as fsim, tasks as ftasks
    from flingbot_amd.primitives import FlingPrimitives

    random.seed(5)
    np.random.seed(5)
    params = [ftasks.draw_task_parameters(min_cloth_size=24, strict_min_edge_length=24, max_cloth_size=40) for _ in range(3)]
    gen = fsim.FlingSim(n_envs=3, solver=0)
    tasks = ftasks.generate_hard_tasks(gen, params)
    assert all(t is not None for t in tasks)
    ctx = fsim.FlingSim(n_envs=3, solver=0)
    assert ftasks.load_tasks(ctx, tasks) == [0, 1, 2]
    for e, t in enumerate(tasks):
        assert ctx.n_particles(e) == int(np.prod(t["cloth_size"]))
        assert np.array_equal(ctx.get_positions(e).view(np.uint32), np.asarray(t["particle_pos"], np.float32).view(np.uint32))
        assert abs(ctx.coverage()[e] - t["initial_coverage"]) <= 1e-12
    prim = FlingPrimitives(ctx, range(3))
    prim.setup_pickers()
    prim.preaction()
    p = [ctx.get_positions(e).reshape(-1, 4) for e in range(3)]
    p1 = np.array([q[np.argmin(q[:, 0]), :3] for q in p], np.float64)
    p2 = np.array([q[np.argmax(q[:, 0]), :3] for q in p], np.float64)
    out = prim.pick_and_fling(p1, p2, [True] * 3, [True] * 3)
    term = prim.postaction()
    assert len(out) == 3 and len(term) == 3 and all(np.isfinite(ctx.coverage()))
