"""BatchedFlingEnv (flingbot_amd/env.py): SimEnv.reset / step composed from the device-side stages for several episodes.
The stages are pinned individually elsewhere; this checks the composition's contract (shapes, rewards, termination)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batched_env_reset_and_step(gpu_required):
    from flingbot_amd import nets, sim as fsim, tasks as ftasks
    from flingbot_amd.env import BatchedFlingEnv

    random.seed(2)
    np.random.seed(2)
    n = 3
    params = [ftasks.draw_task_parameters(min_cloth_size=40, strict_min_edge_length=40, max_cloth_size=56) for _ in range(n)]
    gen = fsim.FlingSim(n_envs=n, solver=0)
    tasks = ftasks.generate_tasks(gen, params)
    ctx = fsim.FlingSim(n_envs=n, solver=0)
    env = BatchedFlingEnv(ctx, image_dim=96, episode_length=2)
    obs = env.reset(tasks)
    assert sorted(obs) == list(range(n))
    T = len(env.transformations)
    assert all(o.is_cuda and tuple(o.shape) == (T, 4, 64, 64) for o in obs.values())
    torch.manual_seed(0)
    net = nets.SpatialValueNet(rgb_only=True, device=env.device).to(env.device).eval()
    steps_taken = 0
    for step in range(2):
        if not obs:
            break  # every episode ended early ("didn't really move cloth", simEnv.py:473-475)
        steps_taken += 1
        with torch.no_grad():
            vmaps = {e: {"fling": net(o).squeeze(1)} for e, o in obs.items()}
        obs, rewards, terminate, actions = env.step(vmaps)
        assert all(np.isfinite(r) for r in rewards.values())
        assert set(actions.values()) <= {"fling", None}
    assert steps_taken >= 1 and all(terminate.values())  # episode_length = 2 (or an early end)
    assert obs == {}                                       # nothing left to observe
    assert env.prim.sim_steps > 0


def test_step_bookkeeping_matches_reference_golden(gpu_required):
    """SimEnv.step (environment/simEnv.py:464-515) recorded from the reference's own method with the action selection
    scripted (tests/golden/make_golden.py step): BatchedFlingEnv.step_actions on the device reproduces rewards,
    termination (early end when the cloth did not move, episode_length, a fling aborted because the grasp missed), timesteps,
    simulation-step counts, grasp flags, every particle position and the picker states bit for bit."""
    from fling_helpers import load_step_golden, run_step_golden
    from flingbot_amd import sim as fsim

    g = load_step_golden()

    def make(n):
        ctx = fsim.FlingSim(n_envs=n, solver=0)
        for e in range(n):
            env = ctx.env(e)
            env.set_scene(g["scene_params"])
            env.step(1)
            env.set_positions(g["init_pos"].ravel())
            env.set_velocities(np.zeros(3 * g["init_pos"].shape[0], np.float32))
        return ctx

    run_step_golden(make, lambda sim, k: sim.get_positions(k), lambda sim, k: sim.get_shape_states(k))
