"""evaluate.run_episodes: MaximumValuePolicy.act -> BatchedFlingEnv.step until every episode ends (run_sim.py:46-60),
with the statistics of utils.collect_stats computed from the loop."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_run_episodes_statistics(gpu_required):
    from flingbot_amd import nets, sim as fsim, tasks as ftasks
    from flingbot_amd.env import BatchedFlingEnv
    from flingbot_amd.evaluate import run_episodes

    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    n = 3
    gen = fsim.FlingSim(n_envs=n, solver=0)
    tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters(min_cloth_size=24, strict_min_edge_length=24, max_cloth_size=32) for _ in range(n)])
    gen.close()
    ctx = fsim.FlingSim(n_envs=n, solver=0)
    env = BatchedFlingEnv(ctx, image_dim=128, episode_length=2)
    policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                     obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                     depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                     value_expl_decay=1.0, device="cuda:0")
    stats = run_episodes(policy, env, tasks)
    assert all(net._hip is not None for net in policy.value_nets.values())  # the hand-written forward served the loop
    assert stats["coverage_steps"].shape[1] == n and stats["coverage_steps"].shape[0] == stats["delta_coverage_steps"].shape[0] + 1
    assert np.allclose(stats["coverage_steps"][0], stats["init_coverage"])
    assert np.allclose(stats["final_coverage"] - stats["init_coverage"], stats["delta_coverage_steps"].sum(axis=0), atol=1e-6)
    assert (stats["best_coverage"] >= stats["final_coverage"] - 1e-12).all()
    assert (stats["episode_length"] >= 1).all() and (stats["episode_length"] <= 2).all()
    assert sum(stats["action_primitive_counts"].values()) <= int(stats["episode_length"].sum())
    assert (stats["init_coverage"] > 0).all() and (stats["init_coverage"] < 1.05).all()
    assert stats["simulation_steps"] > 0 and all(env.terminate.values())
    # the final coverage is what the simulator reports now
    flat = np.array([t["flatten_area"] for t in tasks])
    assert np.allclose(np.array(ctx.coverage())[:n] / flat, stats["final_coverage"])
    ctx.close()
