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


@pytest.mark.parametrize("loop", ["run_tasks", "run_episodes"])
def test_full_size_eval_loop_config5(gpu_required, loop):
    """BASELINE.json configs[4] at the reference's own sizes (README.md:194, environment/simEnv.py:56-71 defaults): 12 rotations
    x 8 scales, 720 x 720 render -> 400 x 400 observation with adaptive scaling, cloth sides 64..104 ('hard' tasks of the
    reference's generator, environment/tasks.py:105-275), 8 episodes x up to 3 actions, seeded random-init fling policy (no
    flingbot.pth in this image), through the asynchronous loop (evaluate.run_tasks: what bench.py's eval_loop entry times) and
    the lock-step one.  Checks the loop's invariants, that every stage ran on its device path (hand-written value
    net, fs_observe_batch, fs_prepare_image, fs_select_action, streaming / fused solver forms) and reports the rates."""
    import time
    from flingbot_amd import nets, sim as fsim, tasks as ftasks
    from flingbot_amd.env import BatchedFlingEnv
    from flingbot_amd import evaluate

    random.seed(5); np.random.seed(5); torch.manual_seed(5)
    n, actions = 8, 3
    params = [ftasks.draw_task_parameters() for _ in range(n)]          # the reference's defaults: sides 64 .. 104
    sides = np.array([p["cloth_size"] for p in params])
    assert sides.min() >= 64 and sides.max() <= 104 and (sides.prod(axis=1) > 4096).any()
    gen = fsim.FlingSim(n_envs=n, solver=0)
    tasks = ftasks.generate_tasks(gen, params)
    gen.close()
    assert all(t is not None for t in tasks)
    ctx = fsim.FlingSim(n_envs=n, solver=0)
    env = BatchedFlingEnv(ctx, episode_length=actions)                  # image_dim 400, render 720, 12 x 8 transforms
    assert env.image_dim == 400 and env.render_dim == 720 and len(env.transformations) == 96 and env.obs_dim == 64
    policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                     obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                     depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                     value_expl_decay=1.0, device="cuda:0")
    t0 = time.perf_counter()
    stats = getattr(evaluate, loop)(policy, env, tasks)
    dt = time.perf_counter() - t0
    # stages on their device paths
    assert all(net._hip is not None for net in policy.value_nets.values())
    assert ctx.last_kernel_form() == fsim.FS_FORM_STREAM_GRIDL            # grid cloths above 4096 particles stream
    assert all(np.asarray(d).shape == (400, 400) for d in env.pretransform_depth.values())
    assert all(len(f) == 8 for f in env.adaptive_scale_factors.values())
    # the loop's invariants
    T = stats["coverage_steps"].shape[0]
    assert 2 <= T <= actions + 1 and stats["coverage_steps"].shape[1] == n
    assert np.allclose(stats["coverage_steps"][0], stats["init_coverage"])
    assert np.allclose(stats["final_coverage"] - stats["init_coverage"], stats["delta_coverage_steps"].sum(axis=0), atol=1e-6)
    assert (stats["episode_length"] >= 1).all() and (stats["episode_length"] <= actions).all()
    assert (stats["init_coverage"] > 0.05).all() and (stats["init_coverage"] < 1.0).all()   # crumpled 'hard' tasks
    assert np.isfinite(stats["final_coverage"]).all() and (stats["final_coverage"] < 1.2).all()
    assert all(env.terminate.values())
    flings = sum(stats["action_primitive_counts"].values())
    assert flings >= 1 and stats["simulation_steps"] > 300 * flings
    flat = np.array([t["flatten_area"] for t in tasks])
    assert np.allclose(np.array(ctx.coverage())[:n] / flat, stats["final_coverage"])
    print(f"\n  config 5, {loop} (8 episodes, sides {sides.min()}..{sides.max()}, 12 x 8 transforms, 720 -> 400): {dt:.2f} s, "
          f"{flings} flings ({flings / dt:.1f} /s), {stats['simulation_steps']} episode-steps "
          f"({stats['simulation_steps'] / dt:.0f} /s), coverage {stats['mean']['init_coverage']:.3f} -> "
          f"{stats['mean']['final_coverage']:.3f}")
    ctx.close()


def _policy(env, seed=1):
    from flingbot_amd import nets

    torch.manual_seed(seed)
    return nets.MaximumValuePolicy(action_primitives=list(env.actions), num_rotations=12, scale_factors=list(env.scale_factors),
                                   obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                   depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                   value_expl_decay=1.0, device="cuda:0")


@pytest.mark.parametrize("actions", [("fling",), ("fling", "stretchdrag", "drag", "place")])
def test_async_task_loop_equals_lockstep_loop(gpu_required, actions):
    """evaluate.run_tasks -- every slot runs reset / act / step on its own and pulls the next task when its episode ends
    (run_sim.py:46-60 with utils.step_env's ray.wait asynchrony, utils.py:394-418) -- against evaluate.run_episodes, the
    lock-step loop, on the same six generated tasks: once with one slot per task and once with TWO slots for the six tasks
    (continuous batching: tasks land in slots the lock-step run never used for them), each through the pipelined scheduler
    (the default: chunks of simulation queued ahead, host-side services on the service lane while they run, scenes prebuilt
    on a worker thread) and through the blocking one; with the fling-only policy of the
    reference's released model and with all four primitives in the action space (each episode then runs whichever program
    its own arg-max picked, side by side).  Coverage after every step, episode
    lengths, action counts and the simulation-step total are identical."""
    from flingbot_amd import sim as fsim, tasks as ftasks
    from flingbot_amd.env import BatchedFlingEnv
    from flingbot_amd.evaluate import run_episodes, run_tasks

    random.seed(3); np.random.seed(3)
    n = 6
    gen = fsim.FlingSim(n_envs=n, solver=0)
    tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters(min_cloth_size=26, strict_min_edge_length=26, max_cloth_size=40)
                                        for _ in range(n)])
    gen.close()
    results = []
    for mode, slots in (("lockstep", n), ("async", n), ("async", 2), ("async-blocking", n), ("async-blocking", 2), ("async-deep", 2)):
        ctx = fsim.FlingSim(n_envs=slots, solver=0)
        env = BatchedFlingEnv(ctx, action_primitives=actions, image_dim=128, episode_length=3)
        policy = _policy(env)
        if mode == "lockstep":
            stats = run_episodes(policy, env, tasks)
        elif mode == "async":            # the default: chunks queued ahead, services on the service lane, scenes prebuilt
            stats = run_tasks(policy, env, tasks)
        elif mode == "async-deep":       # one-sequence chunks, three of them open
            stats = run_tasks(policy, env, tasks, cap_min=1, cap=1)
        else:                            # the blocking scheduler, scenes built in place
            stats = run_tasks(policy, env, tasks, pipeline=False, prebuild=False)
        assert ctx.advance_in_flight() == 0
        assert all(net._hip is not None for net in policy.value_nets.values())
        results.append(stats)
        ctx.close()
    ref = results[0]
    assert ref["coverage_steps"].shape[0] >= 2 and sum(ref["action_primitive_counts"].values()) > 0
    # the SHARED task queue of a multi-rank run (distributed.SharedTaskCounter; here two "ranks" one after the other on one counter):
    # the first stops claiming after three tasks, the second takes what is left -- every task exactly once, and every task's
    # coverage trace is the lock-step run's, whichever call ran it
    from flingbot_amd import distributed as fdist
    counter, taken = fdist.SharedTaskCounter(n), []
    limited = lambda k: counter.claim(min(k, max(0, 3 - len(counter.claimed))))   # noqa: E731
    for claim in (limited, counter.claim):
        ctx = fsim.FlingSim(n_envs=2, solver=0)
        env = BatchedFlingEnv(ctx, action_primitives=actions, image_dim=128, episode_length=3)
        part = run_tasks(_policy(env), env, tasks, claim=claim, claim_first=2)
        ctx.close()
        taken.append(part["task_indices"].tolist())
        for col, ti in enumerate(part["task_indices"]):
            steps = int(part["episode_length"][col])
            assert steps == int(ref["episode_length"][ti])
            assert np.array_equal(part["coverage_steps"][:steps + 1, col], ref["coverage_steps"][:steps + 1, ti]), ti
    assert len(taken[0]) == 3 and sorted(taken[0] + taken[1]) == list(range(n))
    for other in results[1:]:
        assert np.array_equal(ref["coverage_steps"], other["coverage_steps"])
        assert np.array_equal(ref["episode_length"], other["episode_length"])
        assert ref["action_primitive_counts"] == other["action_primitive_counts"]
        assert ref["simulation_steps"] == other["simulation_steps"]
        assert other["scheduler"]["calls"] > 0
