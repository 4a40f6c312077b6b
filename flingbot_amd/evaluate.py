"""The evaluation loop of run_sim.py (run_sim.py:46-60: `policy.act(observations)` -> `step_env`) for a batch of episodes
on one GPU, and the per-episode statistics the reference derives from its replay buffer (utils.py:186-391 collect_stats:
init / final / best coverage, delta coverage per step, episode length, primitive counts) -- computed here from the loop
itself, since the HDF5 replay storage is out of scope (h5py is absent from this image).

    sim = FlingSim(n_envs=E); env = BatchedFlingEnv(sim, ...); policy = nets.MaximumValuePolicy(...)
    stats = run_episodes(policy, env, tasks)

One `policy.act` call pushes the transformed observations of ALL running episodes through each value net in one batched
forward (the hand-written forward after `fold_batchnorm()`); one `env.step` call runs action selection and the chosen
primitives for all of them.  Coverages are normalised by the task's flattened area like the reference's
(simEnv.py:495-497, 675-677)."""
import numpy as np
import torch


def run_episodes(policy, env, tasks, max_steps=None, fold=True):
    """Reset `env` on `tasks` and act until every episode terminates.  Returns a dict of per-episode arrays plus the
    aggregate scalars collect_stats prints (means over episodes)."""
    if fold:
        for net in policy.value_nets.values():
            if getattr(net, "_folded", None) is None:
                net.fold_batchnorm()
    obs = env.reset(tasks)
    envs = list(env.envs)
    flat = np.array([float(tasks[e]["flatten_area"]) for e in envs])
    cover = lambda: np.array(env.sim.coverage())[envs] / flat
    init = cover()
    trace = [init]                       # coverage after every env.step, all episodes
    lengths = np.zeros(len(envs), int)
    counts = {a: 0 for a in env.actions}
    steps = 0
    while obs and (max_steps is None or steps < max_steps):
        order = sorted(obs)
        with torch.no_grad():
            maps = policy.act([obs[e] for e in order], keep_on_device=True)  # list of {primitive: [T, D, D]}
        value_maps = {e: {k: v.to(env.device) for k, v in m.items()} for e, m in zip(order, maps)}
        obs, rewards, terminate, actions = env.step(value_maps)
        for e, a in actions.items():
            lengths[envs.index(e)] += 1
            if a is not None:
                counts[a] += 1
        trace.append(cover())
        steps += 1
    trace = np.stack(trace)              # [steps + 1, episodes]
    final = trace[-1]
    deltas = np.diff(trace, axis=0)
    return {
        "init_coverage": init, "final_coverage": final, "best_coverage": trace.max(axis=0),
        "episode_delta_coverage": final - init, "episode_length": lengths,
        "delta_coverage_steps": deltas, "coverage_steps": trace, "action_primitive_counts": counts,
        "simulation_steps": int(env.prim.sim_steps),
        "mean": {"init_coverage": float(init.mean()), "final_coverage": float(final.mean()),
                 "best_coverage": float(trace.max(axis=0).mean()), "episode_delta_coverage": float((final - init).mean()),
                 "episode_length": float(lengths.mean())},
    }


def run_episodes_sharded(policy, env, tasks, episodes_per_rank, runner=run_episodes, **kwargs):
    """BASELINE.json configs[3] for the evaluation loop: global episode g runs on rank g // episodes_per_rank (one process
    per GPU, launched with torch.distributed.run); the only exchange is the all_gather of the per-episode initial / final
    coverages at the end (RCCL over xGMI, 4 bytes per episode).  Returns this rank's statistics plus `all_init_coverage`
    / `all_final_coverage` ordered by global episode id."""
    from . import distributed as fdist

    rank, _, world = fdist.init_from_env()
    mine = [tasks[g] for g in fdist.episode_range(rank, episodes_per_rank)]
    stats = runner(policy, env, mine, **kwargs)
    device = getattr(env, "device", None) if torch.cuda.is_available() else None
    stats["all_init_coverage"] = fdist.gather_rewards(stats["init_coverage"], device=device).cpu().numpy()
    stats["all_final_coverage"] = fdist.gather_rewards(stats["final_coverage"], device=device).cpu().numpy()
    stats["rank"], stats["world"] = rank, world
    return stats
