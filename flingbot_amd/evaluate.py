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


def run_tasks(policy, env, tasks, fold=True, cap_min=None, cap=None, max_steps=None, pipeline=True, prebuild=True, claim=None, claim_first=None):
    """The evaluation loop the way the reference actually runs it: every environment steps on its own, the policy acts for
    whichever environments are ready (utils.step_env, utils.py:394-418: `ray.wait` on the step futures), and an environment
    whose episode ends pulls the NEXT task by itself (SimEnv.step -> on_episode_end -> reset -> get_task_fn, tasks.py
    TaskLoader.get_next_task) -- continuous batching of `len(tasks)` episodes over the `env.sim.n_envs` slots of one GPU
    context.  Each slot runs BatchedFlingEnv.episode_program; schedule.run_programs serves them together: observations of
    all ready slots in one fs_observe_batch call, one batched value-net forward, shared launch sequences for every
    simulation step (fs_advance).  Episodes are independent and every stage is deterministic per episode, so the
    statistics equal run_episodes' on the same tasks exactly (tests/test_evaluate_gpu.py); what changes is that no episode
    waits for another one -- neither inside an action nor between actions nor at the end of an episode.
    PRECONDITION of "equal to run_episodes exactly": the value nets' forward must not depend on which episodes share a
    batch -- the "act" service batches whichever slots are ready.  The hand-written forward (fs_value_net_forward, `_hip`
    set: obs_dim 64) computes every image on its own and is batch-invariant; the PyTorch / MIOpen fallback may pick another
    algorithm for another batch size and flip an arg-max, so with a net on the fallback every episode gets a forward of its
    own here (slower, same guarantee) and a warning says so.
    max_steps: at most that many actions per episode (run_episodes' argument of the same name; None = env.episode_length).
    pipeline: chunks of simulation are queued ahead (schedule.run_programs_pipelined: fs_advance_begin / fs_advance_end, the
    services on the context's service lane) so that the device does not wait while the host serves requests; False: the
    blocking scheduler.  cap_min / cap: bounds of a chunk (defaults 2 / 4 pipelined, 4 / 32 blocking).
    prebuild: the host half of every task's set_scene is built ahead of its turn on a worker thread (tasks.ScenePrebuilder).
    claim: None = this call runs every task of `tasks`.  A callable claim(k) -> up to k indices into `tasks` nobody else has
    taken ([] when the set is used up) makes the queue a SHARED one -- distributed.SharedTaskCounter: several ranks pull from one
    task set the way the reference's environments pull from one TaskLoader actor, a rank with long episodes simply takes fewer --
    and this call runs the tasks it gets; the statistics then cover those (`task_indices`, ascending).  claim_first: how many tasks
    to take up front (default: one per slot; a sharded run passes its fair share, so that the first rank to arrive does not empty
    the set); afterwards a sixteenth of the slots at a time.
    Returns run_episodes' dictionary (arrays ordered by task index) plus `scheduler` (launch statistics of the run);
    `simulation_steps` excludes the step inside every set_scene, as the lock-step path's count does."""
    from collections import deque

    from . import nets, schedule as sch

    if fold:
        for net in policy.value_nets.values():
            if getattr(net, "_folded", None) is None:
                net.fold_batchnorm()
    sim = env.sim
    slots = list(range(min(sim.n_envs, len(tasks))))
    env.open_slots(slots)
    queue = deque(range(len(tasks))) if claim is None else deque()
    records = {}

    batch_invariant = all(getattr(net, "_hip", None) is not None for net in policy.value_nets.values())
    if not batch_invariant:
        import warnings
        warnings.warn("run_tasks: a value net runs on the PyTorch fallback (no fs_value_net_forward for its observation "
                      "size): one forward per episode, so that results do not depend on which slots are ready together")

    from .tasks import ScenePrebuilder
    scenes = ScenePrebuilder(tasks, ahead=max(8, len(slots) // 4), order=None if claim is None else []) if prebuild else None

    def refill(k):
        got = [int(i) for i in claim(k)]
        queue.extend(got)
        if scenes is not None:
            scenes.expect(got)

    if claim is not None:      # one task per slot to start with; afterwards in small chunks, so that the last tasks of the set
        refill(len(slots) if claim_first is None else max(0, min(len(slots), int(claim_first))))   # do not pile up on one rank

    def slot_program(slot):
        while True:
            if not queue and claim is not None:
                refill(max(1, len(slots) // 16))
            if not queue:
                return
            ti = queue.popleft()
            records[ti] = yield from env.episode_program(slot, tasks[ti], max_actions=max_steps,
                                                         prebuilt=scenes.get(ti) if scenes is not None else None)

    def observe(reqs):
        es = [e for e, _ in reqs]
        obs = env.get_obs_batch(es)
        return [nets.prepare_image(obs[k], env.get_transformations(e), env.obs_dim) for k, e in enumerate(es)]

    def act(reqs):
        with torch.no_grad():
            if batch_invariant:
                maps = policy.act([a[0] for _, a in reqs], keep_on_device=True)
            else:
                maps = [policy.act([a[0]], keep_on_device=True)[0] for _, a in reqs]
        return [{k: v.to(env.device) for k, v in m.items()} for m in maps]

    def coverage(reqs):
        cov = np.array(sim.coverage())
        return [cov[e] for e, _ in reqs]

    def snapshot(reqs):
        sim.snapshot_positions([e for e, _ in reqs])
        return [None] * len(reqs)

    services = {"observe": observe, "act": act, "coverage": coverage, "snapshot": snapshot,
                "max_disp": lambda reqs: list(sim.max_displacement([e for e, _ in reqs]))}
    if cap_min is None:
        cap_min = 2 if pipeline else 4
    if cap is None:
        cap = 4 if pipeline else 32
    try:
        # (run_ahead: the programs' code between two requests is host-only -- schedule.py's primitives and episode_program)
        sch.run_programs(env.prim, {s: slot_program(s) for s in slots}, cap_min=cap_min, cap=cap, services=services,
                         pipeline=pipeline, run_ahead=True)
    finally:
        if scenes is not None:
            scenes.close()
    done = sorted(records)                   # every task of `tasks`, or the ones this call claimed
    n = len(done)
    flat = np.array([float(tasks[i]["flatten_area"]) for i in done])
    lengths = np.array([len(records[i]["actions"]) for i in done], int)
    steps = int(lengths.max()) if n else 0
    trace = np.stack([[records[i]["coverage"][min(k, lengths[j])] for j, i in enumerate(done)] for k in range(steps + 1)]) / flat \
        if n else np.zeros((1, 0))
    counts = {a: 0 for a in env.actions}
    for i in done:
        for a in records[i]["actions"]:
            if a is not None:
                counts[a] += 1
    init, final = trace[0], trace[-1]
    nanmean = lambda v: float(np.mean(v)) if len(v) else float("nan")   # noqa: E731  (a rank that claimed nothing)
    return {
        "task_indices": np.array(done, int),
        "init_coverage": init, "final_coverage": final, "best_coverage": trace.max(axis=0),
        "episode_delta_coverage": final - init, "episode_length": lengths,
        "delta_coverage_steps": np.diff(trace, axis=0), "coverage_steps": trace, "action_primitive_counts": counts,
        "simulation_steps": int(env.prim.sim_steps - env.unpaid_steps), "scheduler": dict(getattr(env.prim, "sched_stats", {})),
        "mean": {"init_coverage": nanmean(init), "final_coverage": nanmean(final),
                 "best_coverage": nanmean(trace.max(axis=0)), "episode_delta_coverage": nanmean(final - init),
                 "episode_length": nanmean(lengths)},
        "records": [records[i] for i in done],   # per episode: what SimEnv logs per step (taskio.save_replay stores it)
    }


def run_episodes_sharded(policy, env, tasks, episodes_per_rank, runner=None, **kwargs):
    """BASELINE.json configs[3] for the evaluation loop: global episode g runs on rank g // episodes_per_rank (one process
    per GPU, launched with torch.distributed.run); the only exchange is the all_gather of the per-episode initial / final
    coverages at the end (RCCL over xGMI, 4 bytes per episode).  Returns this rank's statistics plus `all_init_coverage`
    / `all_final_coverage` ordered by global episode id.  kwargs go to the runner: both run_tasks (the default; returns the
    extra key `scheduler`) and run_episodes take `max_steps`."""
    from . import distributed as fdist

    runner = run_tasks if runner is None else runner  # (run_tasks streams the rank's tasks through its GPU context's slots)
    rank, _, world = fdist.init_from_env()
    mine = [tasks[g] for g in fdist.episode_range(rank, episodes_per_rank)]
    stats = runner(policy, env, mine, **kwargs)
    device = getattr(env, "device", None) if torch.cuda.is_available() else None
    stats["all_init_coverage"] = fdist.gather_rewards(stats["init_coverage"], device=device).cpu().numpy()
    stats["all_final_coverage"] = fdist.gather_rewards(stats["final_coverage"], device=device).cpu().numpy()
    stats["rank"], stats["world"] = rank, world
    return stats


def merge_rank_statistics(stats, per_rank, device=None):
    """The per-episode coverages of every rank's run_tasks statistics, gathered in task order (ranks hold contiguous blocks of
    `per_rank` tasks; the last block may be shorter) and reduced to the summary the command line prints."""
    from . import distributed as fdist

    # every rank says how many episodes it ran (its block may be short, or empty); the padding behind them is never looked at, so
    # a coverage that really IS NaN (a diverged episode, a degenerate flatten_area) stays in the statistics and shows
    pad = lambda v: np.concatenate([np.asarray(v, np.float32), np.zeros(per_rank - len(v), np.float32)])  # noqa: E731
    n_mine = len(stats["init_coverage"])
    assert len(stats["final_coverage"]) == n_mine <= per_rank, (n_mine, len(stats["final_coverage"]), per_rank)
    init = fdist.gather_rewards(pad(stats["init_coverage"]), device=device).cpu().numpy()
    final = fdist.gather_rewards(pad(stats["final_coverage"]), device=device).cpu().numpy()
    tail = fdist.gather_rewards([float(stats["simulation_steps"]), float(n_mine)], device=device).cpu().numpy().reshape(-1, 2)
    counts = tail[:, 1].astype(np.int64)
    keep = (np.arange(per_rank)[None, :] < counts[:, None]).ravel()
    init, final = init[keep], final[keep]
    _, _, world = fdist.init_from_env()
    out = {"gpus": world, "episodes": int(counts.sum()), "init_coverage": float(init.mean()) if init.size else float("nan"),
           "final_coverage": float(final.mean()) if final.size else float("nan"),
           "episode_delta_coverage": float((final - init).mean()) if init.size else float("nan"),
           "simulation_steps": int(tail[:, 0].sum()),
           "note": "coverages and steps over all ranks; best_coverage / episode_length / action counts are rank 0's"}
    bad = int((~np.isfinite(init)).sum() + (~np.isfinite(final)).sum())
    if bad:
        out["non_finite_coverages"] = bad     # surfaced, not dropped: the means above are NaN because an episode's coverage is
    return out


def merge_shared_statistics(stats, n_tasks, device=None):
    """The sharded run with ONE task queue for all ranks (run_tasks(claim=SharedTaskCounter.claim)): every rank writes the
    coverages of the tasks it ran at their task indices into vectors of the set's length, zeros elsewhere, and one SUM
    all-reduce gives every rank the whole set; a third vector counts who ran what -- every task exactly once, or the run is
    refused -- and tells how many tasks each rank ended up with."""
    from . import distributed as fdist

    rank, _, world = fdist.init_from_env()
    idx = np.asarray(stats["task_indices"], int)
    buf = np.zeros((3 + world, n_tasks))
    buf[0, idx], buf[1, idx], buf[2, idx] = stats["init_coverage"], stats["final_coverage"], 1.0
    buf[3 + rank, idx] = 1.0
    tot = fdist.sum_over_ranks(np.concatenate([buf.ravel(), [float(stats["simulation_steps"])]]), device=device)
    steps, buf = tot[-1], tot[:-1].reshape(3 + world, n_tasks)
    if not np.array_equal(buf[2], np.ones(n_tasks)):
        raise RuntimeError(f"shared task queue: tasks run {sorted(set(buf[2].astype(int)))} times (expected exactly once each)")
    init, final = buf[0], buf[1]
    out = {"gpus": world, "episodes": int(n_tasks), "init_coverage": float(init.mean()), "final_coverage": float(final.mean()),
           "episode_delta_coverage": float((final - init).mean()), "simulation_steps": int(steps),
           "schedule": "one shared task queue (atomic counter on the process group's store)",
           "tasks_per_rank": [int(v) for v in buf[3:].sum(axis=1)],
           "note": "coverages and steps over all ranks; best_coverage / episode_length / action counts are rank 0's"}
    bad = int((~np.isfinite(init)).sum() + (~np.isfinite(final)).sum())
    if bad:
        out["non_finite_coverages"] = bad
    return out


def main(argv=None):
    """python -m flingbot_amd.evaluate --tasks set.npz [--weights flingbot.pth] [--slots 96] [--episode-length 10] [--gpus N]

    run_sim.py's evaluation (run_sim.py:37-109 with --eval: fling policy, 12 rotations x 8 scales, obs_dim 64) on a task
    set converted by scripts/convert_tasks_hdf5.py; prints the reference's summary statistics as one JSON line.
    --gpus N > 1 (or a launch under torch.distributed.run): one process per GPU, all ranks pulling from ONE task queue
    (distributed.SharedTaskCounter; --static-blocks: one contiguous block per rank), per-episode coverages merged over RCCL;
    rank 0 prints the statistics of ALL episodes."""
    import argparse
    import json
    import os
    import sys

    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--tasks", required=True, help=".npz task set (flingbot_amd/taskio.py; scripts/convert_tasks_hdf5.py makes it)")
    ap.add_argument("--weights", default=None, help="checkpoint with the reference's state_dict layout (flingbot.pth)")
    ap.add_argument("--slots", type=int, default=96, help="episodes resident on a GPU at a time")
    ap.add_argument("--episode-length", type=int, default=10)
    ap.add_argument("--dump", default=None, metavar="REPLAY.npz",
                    help="also write the per-step episode log (what SimEnv.on_episode_end dumps: learning/Memory.py:106-165) as a "
                         ".npz (taskio.save_replay; rank r of a multi-GPU run writes REPLAY.rank<r>.npz)")
    ap.add_argument("--device", type=int, default=None, help="HIP device (default: LOCAL_RANK, else 0)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--static-blocks", action="store_true",
                    help="several ranks: cut the task set into one contiguous block per rank (rounds 1-5) instead of letting every rank "
                         "pull from one shared queue like the reference's environments pull from one TaskLoader")
    a = ap.parse_args(argv)
    if a.device is not None and (a.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1):
        ap.error("--device names ONE HIP device: with --gpus N (or under torch.distributed.run) every rank takes LOCAL_RANK")
    if a.gpus > 1 and os.environ.get("WORLD_SIZE") is None:   # start the ranks ourselves, before this process touches the GPU
        from .launch import launch_local_ranks
        # (`python -m flingbot_amd.evaluate ...` again, once per rank, from the same working directory)
        sys.exit(launch_local_ranks(a.gpus, "-m", ["flingbot_amd.evaluate"] + list(sys.argv[1:] if argv is None else argv)))

    from . import distributed as fdist, nets, sim as fsim, taskio
    from .env import BatchedFlingEnv

    rank, local_rank, world = fdist.init_from_env("nccl") if os.environ.get("WORLD_SIZE") else (0, 0, 1)
    device = local_rank if a.device is None else a.device
    tasks = taskio.TaskLoader(a.tasks, repeat=False).all_tasks()
    per_rank = (len(tasks) + world - 1) // world
    shared = world > 1 and not a.static_blocks
    mine = tasks if shared else tasks[rank * per_rank:(rank + 1) * per_rank]
    dev = f"cuda:{device}"
    torch.cuda.set_device(device)
    ctx = fsim.FlingSim(n_envs=max(1, min(a.slots, len(mine))), device=device, solver=0)
    census = None
    if world > 1:
        # what the process group itself saw (one all_gather of rank / LOCAL_RANK / device identity): one process per GPU is the
        # contract of the sharded run -- two ranks on one device would report N GPUs' worth of throughput from fewer
        try:
            arch = torch.cuda.get_device_properties(device).gcnArchName
        except Exception:
            arch = ""
        census = fdist.rank_census(ctx.device_key(), arch)
        if census["ranks_seen"] != world or census["distinct_devices"] != world:
            ctx.close()
            raise SystemExit(f"evaluate: WORLD_SIZE={world} but the process group's all_gather saw {census['ranks_seen']} rank(s) on "
                             f"{census['distinct_devices']} distinct device(s) ({census['backend']}): one process per GPU, or use "
                             f"--device with a single process")
    env = BatchedFlingEnv(ctx, episode_length=a.episode_length, device=dev)
    policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                     obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                     depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                     value_expl_decay=1.0, device=dev)
    if a.weights:
        ckpt = torch.load(a.weights, map_location=dev)
        policy.load_state_dict(ckpt.get("net", ckpt))          # utils.py:116-118 stores the module under 'net'
    if shared:
        # one queue for all ranks (distributed.SharedTaskCounter: an atomic counter on the process group's store): whoever has a
        # free slot takes the next task, exactly what utils.setup_envs' shared TaskLoader actor does for the reference's workers
        counter = fdist.SharedTaskCounter(len(tasks), key=os.path.basename(a.tasks))
        stats = run_tasks(policy, env, tasks, claim=counter.claim, claim_first=per_rank)
        if a.dump and len(stats["task_indices"]):
            path = a.dump[:-4] + f".rank{rank}.npz" if a.dump.endswith(".npz") else a.dump + f".rank{rank}"
            taskio.save_replay(path, stats["records"], [tasks[i] for i in stats["task_indices"]], episode_ids=stats["task_indices"])
    elif mine:
        stats = run_tasks(policy, env, mine)
        if a.dump:
            path = a.dump if world == 1 else a.dump[:-4] + f".rank{rank}.npz" if a.dump.endswith(".npz") else a.dump + f".rank{rank}"
            taskio.save_replay(path, stats["records"], mine, first_episode=rank * per_rank)
    else:  # more ranks than task blocks (13 tasks over 8 GPUs: blocks of 2, the last rank has none): nothing to run, still gathers
        nothing = np.zeros(0, np.float32)
        stats = {"init_coverage": nothing, "final_coverage": nothing, "simulation_steps": 0,
                 "action_primitive_counts": {a_: 0 for a_ in env.actions},
                 "mean": {k: float("nan") for k in ("init_coverage", "final_coverage", "best_coverage", "episode_delta_coverage",
                                                    "episode_length")}}
    ctx.close()
    out = {"tasks": len(tasks), **stats["mean"], "action_primitive_counts": stats["action_primitive_counts"],
           "simulation_steps": stats["simulation_steps"]}
    if world > 1:
        out.update(merge_shared_statistics(stats, len(tasks), device=dev) if shared else merge_rank_statistics(stats, per_rank, device=dev))
        out.update({k: census[k] for k in ("ranks_seen", "distinct_devices", "backend", "collective_library")})
        fdist.barrier()
    if rank == 0:
        print(json.dumps(out))
    if os.environ.get("WORLD_SIZE") and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    # `python -m flingbot_amd.evaluate` executes this file a second time as __main__; run the package's own instance of the
    # module instead, so that there is one copy of its state (and one place for anything that wraps its functions)
    from flingbot_amd.evaluate import main as _package_main

    _package_main()
