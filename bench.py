#!/usr/bin/env python
"""bench.py -- headline benchmark: cloth sim steps/sec on 64x64-particle cloths (BASELINE.json `metric`).

One bench "step" = one pyflex.step() (dt 1/100 s = 4 substeps x 30 solver iterations) of EVERY episode resident on a
GPU: `--episodes` independent 64x64 cloth episodes per GPU (default 256 = one per CU), advanced by ONE launch of the
fused LDS-resident solver kernel.  value = episode-steps/s summed over all GPUs (weak scaling: episodes per GPU fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--episodes E]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (synthetic, seeded): every episode is a 64x64 grid cloth (scene_params of BASELINE.md C2) hung vertically
above the ground with a per-episode random perturbation and released, so the timed steps cover free fall, ground
contact with friction and heavy self-collision while the sheet crumples.
The JSON line carries `roofline` (algorithmic HBM bytes per launch / HIP-event kernel time vs the 8 TB/s peak) and, at
N=1, `cpu_baseline` (the C oracle on the host cores this process may use, one independent episode of the same workload per core,
about 15 s) and `perception` (the value network's forward of one observation, measured after the timed region).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM = 64
N_PART = DIM * DIM
N_SPRINGS = 23938
SUBSTEPS, ITERS = 4, 30
# algorithmic bytes per pyflex.step() = S x [112 N + I x (32 N + 16 M)]
BYTES_PER_STEP = SUBSTEPS * (112 * N_PART + ITERS * (32 * N_PART + 16 * N_SPRINGS))   # 63,524,608
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md chip table (spec)


def scene_params():
    return np.array([0, 0.2, 0, DIM, DIM, 0.9, 0.9, 0.9, 2, 0, 2, 0, np.pi / 2, -np.pi / 2, 0, 720, 720, 0.5, 0],
                    dtype=np.float64)


def initial_state(seed, inv_mass):
    """Vertical sheet in the x-y plane, lower edge 2 cm above the ground, small seeded perturbation."""
    rng = np.random.RandomState(seed)
    sp = 0.00625
    xs = (np.arange(DIM) - (DIM - 1) / 2.0) * sp
    ys = 0.02 + np.arange(DIM) * sp
    xx, yy = np.meshgrid(xs, ys)
    p = np.zeros((N_PART, 4), np.float32)
    p[:, 0] = xx.ravel()
    p[:, 1] = yy.ravel()
    p[:, 2] = 0.0
    p[:, :3] += (rng.rand(N_PART, 3).astype(np.float32) - 0.5) * 0.002
    p[:, 3] = inv_mass
    return p


def setup_episode(sim, seed):
    sim.set_scene(scene_params())
    w = sim.get_positions().reshape(-1, 4)[0, 3]
    sim.set_positions(initial_state(seed, w).ravel())
    sim.set_velocities(np.zeros(3 * N_PART, np.float32))


def cpu_baseline(warmup, budget_s=15.0):
    """Oracle (C restatement) on the host: one independent episode of the same workload per core, all cores at once (the
    solver step is single-threaded; episodes are what parallelises, exactly like the reference's one-process-per-env
    layout), bounded to ~budget_s of wall time.  `value` is the aggregate over the cores used."""
    import threading
    from oracle import OracleSim

    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    try:  # a container's CPU quota (cgroup v2 cpu.max / v1 cfs) can be far below the visible CPU count
        quota = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else float(q) / float(per)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        if quota is not None:
            cores = max(1, min(cores, int(quota + 0.5)))
    except (OSError, ValueError):
        pass
    sims = []
    for k in range(cores):
        o = OracleSim()
        setup_episode(o, seed=k)
        sims.append(o)
    done = [0] * cores
    stop = threading.Event()

    def work(k):  # ctypes releases the GIL around orc_step
        sims[k].step(warmup)
        ready.wait()
        while not stop.is_set():
            sims[k].step(5)
            done[k] += 5

    ready = threading.Barrier(cores + 1)
    threads = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(cores)]
    for t in threads:
        t.start()
    ready.wait()
    t0 = time.perf_counter()
    time.sleep(budget_s)
    stop.set()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    total = sum(done)
    return {"value": total / dt, "unit": "sim steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} independent episodes (64x64 cloth, the GPU episodes' initial states 0..{cores - 1}), one per "
                      f"host core, {total} pyflex.step() in {dt:.1f} s after {warmup} warm-up steps each, C oracle"}


def traffic_from_profile(episodes):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
        per_ep = rec["bytes_per_launch"] / rec["episodes"]
        return per_ep * episodes
    except Exception:
        return None


def perception_leg(device):
    """Secondary, after the timed region (rank 0, N = 1): the per-action perception kernels of the same path on this GPU --
    SpatialValueNet forward of one observation (96 transforms x 64 x 64, random-init weights, hand-written fp32-MFMA
    kernels) -- so the round's BENCH file carries them next to the solver number.  Not part of `value`."""
    try:
        import torch
        from flingbot_amd import nets
        torch.manual_seed(0)
        net = nets.SpatialValueNet(rgb_only=True, device=device).to(device).eval().fold_batchnorm()
        obs = torch.rand(96, 4, 64, 64, device=device)
        with torch.no_grad():
            for _ in range(5):
                net(obs)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(50):
                net(obs)
            torch.cuda.synchronize(device)
        ms = (time.perf_counter() - t0) / 50 * 1e3
        return {"value_net_ms_per_observation": ms, "value_net_useful_tflops_f32": 96 * 306.7e6 / (ms * 1e-3) / 1e12,
                "value_net_path": "fs_value_net_forward" if net._hip is not None else "pytorch"}
    except Exception as exc:  # the headline number must not depend on this leg
        return {"error": str(exc)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--episodes", type=int, default=256, help="cloth episodes per GPU")
    ap.add_argument("--solver", type=int, default=2, help="2 fused LDS kernel (default), 1 streaming kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from flingbot_amd import distributed as fdist
    from flingbot_amd import sim as fsim

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: launch with torch.distributed.run")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, local_rank, world = fdist.init_from_env("nccl")  # one process per GPU; "nccl" is RCCL on ROCm

    E = args.episodes
    ctx = fsim.FlingSim(n_envs=E, device=local_rank, solver=args.solver)
    for e, g in enumerate(fdist.episode_range(rank, E)):
        setup_episode(ctx.env(e), seed=g)  # global episode id = seed
    ctx.sync()

    def barrier():
        fdist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.step(1)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()                      # HIP events on the stream the kernels are launched on
    for _ in range(args.steps):
        ctx.step(1)                        # one launch of the fused kernel: every episode advances one frame
    kern_ms_total = ctx.timer_stop()
    # episode-batch gather of the coverage rewards (the only exchange step of the path)
    cov_all = fdist.gather_rewards(ctx.coverage(), device="cuda")
    barrier()
    elapsed = fdist.max_over_ranks(time.perf_counter() - t0, device="cuda")

    if rank == 0:
        total_steps = E * world * args.steps
        value = total_steps / elapsed
        kern_ms = kern_ms_total / args.steps
        achieved = BYTES_PER_STEP * E / (kern_ms * 1e-3) / 1e9
        traffic = traffic_from_profile(E)
        out = {
            "metric": "sim steps/sec (64x64-particle cloth)",
            "value": value,
            "unit": "sim steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{E} x 64x64-particle cloth episodes per GPU (4096 particles, 23938 springs, "
                                   f"self-collision + ground friction; vertical sheet released and crumpling), one "
                                   f"pyflex.step() (4 substeps x 30 iterations) of every episode per bench step",
                       "episodes_per_gpu": E, "cloth": "64x64", "substeps": SUBSTEPS, "iterations": ITERS,
                       "solver": "fused-lds" if args.solver == 2 else "stream", "parallelism": f"episodes x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "fs_k_fused_step" if args.solver == 2 else "fs_k_iterate (+stage kernels)",
                         "kernel_ms_per_launch": kern_ms, "algorithmic_bytes_per_launch": BYTES_PER_STEP * E},
            "mean_coverage": float(cov_all.mean().item()),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.warmup)
        if world == 1:
            out["perception"] = perception_leg(torch.device("cuda", local_rank))
        print(json.dumps(out), flush=True)
    fdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
