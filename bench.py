#!/usr/bin/env python
"""bench.py -- headline benchmark: cloth sim steps/sec on 64x64-particle cloths (BASELINE.json `metric`).

One bench "step" = one pyflex.step() (dt 1/100 s = 4 substeps x 30 solver iterations) of EVERY episode resident on a
GPU: `--episodes` independent 64x64 cloth episodes per GPU (default 256 = one per CU), advanced by ONE launch of the
fused LDS-resident solver kernel.  value = episode-steps/s summed over all GPUs (weak scaling: episodes per GPU fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--episodes E]      # N > 1: starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                               # or ranks started by torchrun

Workload (synthetic, seeded): every episode is a 64x64 grid cloth (scene_params of BASELINE.md C2) hung vertically
above the ground with a per-episode random perturbation and released; `--preroll` (60) untimed frames bring it to the
state the timed window should see whatever --steps is: the sheet crumpled on the ground (ground friction + heavy
self-collision, the expensive regime).
The JSON line carries `roofline` (algorithmic HBM bytes per launch / HIP-event kernel time vs the 8 TB/s peak; `basis`
and `limiter` say what that number is and what really bounds the LDS-resident kernel), `parity` (one episode of the
batch that was timed, compared with the C oracle after every timed region), `valu_roofline` (lane-operations/s against the
39.3 T/s non-packed VALU issue peak: the roofline that physically bounds the kernel), `configs` (the 64-episodes-per-GPU
figure of BASELINE.json configs[2] / configs[3] next to the headline: median of three 100-frame windows over the same frames) and, at N=1, `cpu_baseline` (the C oracle on the host
cores this process may use, one independent episode of the same workload per core, about 15 s), `perception` (the
value network's forward of one observation) and `eval_loop` (BASELINE.json configs[4]: the run_sim.py evaluation loop
on 32 generated tasks at the reference's sizes, and 192 tasks streamed through 96 slots), both measured after the timed
region.
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
BYTES_PER_STEP = SUBSTEPS * (112 * N_PART + ITERS * (32 * N_PART + 16 * N_SPRINGS))   # 63,524,608 (= algorithmic_bytes_per_step(64, 64))
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md chip table (spec)


def algorithmic_bytes_per_step(dimx, dimz):
    """SURVEY.md 8(d)'s streaming model for a dimx x dimz grid cloth: S x [112 N + I x (32 N + 16 M)] with the spring count of
    helpers.h:838-924 (stretch + bend + shear)."""
    n = dimx * dimz
    m = (dimx - 1) * dimz + dimx * (dimz - 1) + max(dimx - 2, 0) * dimz + dimx * max(dimz - 2, 0) + 2 * (dimx - 1) * (dimz - 1)
    return SUBSTEPS * (112 * n + ITERS * (32 * n + 16 * m))


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


def host_cores():
    """Cores this process may really use: the affinity mask, cut down to a container's CPU quota (cgroup v2 cpu.max / v1 cfs)."""
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    try:
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
    return cores


def _time_host_episodes(variant, n_threads, warmup, budget_s):
    """n_threads independent episodes of the bench workload on the C restatement `variant`, one per thread, for ~budget_s
    seconds after `warmup` untimed frames each; returns (pyflex.step() calls completed, seconds)."""
    import threading
    from oracle import OracleSim

    sims = []
    for k in range(n_threads):
        o = OracleSim(variant)
        setup_episode(o, seed=k)
        sims.append(o)
    done = [0] * n_threads
    stop = threading.Event()
    ready = threading.Barrier(n_threads + 1)

    def work(k):  # ctypes releases the GIL around orc_step
        sims[k].step(warmup)
        ready.wait()
        while not stop.is_set():
            sims[k].step(5)
            done[k] += 5

    threads = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(n_threads)]
    for t in threads:
        t.start()
    ready.wait()
    t0 = time.perf_counter()
    time.sleep(budget_s)
    stop.set()
    for t in threads:
        t.join()
    return sum(done), time.perf_counter() - t0


def cpu_baseline(warmup, budget_s=15.0):
    """The CPU path next to the GPU figure ("(a) single-thread and (b) OpenMP over episodes on all host cores").
    The reference has no CPU solver, so this is the repository's own C restatement of the step -- and of its builds the FASTEST
    honest one: `host` = oracle/flex_oracle.c with every length through the CPU's own correctly rounded sqrt / divide
    (-DORC_EXACT_RSQRT -O3 -mfma), which stays within 1.1e-7 of the shipped step per step
    (tests/test_oracle_cpu.py::test_approximations_stay_within_1e_4_of_exact_math).  The bit-exact parity checker is NOT what
    is timed: since round 4 it emulates gfx950's v_rsq_f32 through a 4 MiB table and a double-precision sqrt per length, which
    is the cost of checking a GPU instruction, not of simulating cloth on a CPU (round 3: 427 steps/s, round 4: 290 on the same
    16 cores, same algorithm).  (a) one episode on one thread, (b) one independent episode per usable core (episodes are what
    parallelises, as in the reference's one-process-per-env layout); ~budget_s seconds of wall time in total.
    `value` / `cores` = (b)."""
    cores = host_cores()
    variant = "host"
    n1, t1 = _time_host_episodes(variant, 1, warmup, budget_s * 0.35)
    nc, tc = _time_host_episodes(variant, cores, warmup, budget_s * 0.65)
    return {"value": nc / tc, "unit": "sim steps/s", "cores": cores, "kind": "port",
            "arithmetic": "host: C restatement with IEEE sqrtf / divide for every length (-DORC_EXACT_RSQRT -O3 -mfma), <= 1.1e-7 "
                          "per step from the shipped step; the table-driven bit-exact checker is kept for parity only",
            "single_thread": {"value": n1 / t1, "steps": n1, "seconds": round(t1, 2)},
            "all_cores": {"value": nc / tc, "cores": cores, "steps": nc, "seconds": round(tc, 2)},
            "sample": f"64x64 cloth, the GPU episodes' initial states: (a) episode 0 on one thread, {n1} pyflex.step() in {t1:.1f} s; "
                      f"(b) episodes 0..{cores - 1}, one per host core, {nc} pyflex.step() in {tc:.1f} s; each after {warmup} "
                      f"untimed frames (pre-roll + warm-up, as on the GPU)"}


def traffic_from_profile(episodes):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/), or (None, None).  NOT measured in this
    run: it is the PMC pass of the same command (scripts/profile_bench.sh), scaled per episode; `traffic_source` says so."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
        per_ep = rec["bytes_per_launch"] / rec["episodes"]
        return per_ep * episodes, f"profiles/hbm_traffic.json (rocprofv3 --pmc pass '{rec.get('tag', '?')}' of this command, " \
                                  f"FETCH_SIZE x2 + WRITE_SIZE per launch at {rec['episodes']} episodes, scaled per episode)"
    except Exception:
        return None, None


def profile_tags(suffix):
    """Round tags (r06, r05, ...) that have profiles/<tag>_<suffix>, newest first: the bench line quotes the latest committed
    profile of each kind without anybody editing a list."""
    import re
    try:
        names = os.listdir(os.path.join(ROOT, "profiles"))
    except OSError:
        return []
    tags = {m.group(1) for n in names for m in [re.match(r"(r\d\d)_" + re.escape(suffix) + "$", n)] if m}
    return sorted(tags, reverse=True)


def limiter_from_profile():
    """What the committed PMC counters say actually limits the fused kernel (it keeps the iterations in LDS, so the
    contract's algorithmic-bytes `roofline` is an equivalent streamed bandwidth, not HBM traffic)."""
    for tag in profile_tags("pmc.json"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc.json")
        try:
            with open(path) as fh:
                rec = json.load(fh)
            c = rec["counters"]
            valu = c["SQ_ACTIVE_INST_VALU"]["avg_per_launch"] / c["SQ_WAVE_CYCLES"]["avg_per_launch"]
            return {"bound": "valu-issue", "source": f"profiles/{tag}_pmc.json", "episodes": rec.get("episodes", 256),
                    "valu_active_per_wave_cycle": valu, "waves_per_simd": rec.get("waves_per_simd", 4),
                    "note": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: with 4 resident waves per SIMD a VALU that never idles gives "
                            "~0.25 per wave-cycle; the counter pair is an indicator, not a calibrated ceiling -- the bounded "
                            "figure is `valu_roofline` (lane-operations per second against the issue peak)",
                    "valu_instructions_per_launch": c["SQ_INSTS_VALU"]["avg_per_launch"],
                    "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"]["total_corrected"],
                    "hbm_gbs": rec["hbm_bytes_per_launch"]["total_corrected"] / (rec["average_us"] * 1e-6) / 1e9,
                    "hbm_frac_of_peak": rec["hbm_bytes_per_launch"]["total_corrected"] / (rec["average_us"] * 1e-6) / 1e9
                                        / HBM_PEAK_GBS}
        except Exception:
            continue
    return None


VALU_PEAK_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12   # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s (one
                                                    # non-packed fp32 VALU instruction per SIMD and 4 cycles per wave64)


def valu_roofline(limiter, episodes, kern_ms):
    """The roofline that physically bounds the LDS-resident kernel: VALU issue slots per second against the non-packed
    issue peak.  Instructions per launch come from the committed PMC pass (`limiter`: SQ_INSTS_VALU), scaled per episode;
    a quarter-rate instruction occupies the VALU for four slots, and the one such instruction of the inner loops is the
    reciprocal root -- v_rsq_f32, one per spring endpoint = 12 per particle and iteration (counted analytically; contacts add
    a few per cent more) -- so three extra slots are charged for each; the time is this run's HIP-event kernel time."""
    if not limiter or "valu_instructions_per_launch" not in limiter:
        return None
    per_ep = limiter["valu_instructions_per_launch"] / limiter.get("episodes", 256)
    rsq_per_ep = N_PART * 12 * SUBSTEPS * ITERS / 64.0 if limiter["source"].split("/")[-1] >= "r04" else 0.0
    slots = (per_ep + 3.0 * rsq_per_ep) * episodes
    achieved = slots * 64 / (kern_ms * 1e-3) / 1e12
    return {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-slots/s",
            "frac": achieved / VALU_PEAK_TLANEOPS, "wave_instructions_per_launch": per_ep * episodes,
            "quarter_rate_wave_instructions_per_launch": rsq_per_ep * episodes,
            "valu_busy_fraction_pmc": limiter.get("valu_active_per_wave_cycle", 0.0) * limiter.get("waves_per_simd", 4),
            "source": f"SQ_INSTS_VALU of {limiter['source']} (per episode) + 3 extra slots per v_rsq_f32, x 64 lanes / this run's "
                      f"kernel time; peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz; valu_busy_fraction_pmc = SQ_ACTIVE_INST_VALU x "
                      f"waves per SIMD / SQ_WAVE_CYCLES of the same pass (how much of the time a SIMD's VALU is occupied)"}


def stream_limiter_from_profile():
    """What the committed profiles say about the 64-episodes-per-GPU launch shape (streaming back-end, two launch
    chains): per-kernel share and duration of one frame's 129 dependent launches."""
    import csv
    import re
    for tag in profile_tags("stream64_kernel_stats.csv"):
        path = os.path.join(ROOT, "profiles", f"{tag}_stream64_kernel_stats.csv")
        try:
            rows = list(csv.reader(open(path)))[1:]
            ker = [{"kernel": r[0].split("(")[0].replace("void ", ""), "calls": int(r[1]), "average_us": float(r[3]),
                    "percent": float(r[4])} for r in rows if float(r[4]) >= 1.0]
            wave = "one wave's critical path"
            try:   # the PMC pass of the same launch shape (scripts/pmc_stream64.sh), when the round committed one
                txt = open(os.path.join(ROOT, "profiles", f"{tag}_stream64_pmc.txt")).read()
                m = re.search(r"fs_k_iterate_gridl\S* launches \d+\s+waves \d+\s+VALU/wave (\d+).*?valu_active/wave_cycles ([0-9.]+)", txt)
                if m:
                    wave = (f"one wave's critical path (fs_k_iterate_gridl: {m.group(1)} VALU instructions per wave at {m.group(2)} "
                            f"VALU-active per wave-cycle, profiles/{tag}_stream64_pmc.txt; 0.25 = a VALU that never idles at 4 waves per SIMD)")
            except Exception:
                pass
            return {"bound": "launch-latency", "source": f"profiles/{tag}_stream64_kernel_stats.csv",
                    "dependent_launches_per_frame": 129, "kernels": ker,
                    "note": f"each launch is {wave} plus dispatch/drain; two concurrent chains hide part of the turn-around "
                            "(DESIGN.md 4.2)"}
        except Exception:
            continue
    return None


def eval_limiter_from_profile():
    """What the committed counters say bounds the evaluation loop's launches: the PMC pass over the loop-less
    reconstruction of its launch shape -- 132 stored hard tasks in their crumpled states, scripts/profile_shapes.sh case A (a PMC
    pass over the loop itself is impractical: counter collection serialises its ~2 M dispatches)."""
    import re
    for tag in profile_tags("eval_shapes.txt"):
        try:
            txt = open(os.path.join(ROOT, "profiles", f"{tag}_eval_shapes.txt")).read()
            block = txt.rsplit("# PMC pass, case A", 1)[1].split("# PMC pass, case B", 1)[0]   # the newest pass of the file
            ker = {}
            for m in re.finditer(r"^(?:void )?(fs_k_\w+).*?\s+launches\s+(\d+)\s+waves/launch\s+(\d+)\s+VALU/wave\s+(\d+)\s+"
                                 r"valu_active/wave_cycles ([0-9.]+)", block, re.M):
                ker[m.group(1)] = {"waves_per_launch": int(m.group(3)), "valu_per_wave": int(m.group(4)),
                                   "valu_active_per_wave_cycle": float(m.group(5))}
            name, waves = ("fs_k_iterate_gridl_tp", 6) if "fs_k_iterate_gridl_tp" in ker else ("fs_k_iterate_gridl", 5)
            if name not in ker:
                continue
            it = ker[name]
            return {"bound": "valu-issue", "source": f"profiles/{tag}_eval_shapes.txt (case A: 132 stored tasks, crumpled, plain fs_step)",
                    "kernels": ker, "iterate_kernel": name, "iterate_waves_per_simd": waves,
                    "iterate_valu_busy_fraction": it["valu_active_per_wave_cycle"] * waves,
                    "note": f"{name} (~79 % of the loop's kernel time) runs {waves} waves per SIMD, so VALU-active per wave-cycle tops out "
                            f"at 1/{waves}: at this launch size the kernel is VALU-bound on instructions of which ~40 % are contact "
                            "evaluations -- not the one-wave critical path of the 64-episode launch (DESIGN.md 6 has the frame's "
                            "accounting: uniform benchmark, + ground contact and size mix, + particle contacts, + the loop itself)"}
        except Exception:
            continue
    return None


def oracle_trajectory(seed, steps):
    """The checker: one episode of the workload on the C oracle, `steps` frames from the initial state."""
    from oracle import OracleSim

    o = OracleSim()
    setup_episode(o, seed=seed)
    o.step(steps)
    return o.get_positions(), o.get_velocities()


def mfma_from_profile():
    """MFMA utilisation of the value network's block kernel from the committed rocprofv3 --pmc pass (scripts/profile_cnn.sh):
    SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 flops over the kernel's summed duration = fp32 MFMA flops ISSUED per second (useful work
    + the 12.5 % halo the fused blocks recompute), against the 157.3 TFLOP/s dense fp32-MFMA peak."""
    import re
    for tag in profile_tags("cnn_pmc.txt"):
        path = os.path.join(ROOT, "profiles", f"{tag}_cnn_pmc.txt")
        try:
            for line in open(path):
                if line.startswith("fs_k_vn_block") and "SQ_INSTS_VALU_MFMA_MOPS_F32" in line:
                    us = float(re.search(r"([0-9.]+) us total", line).group(1))
                    mops = float(re.search(r"SQ_INSTS_VALU_MFMA_MOPS_F32=([0-9.e+]+)", line).group(1))
                    n = int(re.search(r"launches\s+(\d+)", line).group(1))
                    tf = mops * 512 / (us * 1e-6) / 1e12
                    return {"source": f"profiles/{tag}_cnn_pmc.txt", "kernel": "fs_k_vn_block", "launches": n,
                            "average_us": us / n, "mfma_tflops_issued_f32": tf, "mfma_frac_of_peak": tf / 157.3,
                            "counter": "SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 / summed kernel duration"}
        except Exception:
            continue
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
        out = {"value_net_ms_per_observation": ms, "value_net_useful_tflops_f32": 96 * 306.7e6 / (ms * 1e-3) / 1e12,
               "value_net_fraction_of_f32_mfma_peak": 96 * 306.7e6 / (ms * 1e-3) / 1e12 / 157.3,
               "value_net_path": "fs_value_net_forward" if net._hip is not None else "pytorch"}
        out["value_net_mfma"] = mfma_from_profile()
        out.update(render_leg(device.index or 0))
        return out
    except Exception as exc:  # the headline number must not depend on this leg
        return {"error": str(exc)[:200]}


RENDER_BYTES = 720 * 720 * (4 + 4) + N_PART * 32 + 7938 * 12      # RGBA8 + depth out, positions + normals + triangles in


def render_leg(device_index, frames=30):
    """BASELINE.json configs[1] (64 x 64 cloth + depth render, one GPU): one crumpled bench episode with the two pickers
    parked where SimEnv leaves them; `pyflex.render()` as the reference calls it (720 x 720 RGBA + depth downloaded to the
    host, pyflex.cpp:924-1133) and the device-resident observation (render -> 400 x 400 obs + cloth mask + bounding box,
    nothing downloaded but 5 integers), milliseconds per call and the render's 4.4 MB of algorithmic bytes per second."""
    from flingbot_amd import sim as fsim
    ctx = fsim.FlingSim(n_envs=1, device=device_index)
    setup_episode(ctx.env(0), seed=0)
    for c in ((0.5, 0.5, -0.5), (-0.5, 0.5, -0.5)):
        ctx.env(0).add_sphere(0.02, c, [1, 0, 0, 0])
    ctx.step(80)
    ctx.render(0); ctx.observe(0, 400)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        ctx.render(0)
    t_render = (time.perf_counter() - t0) / frames
    t0 = time.perf_counter()
    for _ in range(frames):
        ctx.observe(0, 400)
    ctx.sync()
    t_obs = (time.perf_counter() - t0) / frames
    ctx.close()
    return {"render": {"baseline_config": "configs[1]", "cloth": "64x64 crumpled + 2 picker spheres", "render_dim": 720,
                       "pyflex_render_ms": t_render * 1e3, "pyflex_render_includes": "2 x 2.07 MB download to the host (the reference's return value)",
                       "render_observe_device_ms": t_obs * 1e3, "observe_dim": 400,
                       "algorithmic_bytes_per_frame": RENDER_BYTES,
                       "device_path_gbs": RENDER_BYTES / t_obs / 1e9,
                       "device_path_frac_of_hbm_peak": RENDER_BYTES / t_obs / 1e9 / HBM_PEAK_GBS,
                       "note": "4.4 MB per frame: a latency-bound chain of small kernels (sphere mesh, normals, two raster passes, "
                               "shade, resize, labelling rounds), not a bandwidth problem; profiles/r04_render_kernel_stats.csv"}}


def timed_batch(ctx, fdist, torch, steps, warmup, preroll):
    """Pre-roll + warm-up (untimed), then `steps` frames of every episode of `ctx` bracketed by barrier + synchronize on
    both sides.  Returns (wall seconds of the timed region, MAX over ranks; HIP-event kernel milliseconds of this rank)."""
    def barrier():
        fdist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    for _ in range(preroll):
        ctx.step(1)                        # bring the workload to its steady crumpled state (untimed, not warm-up); one
                                           # frame per launch like the timed steps, so a profile sees launches of ONE shape
    for _ in range(warmup):
        ctx.step(1)
    # the exchange step is warmed up like the kernels: the first collective of a process group builds RCCL's communicators
    # (hundreds of milliseconds) and the first device tensors go through the allocator -- neither belongs in a timed window
    fdist.gather_rewards(ctx.coverage(), device="cuda")
    fdist.max_over_ranks(0.0, device="cuda")
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()                      # HIP events on the stream the kernels are launched on
    for _ in range(steps):
        ctx.step(1)                        # fused: ONE launch advances every episode one frame
    kern_ms_total = ctx.timer_stop()
    # episode-batch gather of the coverage rewards (the only exchange step of the path)
    cov_all = fdist.gather_rewards(ctx.coverage(), device="cuda")
    barrier()
    elapsed = fdist.max_over_ranks(time.perf_counter() - t0, device="cuda")
    return elapsed, kern_ms_total, cov_all


def timed_windows(ctx, fdist, torch, steps, warmup, preroll, windows=3):
    """The 64-episodes-per-GPU figure: pre-roll + warm-up (untimed), then `windows` timed windows of `steps` frames that
    all start from the SAME state (positions and velocities of every episode are put back in between), each bracketed by
    barrier + synchronize like the headline.  Identical work per window, so their spread is run-to-run noise of the
    launch-bound path and nothing else.  Returns a list of (wall seconds MAX over ranks, HIP-event kernel ms) and the
    gathered coverage of the last window."""
    def barrier():
        fdist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    for _ in range(preroll + warmup):
        ctx.step(1)
    fdist.gather_rewards(ctx.coverage(), device="cuda")   # (warm: see timed_batch)
    fdist.max_over_ranks(0.0, device="cuda")
    ctx.sync()
    snap = [(ctx.get_positions(e), ctx.get_velocities(e)) for e in range(ctx.n_envs)]
    out, cov_all = [], None
    for w in range(windows):
        for e, (p, v) in enumerate(snap):  # device state is the truth: both calls upload at once, nothing is deferred
            ctx.set_positions(e, p)        # into the timed region
            ctx.set_velocities(e, v)
        barrier()
        t0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(steps):
            ctx.step(1)
        kern_ms_total = ctx.timer_stop()
        cov_all = fdist.gather_rewards(ctx.coverage(), device="cuda")
        barrier()
        out.append((fdist.max_over_ranks(time.perf_counter() - t0, device="cuda"), kern_ms_total))
    return out, cov_all


class ParityCheck:
    """The checker after the timed region: episode `env` of the batch that was just timed against the C oracle.
    Full trajectory from the initial state when that is affordable on one host core (the oracle then runs in a thread
    while the bench goes on; `result()` joins it), else the state is handed to the oracle and both sides advance 5 more
    frames (the GPU side as the same batched launch that was timed)."""

    def __init__(self, ctx, env, seed, total_steps, max_oracle_steps=450):
        import threading
        from oracle import OracleSim

        self.seed, self.thread, self.ref = int(seed), None, None
        if total_steps <= max_oracle_steps:
            # the oracle thread is only CREATED here (the GPU state is captured now); start() runs it -- after every
            # timed region of this process, so that no checker shares the host with a launch-bound timed loop
            self.mode = f"full trajectory, {total_steps} frames from the initial state"
            self.gpu = (ctx.get_positions(env), ctx.get_velocities(env))
            self.thread = threading.Thread(target=self._run, args=(seed, total_steps), daemon=True)
        else:
            self.mode = f"5 frames from the state after {total_steps} frames"
            o = OracleSim()
            setup_episode(o, seed=seed)
            o.set_positions(ctx.get_positions(env))
            o.set_velocities(ctx.get_velocities(env))
            ctx.step(5)
            o.step(5)
            self.ref = (o.get_positions(), o.get_velocities())
            self.gpu = (ctx.get_positions(env), ctx.get_velocities(env))

    def _run(self, seed, steps):
        self.ref = oracle_trajectory(seed, steps)

    def start(self):
        if self.thread is not None and not self.thread.is_alive() and self.ref is None:
            self.thread.start()

    def result(self):
        if self.thread is not None:
            self.start()
            self.thread.join()
        (ph, vh), (po, vo) = self.gpu, self.ref
        exact = bool(np.array_equal(ph.view(np.uint32), po.view(np.uint32)) and
                     np.array_equal(vh.view(np.uint32), vo.view(np.uint32)))
        rel = float(np.abs(ph - po).max() / max(1.0, float(np.abs(po).max())))
        return {"parity_checked": True, "bit_exact": exact, "max_rel_position_error": rel, "within_1e-4": rel <= 1e-4,
                "episode": self.seed, "mode": self.mode, "checker": "oracle/flex_oracle.c"}


# ---------------------------------------------------------------- crumple + scripted two-corner fling
C2_RAISE, C2_HOLD, C2_SETTLE, C2_FLING_SETTLE = 200, 100, 150, 300


def c2_crumple(ctx, seeds):
    """The deterministic "hard task" crumple of SURVEY 8(d) C2 for every episode of `ctx` (tests/scenarios.py scenario_c2 is
    the same recipe for one episode of any solver; mirror of environment/tasks.py:177-224): flattened 64 x 64 sheet,
    center_object's step, particle seed % N pinned and raised to 0.5 + u over 200 steps, held 100, released, 150 settle
    steps.  The pinned particle is rewritten on the device (fs_set_particles) instead of through 2 x 64 KiB per step."""
    import scenarios as sc

    E = ctx.n_envs
    flat = sc.set_to_flatten_positions(DIM, DIM).flatten()
    for e in range(E):
        env = ctx.env(e)
        env.set_scene(sc.survey_params(DIM))
        env.step(1)
        env.set_positions(flat)
        pos = env.get_positions().reshape(-1, 4).copy()
        pos[:, [0, 2]] -= np.mean(pos[:, [0, 2]], axis=0, keepdims=True)
        env.set_positions(pos.ravel())
    ctx.step(1)
    ks = np.array([int(sd) % N_PART for sd in seeds], np.int32)
    heights = np.array([float(np.random.RandomState(int(sd)).random_sample(1)[0]) + 0.5 for sd in seeds])
    pick = np.stack([ctx.get_positions(e).reshape(-1, 4)[ks[e]].copy() for e in range(E)]).astype(np.float32)
    w0 = pick[:, 3].copy()
    init_h = pick[:, 1].copy()
    pick[:, 3] = 0.0
    envs = np.arange(E, dtype=np.int32)
    speed = 1.0 / C2_RAISE
    for j in range(C2_RAISE + C2_HOLD):
        if j < C2_RAISE:
            pick[:, 1] = ((heights - init_h.astype(np.float64)) * (j * speed) + init_h).astype(np.float32)
        ctx.set_particles(envs, ks, pick, zero_velocity=True)
        ctx.step(1)
    for e in range(E):  # release: the particle gets its mass back where the solver left it (kinematic: where it was put)
        pick[e, :3] = ctx.get_positions(e).reshape(-1, 4)[ks[e], :3]
    pick[:, 3] = w0
    ctx.set_particles(envs, ks, pick, zero_velocity=False)
    ctx.step(C2_SETTLE)


def c2_fling_script(batch, envs, corners, timed=None):
    """The scripted fling of SURVEY 8(d) on `batch` (FlingSim, or the oracle stand-in of the checker -- both expose the
    batched movep of SimEnv.movep): both pickers to 5 cm above the cloth corners 0 and DIM - 1, down onto them, grasp, lift
    to y = 0.3 at 5e-3 per step, forward / back +-0.2 at 6e-3 per step (simEnv.py:262-275 speeds), down to 0.05, release,
    300 settle steps.  Returns the simulation steps taken, summed over the episodes."""
    n = len(envs)
    steps = 0

    def go(targets, grasp, speed):
        nonlocal steps
        batch.movep(envs, targets, np.full((n, 2), int(grasp)), speed=speed, limit=2000)
        steps += int(batch.last_movep_steps)

    c = np.asarray(corners, np.float64).reshape(n, 2, 3)
    above = c.copy(); above[:, :, 1] += 0.05
    go(above, False, 0.05)
    on = c.copy(); on[:, :, 1] += 0.01
    go(on, False, 5e-3)
    up = on.copy(); up[:, :, 1] = 0.3
    go(up, True, 5e-3)
    fwd = up.copy(); fwd[:, :, 2] += 0.2
    go(fwd, True, 6e-3)
    back = up.copy(); back[:, :, 2] -= 0.2
    go(back, True, 6e-3)
    low = back.copy(); low[:, :, 1] = 0.05
    go(low, True, 6e-3)
    go(low, False, 6e-3)          # release (the pickers are on their targets: no simulation step)
    batch.step_list(list(envs), C2_FLING_SETTLE)
    return steps + n * C2_FLING_SETTLE


class C2Check:
    """The checker of a C2 entry: episode `env` from the state the timed region started in (particles, velocities, pickers),
    the same script on the CPU oracle + the numpy restatement of the reference picker (tests/fling_helpers.OracleBatch),
    compared bit for bit with where the GPU left the episode.  Runs in a thread after every timed region."""

    def __init__(self, env, start_state, corners, gpu_end):
        import threading
        self.env, self.start, self.corners, self.gpu_end, self.out = env, start_state, corners, gpu_end, None
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import scenarios as sc
        from fling_helpers import OracleBatch
        from flingbot_amd.primitives import FlingPrimitives

        pos, vel = self.start
        ob = OracleBatch(1, sc.survey_params(DIM), pos.reshape(-1, 4), pickers=False)
        ob.sims[0].set_velocities(vel)
        FlingPrimitives(ob, [0]).place_pickers(0)
        steps = c2_fling_script(ob, [0], self.corners[None])
        po, vo = ob.sims[0].get_positions(), ob.sims[0].get_velocities()
        ph, vh = self.gpu_end
        exact = bool(np.array_equal(ph.view(np.uint32), po.view(np.uint32)) and np.array_equal(vh.view(np.uint32), vo.view(np.uint32)))
        rel = float(np.abs(ph - po).max() / max(1.0, float(np.abs(po).max())))
        self.out = {"parity_checked": True, "bit_exact": exact, "max_rel_position_error": rel, "within_1e-4": rel <= 1e-4,
                    "episode": int(self.env), "mode": f"fling + settle phases ({steps} frames) from the GPU's post-crumple state",
                    "checker": "oracle/flex_oracle.c + oracle/picker.py"}

    def start_thread(self):
        self.thread.start()

    def result(self):
        self.thread.join()
        return self.out


def c2_leg(fsim, fdist, torch, local_rank, rank, world, E, check=True):
    """One C2 / C3 entry: E episodes per GPU, crumple (untimed) then the scripted fling + settle timed with the same
    brackets as the headline.  Returns (entry or None on ranks > 0, checker or None)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from flingbot_amd.primitives import FlingPrimitives

    # A failure inside this leg (a movep that runs into its step limit, a HIP error) must neither take the headline down nor
    # leave the other ranks waiting in a collective: every rank reports its own outcome and all of them agree on "somebody
    # failed" (one max-over-ranks) BEFORE the next collective of the leg runs.
    def agreed_failure(exc):
        failed = fdist.max_over_ranks(1.0 if exc is not None else 0.0, device="cuda") > 0.0
        if failed and rank == 0:
            return {"name": f"C2 scripted fling, {E} x 64x64 episodes per GPU", "episodes_per_gpu": E,
                    "error": (f"{type(exc).__name__}: {exc}"[:300] if exc is not None else "another rank failed in this leg")}
        return {} if failed else None

    ctx, err = None, None
    try:
        ctx = fsim.FlingSim(n_envs=E, device=local_rank, solver=fsim.FS_SOLVER_AUTO)
        seeds = list(fdist.episode_range(rank, E))
        c2_crumple(ctx, seeds)
        prim = FlingPrimitives(ctx, range(E))
        for e in range(E):
            prim.place_pickers(e)
        corners = np.stack([ctx.get_positions(e).reshape(-1, 4)[[0, DIM - 1], :3] for e in range(E)]).astype(np.float64)
        ce = E // 2
        start_state = (ctx.get_positions(ce), ctx.get_velocities(ce))
        cov0 = ctx.coverage()
    except Exception as exc:  # noqa: BLE001 -- reported in the entry
        err = exc
    bad = agreed_failure(err)
    if bad is not None:
        if ctx is not None:
            ctx.close()
        return (bad or None), None
    fdist.gather_rewards(cov0, device="cuda")
    fdist.barrier(); ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps, gpu_ms, cov1 = 0, 0.0, None
    try:
        ctx.timer_start()
        steps = c2_fling_script(ctx, np.arange(E, dtype=np.int32), corners)
        gpu_ms = ctx.timer_stop()
        cov1 = ctx.coverage()
    except Exception as exc:  # noqa: BLE001
        err = exc
    bad = agreed_failure(err)
    if bad is not None:
        ctx.close()
        return (bad or None), None
    cov = fdist.gather_rewards(cov1, device="cuda")
    fdist.barrier(); ctx.sync(); torch.cuda.synchronize()
    elapsed = fdist.max_over_ranks(time.perf_counter() - t0, device="cuda")
    all_steps = float(fdist.gather_rewards([float(steps)], device="cuda").double().sum().item())   # over all ranks
    form = ctx.last_kernel_form()
    entry, chk = None, None
    if rank == 0:
        fused = form in (fsim.FS_FORM_FUSED_12, fsim.FS_FORM_FUSED_16, fsim.FS_FORM_FUSED_GENERIC, fsim.FS_FORM_FUSED_GRID64)
        entry = {"name": f"C2 scripted fling, {E} x 64x64 episodes per GPU" + (f" ({E * world} over {world} GPUs)" if world > 1 else ""),
                 "baseline_config": "SURVEY 8(d) C2 / C3: crumple recipe (tasks.py:177-224) + two-corner fling at simEnv.py:262-275 speeds",
                 "value": all_steps / elapsed, "unit": "sim steps/s", "episodes_per_gpu": E,
                 "episode_steps": int(all_steps), "seconds": elapsed, "gpu_ms": gpu_ms,
                 "phases": "pickers to the corners, grasp, lift to 0.3 at 5e-3/step, +-0.2 at 6e-3/step, lower, release, "
                           f"{C2_FLING_SETTLE} settle steps; the crumple ({C2_RAISE} + {C2_HOLD} + {C2_SETTLE} steps) is untimed set-up",
                 "calls": "one fs_movep_batch per leg (picker kernel + solver per step, planned on the host) + one fs_step_list",
                 "solver": ("fused (AUTO)" if fused else "stream (AUTO)"), "kernel_form": int(form),
                 # SURVEY 8(d)'s equivalent bandwidth for this entry: algorithmic bytes of the steps this GPU took / its
                 # HIP-event time over the whole script (picker kernels included)
                 "roofline_frac_equivalent": BYTES_PER_STEP * float(steps) / max(gpu_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                 "mean_coverage": float(cov.mean().item())}
        if check:
            chk = C2Check(ce, start_state, corners[ce], (ctx.get_positions(ce), ctx.get_velocities(ce)))
    ctx.close()
    return entry, chk


def run_rank(args):
    import torch

    from flingbot_amd import distributed as fdist
    from flingbot_amd import sim as fsim

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, local_rank, world = fdist.init_from_env("nccl")  # one process per GPU; "nccl" is RCCL on ROCm

    E = args.episodes
    ctx = fsim.FlingSim(n_envs=E, device=local_rank, solver=args.solver)
    # what the collective saw, not what WORLD_SIZE claims: one all_gather of (rank, LOCAL_RANK, device identity, architecture).
    # Every rank gets the same table, so every rank takes the same decision: two ranks on one GPU => no figure at all.
    try:
        arch = torch.cuda.get_device_properties(local_rank).gcnArchName
    except Exception:
        arch = ""
    census = fdist.rank_census(ctx.device_key(), arch)
    if census["ranks_seen"] != world or census["distinct_devices"] != world:
        ctx.close()
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but the process group's all_gather saw {census['ranks_seen']} rank(s) on "
                         f"{census['distinct_devices']} distinct device(s) ({census['backend']}): one process per GPU is the "
                         f"contract, refusing to report a {world}-GPU figure")
    for e, g in enumerate(fdist.episode_range(rank, E)):
        setup_episode(ctx.env(e), seed=g)  # global episode id = seed
    ctx.sync()
    elapsed, kern_ms_total, cov_all = timed_batch(ctx, fdist, torch, args.steps, args.warmup, args.preroll)
    form = ctx.last_kernel_form()

    out, parity_main, parity_2 = None, None, None
    if rank == 0:
        total_steps = E * world * args.steps
        value = total_steps / elapsed
        kern_ms = kern_ms_total / args.steps
        achieved = BYTES_PER_STEP * E / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = traffic_from_profile(E)
        fused = form in (fsim.FS_FORM_FUSED_12, fsim.FS_FORM_FUSED_16, fsim.FS_FORM_FUSED_GENERIC, fsim.FS_FORM_FUSED_GRID64)
        out = {
            "metric": "sim steps/sec (64x64-particle cloth)",
            "value": value,
            "unit": "sim steps/s",
            "n_gpus": world,
            "ranks_seen": census["ranks_seen"], "distinct_devices": census["distinct_devices"], "backend": census["backend"],
            "collective_library": census["collective_library"],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{E} x 64x64-particle cloth episodes per GPU (4096 particles, 23938 springs, "
                                   f"self-collision + ground friction; vertical sheet released, timed from frame "
                                   f"{args.preroll + args.warmup} on: crumpled on the ground), one pyflex.step() (4 substeps x 30 "
                                   f"iterations) of every episode per bench step",
                       "episodes_per_gpu": E, "cloth": "64x64", "substeps": SUBSTEPS, "iterations": ITERS,
                       "preroll_frames": args.preroll,
                       "solver": "fused-lds" if fused else "stream", "parallelism": f"episodes x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "basis": "equivalent streamed bandwidth: ALGORITHMIC bytes of SURVEY 8(d)'s streaming model per "
                                  "launch / HIP-event kernel time; the fused kernel keeps the iterations in LDS, so its "
                                  "HBM traffic (`traffic`) is far below this and `limiter` names what bounds it",
                         "kernel": ("fs_k_fused_grid64" if form == fsim.FS_FORM_FUSED_GRID64 else "fs_k_fused_step") if fused
                                   else "fs_k_iterate (+stage kernels)",
                         "kernel_ms_per_launch": kern_ms, "algorithmic_bytes_per_launch": BYTES_PER_STEP * E},
            "mean_coverage": float(cov_all.mean().item()),
        }
        # `roofline` above is the contract's figure (algorithmic bytes of the streaming model / kernel time).  The
        # LDS-resident kernel does not move those bytes, so the figure saturates: a faster kernel prints a fraction above 1 of
        # a bandwidth nobody uses.  What physically bounds it is VALU issue: `valu_roofline` is the number to read, and it leads.
        out["roofline"]["saturated"] = bool(out["roofline"]["frac"] > 1.0)
        if out["roofline"]["saturated"]:
            out["roofline"]["note"] = "metric saturated: equivalent bandwidth above the HBM peak -- see valu_roofline"
        if fused:
            limiter = limiter_from_profile()
            vr = valu_roofline(limiter, E, kern_ms)
            if vr:   # lead with the physical bound: re-insert it right in front of the contract's `roofline`
                head = {k: out.pop(k) for k in list(out) if k not in ("metric", "value", "unit", "n_gpus", "ranks_seen", "distinct_devices", "backend",
                                                                      "collective_library", "steps", "warmup",
                                                                      "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                                                      "dtype", "data", "config")}
                out["valu_roofline"] = vr
                out["valu_frac_of_peak"] = vr["frac"]
                out["roofline_that_bounds_the_kernel"] = "valu_roofline"
                out.update(head)
            out["limiter"] = limiter
        if not args.no_parity:
            parity_main = ParityCheck(ctx, E // 2, E // 2, args.preroll + args.warmup + args.steps)
    ctx.close()

    # ---- secondary figure, every rank, after the headline's timed region: 64 episodes per GPU (BASELINE.json configs[2];
    #      at 8 GPUs this is configs[3]: 512 episodes sharded 8 x 64 with the coverage gather over RCCL)
    if not args.no_secondary:
        E2, K2, W2, NW = 64, 100, max(3, min(args.warmup, 10)), 3
        ctx2 = fsim.FlingSim(n_envs=E2, device=local_rank, solver=fsim.FS_SOLVER_AUTO)
        for e, g in enumerate(fdist.episode_range(rank, E2)):
            setup_episode(ctx2.env(e), seed=g)
        ctx2.sync()
        wins, cov2 = timed_windows(ctx2, fdist, torch, K2, W2, args.preroll, NW)
        form2 = ctx2.last_kernel_form()
        if rank == 0:
            rates = sorted(E2 * world * K2 / el for el, _ in wins)
            el2, k2_ms = sorted(wins)[len(wins) // 2]          # the median window
            rate2 = E2 * world * K2 / el2
            entry = {"name": "64 x 64x64 episodes per GPU" + (f", {E2 * world} episodes over {world} GPUs" if world > 1 else ""),
                     "baseline_config": "configs[2]" + (" / configs[3] at 8 GPUs" if world == 8 else ""),
                     "value": rate2, "unit": "sim steps/s", "episodes_per_gpu": E2, "steps": K2, "warmup": W2,
                     "windows": NW, "value_min": rates[0], "value_max": rates[-1],
                     "timing": f"median of {NW} windows of {K2} frames, every window from the same state (frame "
                               f"{args.preroll + W2}); no checker or other host work runs during them",
                     "ms_per_step": el2 / K2 * 1e3, "gpu_ms_per_step": k2_ms / K2,
                     "solver": "stream (AUTO)" if form2 in (fsim.FS_FORM_STREAM_EAGER, fsim.FS_FORM_STREAM_CODED, fsim.FS_FORM_STREAM_ELL,
                                                            fsim.FS_FORM_STREAM_GRID, fsim.FS_FORM_STREAM_GRIDL, fsim.FS_FORM_STREAM_GRIDL_TP) else "fused (AUTO)",
                     "kernel_form": int(form2), "concurrent_launch_chains": int(ctx2.last_stream_groups()),
                     "calls": "one fs_step call per frame (as pyflex.step() is called); frames batched into one call run ~5 % faster",
                     "roofline_frac_equivalent": BYTES_PER_STEP * E2 / (k2_ms / K2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "limiter": stream_limiter_from_profile(),
                     "mean_coverage": float(cov2.mean().item())}
            if not args.no_parity:
                parity_2 = ParityCheck(ctx2, E2 // 2, E2 // 2, args.preroll + W2 + K2)
            out["configs"] = [
                {"name": f"{E} x 64x64 episodes per GPU (headline)", "value": out["value"], "unit": "sim steps/s",
                 "episodes_per_gpu": E, "ms_per_step": out["ms_per_step"], "solver": out["config"]["solver"]},
                entry]
        ctx2.close()

    # ---- SURVEY 8(d) C2 / C3 as timed entries: crumple + scripted fling with both pickers, at the headline's batch and at 64
    c2_checks = []
    if not args.no_secondary and not args.no_c2:
        for Ec in (E, 64):
            entry, chk = c2_leg(fsim, fdist, torch, local_rank, rank, world, Ec, check=not args.no_parity)
            if rank == 0 and entry is not None:
                entry["key"] = f"c2_fling_{Ec}"
                if "error" not in entry:
                    sheet = next(c for c in out["configs"] if c.get("episodes_per_gpu") == Ec and "C2" not in c["name"])
                    entry["ratio_to_crumpled_sheet"] = entry["value"] / sheet["value"]
                out["configs"].append(entry)
                c2_checks.append((entry, chk))
        if rank == 0:
            head = next((c for c in out["configs"] if c.get("key") == f"c2_fling_{E}"), None)
            if head is not None and "ratio_to_crumpled_sheet" in head:
                out["fling_phase_ratio"] = head["ratio_to_crumpled_sheet"]

    if rank == 0:
        # every timed region of the solver is over: now the checkers run (one host core each, side by side), then the
        # CPU baseline (all cores, nothing else running), then the perception and evaluation-loop legs
        for chk in (parity_main, parity_2):
            if chk is not None:
                chk.start()
        for _, chk in c2_checks:
            if chk is not None:
                chk.start_thread()
        if parity_main is not None:
            out["parity"] = parity_main.result()
            out["parity_checked"] = bool(out["parity"]["bit_exact"])
        if parity_2 is not None:
            out["configs"][1]["parity"] = parity_2.result()
        for entry, chk in c2_checks:
            if chk is not None:
                entry["parity"] = chk.result()
        if world > 1 and not args.no_parity:
            out["two_devices_one_process"] = two_device_check(fsim, torch, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.preroll + args.warmup)
        if getattr(args, "dropin", None) is not None:
            out["dropin"] = args.dropin
        if world == 1 and not args.no_secondary:
            out["perception"] = perception_leg(torch.device("cuda", local_rank))
            if not args.no_eval_loop:
                out["eval_loop"] = eval_loop_leg(local_rank)
        print(json.dumps(out), flush=True)
    fdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


def two_device_check(fsim, torch, local_rank):
    """Multi-GPU runs only (rank 0, after every timed region): ONE process driving two devices -- its own and its
    neighbour's -- with the fused kernel, whose dynamic-LDS attribute belongs to each DEVICE's copy of the kernel; both
    contexts against the oracle, bit for bit (the body of tests/test_parity_gpu.py::test_two_contexts_on_two_devices, which a
    one-GPU test box has to skip).  Failure-safe: the entry carries `error`, the headline prints as usual."""
    try:
        n_dev = torch.cuda.device_count()
        if n_dev < 2:
            return {"checked": False, "reason": f"{n_dev} device(s) visible to rank 0"}
        from oracle import OracleSim
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import cloth_params

        p = cloth_params(64, 64, pos=(0.0, -0.05, 0.0))
        orc = OracleSim()
        orc.set_scene(p)
        orc.step(10)
        po, vo = orc.get_positions().view(np.uint32), orc.get_velocities().view(np.uint32)
        devices, exact = [local_rank, (local_rank + 1) % n_dev], []
        ctxs = [fsim.FlingSim(n_envs=1, device=d, solver=fsim.FS_SOLVER_FUSED) for d in devices]
        for ctx in ctxs:
            ctx.set_scene(0, p)
            ctx.step(10)
            exact.append(bool(np.array_equal(np.asarray(ctx.get_positions(0)).view(np.uint32), po) and
                              np.array_equal(np.asarray(ctx.get_velocities(0)).view(np.uint32), vo)))
        keys = [ctx.device_key() for ctx in ctxs]
        for ctx in ctxs:
            ctx.close()
        return {"checked": True, "devices": devices, "device_keys": keys, "distinct": len(set(keys)) == 2, "bit_exact": all(exact),
                "workload": "64 x 64 cloth, 10 frames, fused kernel on both devices from one process, vs oracle/flex_oracle.c"}
    except Exception as exc:
        return {"checked": False, "error": str(exc)[:300]}


def eval_loop_leg(device_index, episodes=32, actions=3, stream_tasks=384, stream_slots=192):
    """Secondary, after the timed region (rank 0, N = 1): BASELINE.json configs[4] at the reference's own sizes -- the
    run_sim.py evaluation loop on generated 'hard' tasks with cloth sides 64..103 (environment/tasks.py:105-275), 720 x 720
    render -> 400 x 400 observation with adaptive scaling, 12 rotations x 8 scales, seeded random-init fling policy
    (flingbot.pth is not in this image), up to `actions` actions per episode.  The loop is flingbot_amd.evaluate.run_tasks:
    like the reference's (utils.step_env's ray.wait, SimEnv pulling its next task) every slot steps on its own and refills
    itself, and the device is kept busy while the host serves requests (schedule.run_programs_pipelined).  Two figures: `episodes` tasks on as many slots (the configuration of the earlier rounds' figure), and
    `continuous`: stream_tasks tasks through stream_slots slots (throughput of a long evaluation run; 192 slots since the end of
    round 3 -- about a third of the slots is in a host-side stage at any time, and the streaming kernels only fill the chip from
    ~130 active episodes on: 96 slots give 28 flings/s, 192 give 32, 256 give 32.6).  Reports flings/s and
    simulated episode-steps/s of the whole loop (perception + action selection + primitives + resets)."""
    try:
        import random
        import torch
        from flingbot_amd import nets, sim as fsim, tasks as ftasks
        from flingbot_amd.env import BatchedFlingEnv
        from flingbot_amd.evaluate import run_tasks

        def one(n_tasks, n_slots, seed):
            random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
            params, tasks = [], []
            t0 = time.perf_counter()
            for k in range(0, n_tasks, n_slots):
                part = [ftasks.draw_task_parameters() for _ in range(min(n_slots, n_tasks - k))]
                gen = fsim.FlingSim(n_envs=len(part), device=device_index, solver=0)
                made = ftasks.generate_tasks(gen, part)
                gen.close()
                # the generator rejects a task whose cloth did not come down ("probably an error", tasks.py:226-229 -> None);
                # like the reference's generation loop, such a task is simply not part of the set
                tasks += [t for t in made if t is not None]
                params += [p_ for p_, t in zip(part, made) if t is not None]
            t_gen = time.perf_counter() - t0
            ctx = fsim.FlingSim(n_envs=n_slots, device=device_index, solver=0)
            env = BatchedFlingEnv(ctx, episode_length=actions, device=f"cuda:{device_index}")
            policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                             obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                             depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                             value_expl_decay=1.0, device=f"cuda:{device_index}")
            t0 = time.perf_counter()
            stats = run_tasks(policy, env, tasks)
            dt = time.perf_counter() - t0
            flings = int(sum(stats["action_primitive_counts"].values()))
            sides = np.array([p["cloth_size"] for p in params])
            sched = stats["scheduler"]
            ctx.close()
            n_tasks = len(tasks)
            mean_bytes = float(np.mean([algorithmic_bytes_per_step(int(a_), int(b_)) for a_, b_ in sides])) if len(sides) else 0.0
            return {"episodes": n_tasks, "slots": n_slots,
                    # SURVEY 8(d)'s equivalent bandwidth for configs[4]: the episode-steps of the loop x the mean algorithmic bytes of
                    # a step over the task set's cloth sizes (every task weighted equally; steps per task are not recorded) / the
                    # WALL time of the whole loop -- perception, action selection, resets and host scheduling included
                    "roofline_frac_equivalent": stats["simulation_steps"] * mean_bytes / dt / 1e9 / HBM_PEAK_GBS,
                    "mean_algorithmic_bytes_per_step": mean_bytes, "max_actions": actions, "cloth_sides": [int(sides.min()), int(sides.max())],
                    "transforms": len(env.transformations), "render_dim": env.render_dim, "image_dim": env.image_dim,
                    "seconds": dt, "task_generation_seconds": t_gen, "flings": flings, "flings_per_s": flings / dt,
                    "episode_steps": int(stats["simulation_steps"]), "episode_steps_per_s": stats["simulation_steps"] / dt,
                    "launch_sequences": int(sched.get("sequences", 0)),
                    "mean_active_episodes": sched.get("episode_steps", 0) / max(sched.get("sequences", 1), 1),
                    "mean_init_coverage": stats["mean"]["init_coverage"], "mean_final_coverage": stats["mean"]["final_coverage"]}

        out = one(episodes, episodes, 0)
        out.update({"baseline_config": "configs[4]",
                    "loop": "evaluate.run_tasks (asynchronous slots like run_sim.py / utils.step_env; chunks of simulation queued "
                            "ahead with fs_advance_begin / fs_advance_end, host-side services on the service lane, scenes "
                            "prebuilt on a worker thread)",
                    "policy": "random-init fling value net (seeded)"})
        if stream_tasks > 0:
            out["continuous"] = one(stream_tasks, stream_slots, 1)
            lim = eval_limiter_from_profile()
            if lim:
                out["continuous"]["limiter"] = lim
        return out
    except Exception as exc:  # the headline number must not depend on this leg
        import traceback
        return {"error": str(exc)[:300], "where": traceback.format_exc().strip().splitlines()[-6:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--episodes", type=int, default=256, help="cloth episodes per GPU")
    ap.add_argument("--preroll", type=int, default=60,
                    help="untimed frames before the warm-up: the timed window then sees the crumpled sheet whatever --steps is")
    ap.add_argument("--solver", type=int, default=2, help="2 fused LDS kernel (default), 1 streaming kernels, 0 AUTO")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed batch")
    ap.add_argument("--no-secondary", action="store_true", help="headline only: no 64-episode figure, no perception leg")
    ap.add_argument("--no-eval-loop", action="store_true", help="skip the evaluation-loop leg (configs[4], ~50 s)")
    ap.add_argument("--no-c2", action="store_true", help="skip the C2 / C3 crumple + scripted-fling entries")
    ap.add_argument("--no-dropin", action="store_true", help="skip the pyflex-module drop-in leg (1 and 16 processes)")
    ap.add_argument("--timeout", type=float, default=1800.0,
                    help="seconds after which self-launched ranks (--gpus N without WORLD_SIZE) are stopped: a hung rendezvous ends")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # nobody started the ranks for us: start them ourselves, one fresh interpreter per GPU, BEFORE this process has
        # made any GPU call (it never makes one: flingbot_amd.launch is standard library only)
        from flingbot_amd.launch import launch_local_ranks

        sys.exit(launch_local_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:], timeout=args.timeout))
    if world_env is None and args.gpus == 1 and not args.no_secondary and not args.no_dropin:
        # the drop-in leg runs FIRST, from this process while it has not touched the GPU: its workers are fresh interpreters
        # (one pyflex.init each, like the reference's Ray workers) and must not be children of a process that holds the device
        try:
            import bench_dropin
            args.dropin = bench_dropin.measure()
        except Exception as exc:
            args.dropin = {"error": str(exc)[:300]}
    if int(world_env or "1") != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: launch with torch.distributed.run, or leave "
                         f"WORLD_SIZE unset and bench.py starts the ranks itself")
    run_rank(args)


if __name__ == "__main__":
    main()
