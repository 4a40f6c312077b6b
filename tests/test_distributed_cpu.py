"""N > 1 path on CPU: two `gloo` ranks shard episodes, gather per-episode coverage rewards and reduce the timing, using
the same helpers bench.py runs over RCCL on the GPU node (flingbot_amd/distributed.py)."""
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
import numpy as np
import pytest
sys.path.insert(0, os.environ["FS_ROOT"])
import torch
from flingbot_amd import distributed as fdist
from oracle.coverage import covered_area

rank, local_rank, world = fdist.init_from_env("gloo")
E = 3
episodes = list(fdist.episode_range(rank, E))
cov = []
for g in episodes:                      # each rank evaluates only its own episodes
    rng = np.random.RandomState(g)
    pos = np.concatenate([rng.rand(200, 3) * [0.4, 0.1, 0.3], np.ones((200, 1))], 1).astype(np.float32)
    cov.append(covered_area(pos))
allcov = fdist.gather_rewards(cov)
tmax = fdist.max_over_ranks(1.0 + rank)
fdist.barrier()
print(json.dumps({"rank": rank, "world": world, "episodes": episodes, "cov": allcov.tolist(), "tmax": tmax}))
torch.distributed.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gloo_shard_and_gather(tmp_path):
    import json

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FS_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["episodes"] == [0, 1, 2] and outs[1]["episodes"] == [3, 4, 5]  # disjoint, complete partition
    from oracle.coverage import covered_area

    expect = []
    for g in range(6):
        rng = np.random.RandomState(g)
        pos = np.concatenate([rng.rand(200, 3) * [0.4, 0.1, 0.3], np.ones((200, 1))], 1).astype(np.float32)
        expect.append(np.float32(covered_area(pos)))
    for o in outs:  # every rank ends with the full vector, ordered by global episode id
        assert o["world"] == 2
        assert np.array_equal(np.array(o["cov"], np.float32), np.array(expect, np.float32))
        assert o["tmax"] == 2.0


def test_single_rank_helpers_need_no_process_group():
    from flingbot_amd import distributed as fdist

    assert list(fdist.episode_range(3, 4)) == [12, 13, 14, 15]
    assert fdist.gather_rewards([1.0, 2.0]).tolist() == [1.0, 2.0]
    assert fdist.max_over_ranks(0.25) == 0.25
    fdist.barrier()


SHARDED_WORKER = r"""
import os, sys, json
import numpy as np
import pytest
sys.path.insert(0, os.environ["FS_ROOT"])
import torch
from flingbot_amd import distributed as fdist
from flingbot_amd.evaluate import run_episodes_sharded

def fake_runner(policy, env, tasks):           # stands in for the GPU loop: coverage derived from the task itself
    init = np.array([t["seed"] * 0.01 for t in tasks])
    return {"init_coverage": init, "final_coverage": init + 0.5, "n": len(tasks)}

W = int(os.environ["WORLD_SIZE"])
tasks = [{"seed": g} for g in range(4 * W)]
stats = run_episodes_sharded(None, None, tasks, episodes_per_rank=4, runner=fake_runner)
print(json.dumps({"rank": stats["rank"], "world": stats["world"], "n": stats["n"], "mine": stats["init_coverage"].tolist(),
                  "all_init": stats["all_init_coverage"].tolist(), "all_final": stats["all_final_coverage"].tolist()}))
fdist.barrier()
torch.distributed.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_evaluation_gathers_by_global_episode(tmp_path, world):
    """evaluate.run_episodes_sharded on two and on EIGHT gloo ranks (the node's shape): each rank runs only its own tasks,
    every rank ends with the coverages of all episodes in global order."""
    import json

    script = tmp_path / "sharded.py"
    script.write_text(SHARDED_WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FS_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    for r in range(world):
        assert np.allclose(outs[r]["mine"], (np.arange(4) + 4 * r) * 0.01)
    for o in outs:
        assert o["world"] == world and o["n"] == 4
        assert np.allclose(o["all_init"], np.arange(4 * world) * 0.01) and np.allclose(o["all_final"], np.arange(4 * world) * 0.01 + 0.5)


MERGE_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["FS_ROOT"])
import torch
from flingbot_amd import distributed as fdist
from flingbot_amd.evaluate import merge_rank_statistics

rank, _, world = fdist.init_from_env("gloo")
n_tasks, per_rank = 5, 3                                   # 5 tasks over 2 ranks: blocks of 3 and 2
mine = list(range(n_tasks))[rank * per_rank:(rank + 1) * per_rank]
stats = {"init_coverage": np.array([0.1 * (t + 1) for t in mine]), "final_coverage": np.array([0.1 * (t + 1) + 0.05 * t for t in mine]),
         "simulation_steps": 100 * (rank + 1)}
if os.environ.get("FS_NAN_EPISODE") and rank == 1:
    stats["final_coverage"][0] = np.nan                     # a diverged episode: its coverage really is NaN
out = merge_rank_statistics(stats, per_rank)
print(json.dumps({"rank": rank, **out}))
fdist.barrier()
torch.distributed.destroy_process_group()
"""


def test_evaluate_command_merges_uneven_rank_blocks(tmp_path):
    """`python -m flingbot_amd.evaluate --gpus N`: the task set is cut into contiguous blocks, the last one shorter; every
    rank ends with the statistics over ALL episodes (two gloo ranks, 5 tasks as 3 + 2)."""
    import json

    script = tmp_path / "merge.py"
    script.write_text(MERGE_WORKER)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port), FS_ROOT=ROOT)) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        rec = json.loads(out.strip().splitlines()[-1])
        init = np.array([0.1 * (t + 1) for t in range(5)], np.float32)
        final = np.array([0.1 * (t + 1) + 0.05 * t for t in range(5)], np.float32)
        assert rec["gpus"] == 2 and rec["episodes"] == 5 and rec["simulation_steps"] == 300
        assert rec["init_coverage"] == pytest.approx(float(init.mean())) and rec["final_coverage"] == pytest.approx(float(final.mean()))
        assert rec["episode_delta_coverage"] == pytest.approx(float((final - init).mean())) and "non_finite_coverages" not in rec
    # an episode whose coverage really is NaN is COUNTED and SHOWN (the padding behind a short block is sliced off by the gathered
    # episode counts, not recognised by its value): 5 episodes, NaN means, one non-finite coverage reported
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port), FS_ROOT=ROOT, FS_NAN_EPISODE="1")) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        rec = json.loads(out.strip().splitlines()[-1])
        assert rec["episodes"] == 5 and rec["non_finite_coverages"] == 1 and np.isnan(rec["final_coverage"])
        assert rec["init_coverage"] == pytest.approx(float(init.mean()))


LAUNCHED_WORKER = r"""
import os, sys, json
sys.path.insert(0, os.environ["FS_ROOT"])
import torch
from flingbot_amd import distributed as fdist

rank, local_rank, world = fdist.init_from_env("gloo")
t = torch.tensor([float(rank + 1)])
torch.distributed.all_reduce(t)
with open(os.path.join(sys.argv[1], f"rank{rank}.json"), "w") as fh:
    json.dump({"rank": rank, "local_rank": local_rank, "world": world, "sum": t.item(), "argv": sys.argv[2:]}, fh)
fdist.barrier()
torch.distributed.destroy_process_group()
if "--fail-rank-1" in sys.argv and rank == 1:
    sys.exit(7)
if "--kill-rank-1" in sys.argv and rank == 1:
    import signal
    os.kill(os.getpid(), signal.SIGKILL)
"""


def test_launcher_starts_ranks_that_rendezvous(tmp_path):
    """flingbot_amd.launch.launch_local_ranks (what `python bench.py --gpus N` uses when nobody set WORLD_SIZE): N fresh
    interpreters with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set find each other (gloo here, RCCL on the GPU node);
    a failing rank's exit code comes back."""
    import json
    from flingbot_amd.launch import launch_local_ranks

    script = tmp_path / "launched.py"
    script.write_text(LAUNCHED_WORKER)
    env = dict(os.environ, FS_ROOT=ROOT)
    env.pop("WORLD_SIZE", None)
    assert launch_local_ranks(2, str(script), [str(tmp_path), "--steps", "3"], env=env, timeout=180) == 0
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r, rec in enumerate(recs):
        assert rec == {"rank": r, "local_rank": r, "world": 2, "sum": 3.0, "argv": ["--steps", "3"]}
    assert launch_local_ranks(2, str(script), [str(tmp_path), "--fail-rank-1"], env=env, timeout=180) == 7
    # a rank killed by a signal comes back as 128 + signal (not as a negative number handed to sys.exit), and the caller's
    # MASTER_PORT is the one the ranks meet on
    port = _free_port()
    assert launch_local_ranks(2, str(script), [str(tmp_path), "--kill-rank-1"], env=dict(env, MASTER_PORT=str(port)),
                              timeout=180) == 128 + 9
    # the node's shape: eight ranks, LOCAL_RANK = RANK = the device each one binds, a port found by the launcher
    assert launch_local_ranks(8, str(script), [str(tmp_path)], env=env, timeout=300) == 0
    for r in range(8):
        assert json.load(open(tmp_path / f"rank{r}.json")) == {"rank": r, "local_rank": r, "world": 8, "sum": 36.0, "argv": []}


def test_launcher_port_probe_ignores_time_wait_but_sees_a_listener():
    """launch._port_in_use decides whether a failed launch is repeated on another port: a LISTENING socket counts, the
    TIME_WAIT leftovers of our own dead rank 0 do not (a script failure must not be relaunched three times)."""
    from flingbot_amd.launch import _port_in_use, free_port

    port = free_port()
    assert not _port_in_use("127.0.0.1", port)
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", port))
    srv.listen(1)
    assert _port_in_use("127.0.0.1", port)
    cli = socket.socket()
    cli.connect(("127.0.0.1", port))
    conn, _ = srv.accept()
    conn.close()            # the server side closes first: its end of the connection goes to TIME_WAIT on `port`
    cli.close()
    srv.close()
    assert not _port_in_use("127.0.0.1", port)


def test_launcher_module_never_touches_the_gpu_runtime():
    """The parent of the ranks may not initialise HIP: the launcher imports neither torch nor libflingsim."""
    code = "import sys; sys.path.insert(0, %r); import flingbot_amd.launch; " \
           "bad = [m for m in sys.modules if m.split('.')[0] in ('torch', 'numpy', 'ctypes')]; " \
           "assert not [m for m in bad if m.startswith('torch')], bad; print('ok')" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_bench_starts_its_own_ranks_without_world_size():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts two ranks (which, on this GPU-less machine, each
    stop with the no-CPU-fallback message) instead of refusing to run."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the launcher's happy path is exercised by bench.py itself")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    # (the launcher terminates the other ranks as soon as one fails, so the second rank's message may or may not get out)
    assert 1 <= out.stderr.count("no HIP device visible") <= 2, out.stderr[-2000:]
    assert "launch with torch.distributed.run" not in out.stderr


BENCH_WORKER = r"""
# One rank of `bench.py --gpus 2` on a machine without a GPU: bench.run_rank's OWN control flow (barriers, timed windows,
# gathers, rank-0-only JSON) with the solver context replaced at the fsim.FlingSim seam by a recording stub, torch.cuda's
# three calls made no-ops and the process group on gloo.  Nothing in bench.py knows about this.
import json, os, sys, time, types
import numpy as np
import pytest
sys.path.insert(0, os.environ["FS_ROOT"])
import torch
from flingbot_amd import distributed as fdist
from flingbot_amd import sim as fsim

RANK = int(os.environ["RANK"])
LOG = []                                             # (event, detail) in program order of THIS rank

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: LOG.append(("set_device", int(d)))
torch.cuda.synchronize = lambda *a, **k: LOG.append(("cuda_sync", None))
_init, _gather, _max, _barrier = fdist.init_from_env, fdist.gather_rewards, fdist.max_over_ranks, fdist.barrier
def init_from_env(backend=None):
    LOG.append(("init", backend))
    return _init("gloo")
def gather_rewards(r, device=None):
    LOG.append(("gather", device)); return _gather(r, None)
def max_over_ranks(v, device=None):
    LOG.append(("max", device)); return _max(v, None)
def barrier():
    LOG.append(("barrier", None)); return _barrier()
fdist.init_from_env, fdist.gather_rewards, fdist.max_over_ranks, fdist.barrier = init_from_env, gather_rewards, max_over_ranks, barrier


class StubSim:
    instances = []
    def __init__(self, n_envs=1, device=0, solver=0, **kw):
        self.n_envs, self.device, self.solver = n_envs, device, solver
        self.pos = [None] * n_envs
        self.first_pos = [None] * n_envs
        self.steps = 0
        StubSim.instances.append(self)
        LOG.append(("create", (n_envs, device, solver)))
    def env(self, e):
        sim = self
        class View:
            def set_scene(self, p): pass
            def step(self, n=1): pass
            def get_positions(self): return np.full(4 * 4096, 8192.0, np.float32)
            def set_positions(self, p):
                sim.pos[e] = np.array(p, np.float32)
                if sim.first_pos[e] is None: sim.first_pos[e] = sim.pos[e].copy()
            def set_velocities(self, v): pass
        return View()
    def sync(self): LOG.append(("ctx_sync", None))
    def device_key(self):   # the device's PCI bus id; FS_SAME_DEVICE: two ranks were (wrongly) started on one GPU
        return "0000:%02x:00.0" % (5 if os.environ.get("FS_SAME_DEVICE") and RANK < 2 else 5 + RANK)
    def step(self, n=1):
        self.steps += n; LOG.append(("step", n)); time.sleep(0.001 * (1 + 2 * (RANK == 1)))      # rank 1 is the slow one
    def timer_start(self): LOG.append(("timer_start", None)); self._t = time.perf_counter()
    def timer_stop(self): LOG.append(("timer_stop", None)); return (time.perf_counter() - self._t) * 1e3
    def coverage(self): return np.arange(self.n_envs, dtype=np.float64) + 1000.0 * RANK
    def last_kernel_form(self): return fsim.FS_FORM_FUSED_GRID64 if self.n_envs != 64 else fsim.FS_FORM_STREAM_GRIDL
    def last_stream_groups(self): return 2
    def get_positions(self, e=0): return self.pos[e]
    def get_velocities(self, e=0): return np.zeros(3 * 4096, np.float32)
    def set_positions(self, e, p): self.pos[e] = np.array(p, np.float32)
    def set_velocities(self, e, v): pass
    def close(self): LOG.append(("close", self.n_envs))
    # what the C2 entries touch beyond the above (bench.c2_crumple / c2_fling_script / FlingPrimitives.place_pickers)
    def set_particles(self, envs, pids, pos4, zero_velocity=True): pass
    def step_list(self, envs, n=1): self.steps += n; LOG.append(("step_list", n))
    def movep(self, envs, targets, grasp, speed=0.1, limit=1000, min_steps=None, eps=1e-4):
        if os.environ.get("FS_C2_FAIL_RANK") == str(RANK) and self.n_envs == 4:
            raise RuntimeError("movep: step limit reached (injected)")       # one rank's fling script fails in the first C2 leg
        self.last_movep_steps = 3 * len(envs); LOG.append(("movep", len(envs))); return np.full(len(envs), 3, np.int32)
    def add_sphere(self, e, radius, pos, quat): self.shapes = getattr(self, "shapes", {}); self.shapes.setdefault(e, []).append(list(pos))
    def get_shape_states(self, e): return np.zeros(14 * len(self.shapes.get(e, [])), np.float32)
    def set_shape_states(self, e, s): pass
    def picker_reset(self, e, *a, **k): pass

fsim.FlingSim = StubSim
import bench
args = types.SimpleNamespace(episodes=4, steps=3, warmup=1, preroll=2, solver=2, no_parity=True, no_cpu_baseline=True,
                             no_secondary=False, no_eval_loop=True, no_c2=os.environ["WORLD_SIZE"] != "2", no_dropin=True,
                             gpus=int(os.environ["WORLD_SIZE"]))
bench.run_rank(args)
seeds_ok = all(np.array_equal(s.first_pos[e], bench.initial_state(RANK * s.n_envs + e, 8192.0).ravel())
               for s in StubSim.instances[:2] for e in range(s.n_envs))
if os.environ.get("FS_C2_FAIL_RANK"):
    seeds_ok = True
with open(os.path.join(os.environ["FS_OUT"], f"rank{RANK}.json"), "w") as fh:
    json.dump({"log": LOG, "seeds_ok": bool(seeds_ok), "sizes": [s.n_envs for s in StubSim.instances]}, fh)
"""


@pytest.mark.parametrize("world", [2, 8])
def test_bench_run_rank_control_flow_two_ranks(tmp_path, world):
    """8-GPU readiness without the hardware: `bench.run_rank` itself runs as two -- and as EIGHT -- gloo ranks with the solver context stubbed
    at the fsim.FlingSim seam (inside this test only).  Checked: one JSON line, from rank 0 only; n_gpus / weak scaling /
    whole-job value from the MAX over ranks; every rank sets its own global episodes up (rank r: seeds r E .. r E + E - 1);
    the headline's timed region is barrier + synchronize -> timer -> exactly K steps -> timer -> coverage gather ->
    barrier + synchronize -> max-over-ranks; the secondary entry is 64 episodes per rank = 128 over 2 GPUs (configs[3]'s
    shape: 512 at 8), three windows of 100 frames, on every rank."""
    import json
    sys.path.insert(0, ROOT)
    import bench as _bench
    bench_fling_settle = _bench.C2_FLING_SETTLE

    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FS_ROOT=ROOT, FS_OUT=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        outs.append(out)
    lines0 = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not [l for o in outs[1:] for l in o.splitlines() if l.startswith("{")]   # rank 0 prints, the others are silent
    rec = json.loads(lines0[0])
    assert rec["n_gpus"] == world and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    # ... and what the process group's own all_gather saw (not WORLD_SIZE): as many ranks, on as many distinct devices
    assert rec["ranks_seen"] == world and rec["distinct_devices"] == world and rec["backend"] == "gloo"
    assert rec["config"]["episodes_per_gpu"] == 4 and rec["config"]["parallelism"] == f"episodes x{world}"
    assert rec["unit"] == "sim steps/s" and rec["higher_is_better"] is True and rec["vs_baseline"] is None
    # whole-job value = all ranks' episode-steps / the SLOWEST rank's time: rank 1 sleeps 3 ms per step, rank 0 1 ms
    assert rec["value"] == pytest.approx(4 * world * 3 / (rec["ms_per_step"] * 3e-3))
    assert rec["ms_per_step"] >= 3.0
    assert rec["roofline"]["bound"] == "hbm" and rec["roofline"]["kernel"] == "fs_k_fused_grid64"
    assert rec["mean_coverage"] == pytest.approx(np.mean([1000.0 * r + e for r in range(world) for e in range(4)]))   # every rank's rewards, gathered
    assert "cpu_baseline" not in rec and "eval_loop" not in rec                                 # N = 1 only
    sec = rec["configs"][1]
    assert sec["episodes_per_gpu"] == 64 and f"{64 * world} episodes over {world} GPUs" in sec["name"] and sec["windows"] == 3
    assert ("configs[3]" in sec["baseline_config"]) == (world == 8)          # 512 episodes over 8 GPUs IS configs[3]
    assert sec["steps"] == 100 and sec["value_min"] <= sec["value"] <= sec["value_max"]
    assert sec["value"] == pytest.approx(64 * world * 100 / (sec["ms_per_step"] * 0.1)) and sec["ms_per_step"] >= 3.0
    assert sec["mean_coverage"] == pytest.approx(np.mean([1000.0 * r + e for r in range(world) for e in range(64)]))
    if world == 2:   # the C2 / C3 entries: every rank crumples and flings its own episodes, the step counts are summed over ranks
        c2a, c2b = rec["configs"][2], rec["configs"][3]
        assert c2a["episodes_per_gpu"] == 4 and c2b["episodes_per_gpu"] == 64 and "C2 scripted fling" in c2a["name"]
        per_rank = 7 * 3 * 4 + 4 * bench_fling_settle    # seven movep legs of 3 steps per episode + the settle steps
        assert c2a["episode_steps"] == 2 * per_rank and c2a["value"] == pytest.approx(c2a["episode_steps"] / c2a["seconds"])
        assert rec["fling_phase_ratio"] == pytest.approx(c2a["value"] / rec["value"])
    for r in range(world):
        got = json.load(open(tmp_path / f"rank{r}.json"))
        assert got["seeds_ok"] and got["sizes"] == ([4, 64, 4, 64] if world == 2 else [4, 64])   # (+ the two C2 contexts at world 2)
        log = [tuple(x) for x in got["log"]]
        assert log[0] == ("set_device", r) and log[1] == ("init", "nccl")
        ev = [e for e, _ in log]
        # headline: pre-roll + warm-up, then the bracketed window of exactly K = 3 steps
        t0 = ev.index("timer_start")
        assert ev[:t0].count("step") == 2 + 1
        assert ev[t0 - 3:t0] == ["barrier", "ctx_sync", "cuda_sync"]
        assert ev[t0 - 5:t0 - 3] == ["gather", "max"]  # the exchange step is warmed up (RCCL builds its communicators in the first collective)
        assert ev[t0:t0 + 10] == ["timer_start", "step", "step", "step", "timer_stop", "gather", "barrier", "ctx_sync",
                                  "cuda_sync", "max"]
        assert log[t0 + 5] == ("gather", "cuda") and log[t0 + 9] == ("max", "cuda")
        # secondary: three bracketed windows of 100 steps each on the 64-episode context
        starts = [i for i, e in enumerate(ev) if e == "timer_start"][1:4]
        assert len(starts) == 3
        for i in starts:
            assert ev[i - 3:i] == ["barrier", "ctx_sync", "cuda_sync"]
            assert ev[i + 1:i + 101] == ["step"] * 100 and ev[i + 101:i + 107] == ["timer_stop", "gather", "barrier", "ctx_sync",
                                                                                "cuda_sync", "max"]
        assert ev[-1] == "barrier" or ev[-2:] == ["close", "barrier"] or "barrier" in ev[-3:]


def test_bench_refuses_two_ranks_on_one_device(tmp_path):
    """The day a multi-GPU run happens the first question is "did RCCL really see N ranks on N GPUs?".  bench.run_rank answers
    from a collective (distributed.rank_census: all_gather of rank / LOCAL_RANK / device identity), and when two ranks report the
    same physical device every rank ends non-zero and NO JSON line is printed."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FS_ROOT=ROOT, FS_OUT=str(tmp_path), FS_SAME_DEVICE="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode not in (0, None) and "saw 2 rank(s) on 1 distinct device(s)" in err, (p.returncode, err[-2000:])
        assert not [l for l in out.splitlines() if l.startswith("{")]


def test_bench_two_device_check_control_flow():
    """bench.two_device_check (rank 0 of a multi-GPU run: its own and its neighbour's device driven from ONE process against the
    oracle) with the GPU context replaced by an oracle-backed stand-in: skipped with a reason on a one-device box, `bit_exact` and
    `distinct` on two devices, and a context that answers wrongly is reported, not raised."""
    import types
    sys.path.insert(0, ROOT)
    import bench
    from oracle import OracleSim

    class Sim:
        FS_SOLVER_FUSED = 2
        wrong = False

        class FlingSim:
            def __init__(self, n_envs=1, device=0, solver=0):
                self.device, self.o = device, OracleSim()
            def set_scene(self, e, p): self.o.set_scene(p)
            def step(self, n=1): self.o.step(n)
            def get_positions(self, e=0):
                p = self.o.get_positions()
                if Sim.wrong and self.device == 1:
                    p = p.copy(); p[5] += 1e-6
                return p
            def get_velocities(self, e=0): return self.o.get_velocities()
            def device_key(self): return "0000:%02x:00.0" % (5 + self.device)
            def close(self): pass

    cuda = lambda n: types.SimpleNamespace(cuda=types.SimpleNamespace(device_count=lambda: n))   # noqa: E731
    one = bench.two_device_check(Sim, cuda(1), 0)
    assert one["checked"] is False and "1 device" in one["reason"]
    two = bench.two_device_check(Sim, cuda(2), 1)
    assert two == {**two, "checked": True, "devices": [1, 0], "distinct": True, "bit_exact": True}
    Sim.wrong = True
    assert bench.two_device_check(Sim, cuda(2), 0)["bit_exact"] is False


def test_rank_census_single_process():
    """World size 1 needs no process group: the census reports itself."""
    sys.path.insert(0, ROOT)
    from flingbot_amd import distributed as fdist
    c = fdist.rank_census("0000:05:00.0", "gfx950")
    assert c["ranks_seen"] == 1 and c["distinct_devices"] == 1 and c["world_size"] == 1 and c["backend"].startswith("none")


# ---- `python -m flingbot_amd.evaluate --tasks set.npz --gpus 8`, end to end, on a machine without a GPU ----------------------
# The command itself runs (argument parsing, the launcher, eight fresh `python -m flingbot_amd.evaluate` children, the block
# split, the coverage gather, rank 0's JSON line); what a child cannot have here -- a HIP device -- is replaced INSIDE the
# children by a sitecustomize module the test puts on their PYTHONPATH: the process group runs on gloo, the GPU context / env /
# policy are inert stand-ins, and run_tasks returns statistics computed from the tasks it was handed (or fails, when told to).
EVAL_SITECUSTOMIZE = r"""
import os, sys, time
if os.environ.get("FS_EVAL_STUB") and os.environ.get("WORLD_SIZE"):
    sys.path.insert(0, os.environ["FS_ROOT"])
    import numpy as np
    import torch
    from flingbot_amd import distributed as fdist, evaluate, nets, sim as fsim
    import flingbot_amd.env as fenv

    RANK = int(os.environ["RANK"])
    open(os.path.join(os.environ["FS_OUT"], f"pid{RANK}"), "w").write(str(os.getpid()))
    torch.cuda.is_available = lambda: False
    torch.cuda.set_device = lambda d: None
    _init, _gather, _sum = fdist.init_from_env, fdist.gather_rewards, fdist.sum_over_ranks
    fdist.init_from_env = lambda backend=None: _init("gloo")
    fdist.gather_rewards = lambda r, device=None: _gather(r, None)
    fdist.sum_over_ranks = lambda a, device=None: _sum(a, None)

    class Ctx:
        def __init__(self, n_envs=1, device=0, solver=0, **kw):
            self.n_envs = n_envs
        def device_key(self):      # PCI bus id of the rank's device; FS_SAME_DEVICE: ranks 0 and 1 were started on ONE GPU
            return "0000:%02x:00.0" % (5 if os.environ.get("FS_SAME_DEVICE") and RANK < 2 else 5 + RANK)
        def close(self):
            pass
    class Env:
        actions, scale_factors = ["fling"], [1.0]
        def __init__(self, ctx, episode_length=10, device=None):
            self.sim = ctx
    class Policy:
        def __init__(self, **kw):
            pass
    def run_tasks(policy, env, tasks, claim=None, claim_first=None):
        if str(RANK) == os.environ.get("FS_FAIL_RANK"):
            time.sleep(1.0)                    # the other ranks are inside the gather by now
            raise RuntimeError(f"injected failure on rank {RANK}")
        if str(RANK) == os.environ.get("FS_KILL_RANK"):
            import signal
            time.sleep(1.0)
            os.kill(os.getpid(), signal.SIGKILL)
        if claim is None:
            idx = list(range(len(tasks)))
        else:   # the shared queue: one task per slot (at most the fair share) up front, then one at a time; rank 0 is 4x as fast
            idx = list(claim(min(env.sim.n_envs, claim_first if claim_first is not None else env.sim.n_envs)))
            while True:
                time.sleep(0.01 if RANK == 0 else 0.04)
                more = claim(1)
                if not more:
                    break
                idx += more
            idx.sort()
        sel = [tasks[i] for i in idx]
        init = np.array([float(t["initial_coverage"]) / float(t["flatten_area"]) for t in sel], np.float32)
        final = init + np.float32(0.125)
        mean = lambda v: float(v.mean()) if len(v) else float("nan")
        return {"init_coverage": init, "final_coverage": final, "simulation_steps": 10 * len(sel), "task_indices": np.array(idx, int),
                "action_primitive_counts": {"fling": 3 * len(sel)}, "records": [],
                "mean": {"init_coverage": mean(init), "final_coverage": mean(final), "best_coverage": mean(final),
                         "episode_delta_coverage": 0.125, "episode_length": 3.0}}
    fsim.FlingSim, fenv.BatchedFlingEnv, nets.MaximumValuePolicy, evaluate.run_tasks = Ctx, Env, Policy, run_tasks
"""


def _write_task_set(path, n):
    from flingbot_amd import taskio
    tasks = []
    for i in range(n):
        tasks.append({"particle_pos": np.zeros(16, np.float32), "particle_vel": np.zeros(12, np.float32), "shape_pos": np.zeros(28, np.float32),
                      "phase": np.zeros(4, np.int32), "cloth_size": np.array([2, 2]), "cloth_stiff": np.array([0.9, 0.9, 0.9]),
                      "mesh_verts": np.array([]), "mesh_stretch_edges": np.array([]), "mesh_bend_edges": np.array([]),
                      "mesh_shear_edges": np.array([]), "mesh_faces": np.array([]), "flatten_area": 0.5, "initial_coverage": 0.01 * (i + 1),
                      "cloth_mass": 0.5, "flip_mesh": 0, "task_difficulty": "hard"})
    assert taskio.save_tasks(path, tasks) == n


def _run_evaluate(tmp_path, extra_env, n_tasks=13, gpus=8, extra_args=()):
    (tmp_path / "site").mkdir(exist_ok=True)
    (tmp_path / "site" / "sitecustomize.py").write_text(EVAL_SITECUSTOMIZE)
    tasks = str(tmp_path / "set.npz")
    _write_task_set(tasks, n_tasks)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(PYTHONPATH=os.pathsep.join([str(tmp_path / "site"), ROOT, env.get("PYTHONPATH", "")]), FS_ROOT=ROOT,
               FS_OUT=str(tmp_path), FS_EVAL_STUB="1", **extra_env)
    t0 = time.time()
    out = subprocess.run([sys.executable, "-m", "flingbot_amd.evaluate", "--tasks", tasks, "--gpus", str(gpus), "--slots", "4", *extra_args],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    pids = [int(open(tmp_path / f"pid{r}").read()) for r in range(gpus) if (tmp_path / f"pid{r}").exists()]
    return out, pids, time.time() - t0


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:  # a zombie that nobody reaped yet still answers kill(0): look at its state
        with open(f"/proc/{pid}/stat") as fh:
            return fh.read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return False


def test_evaluate_command_shared_task_queue_across_ranks(tmp_path):
    """`evaluate --gpus 4` the way the reference runs its workers: ONE task queue (utils.setup_envs' TaskLoader actor; here an
    atomic counter on the process group's store), every rank takes a task whenever it has a free slot.  40 tasks, 4 ranks of 4
    slots, rank 0 four times as fast as the others: every task runs exactly once, the summary covers all 40, rank 0 ends up
    with clearly more than its fair share of 10 -- and with 13 tasks over 8 ranks nobody is left waiting for a rank's block."""
    import json

    out, pids, _ = _run_evaluate(tmp_path, {}, n_tasks=40, gpus=4)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    init = np.array([0.01 * (i + 1) / 0.5 for i in range(40)], np.float32)
    assert rec["gpus"] == 4 and rec["episodes"] == 40 and rec["simulation_steps"] == 400 and "shared task queue" in rec["schedule"]
    assert sum(rec["tasks_per_rank"]) == 40 and rec["tasks_per_rank"][0] >= 14 and min(rec["tasks_per_rank"]) >= 4, rec["tasks_per_rank"]
    assert rec["init_coverage"] == pytest.approx(float(init.mean()), rel=1e-6)
    assert rec["final_coverage"] == pytest.approx(float((init + np.float32(0.125)).mean()), rel=1e-6)
    assert not any(_alive(p) for p in pids)
    out, pids, _ = _run_evaluate(tmp_path, {}, n_tasks=13, gpus=8)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["episodes"] == 13 and sum(rec["tasks_per_rank"]) == 13 and rec["simulation_steps"] == 130
    assert rec["ranks_seen"] == 8 and rec["distinct_devices"] == 8


def test_evaluate_command_eight_ranks_end_to_end_uneven_blocks(tmp_path):
    """--static-blocks, 13 tasks over 8 ranks: blocks of 2, rank 6 gets one, rank 7 none -- and the summary still covers exactly the 13."""
    import json

    out, pids, _ = _run_evaluate(tmp_path, {}, extra_args=("--static-blocks",))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(pids) == 8                 # rank 0 prints, once
    rec = json.loads(lines[0])
    init = np.array([0.01 * (i + 1) / 0.5 for i in range(13)], np.float32)
    assert rec["gpus"] == 8 and rec["episodes"] == 13 and rec["tasks"] == 13 and rec["simulation_steps"] == 130
    assert rec["ranks_seen"] == 8 and rec["distinct_devices"] == 8 and rec["backend"] == "gloo"     # from the collective, not from WORLD_SIZE
    assert rec["init_coverage"] == pytest.approx(float(init.mean()), rel=1e-6)
    assert rec["final_coverage"] == pytest.approx(float((init + np.float32(0.125)).mean()), rel=1e-6)
    assert rec["episode_delta_coverage"] == pytest.approx(0.125, rel=1e-5)
    assert not any(_alive(p) for p in pids)


def test_evaluate_command_refuses_two_ranks_on_one_device(tmp_path):
    """`evaluate --gpus 8` with ranks 0 and 1 on ONE physical device (their device keys agree): the census all_gather shows 8 ranks
    on 7 devices, every rank ends non-zero, no result line, no child left behind."""
    out, pids, _ = _run_evaluate(tmp_path, {"FS_SAME_DEVICE": "1"})
    assert out.returncode != 0 and "saw 8 rank(s) on 7 distinct device(s)" in out.stderr, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert not any(_alive(p) for p in pids)


@pytest.mark.parametrize("how", ["raises", "killed"])
def test_evaluate_command_propagates_a_failing_rank_and_leaves_no_children(tmp_path, how):
    """The failure path of the one-process-per-GPU layout (the reference's counterpart: a Ray worker that dies, utils.py:144-157):
    rank 5 raises inside its evaluation (or is killed) while the other seven wait in the coverage gather.  The command must end
    promptly with that rank's status -- 1 for the exception, 128 + 9 for SIGKILL -- print no result line, and leave none of
    the eight children behind."""
    out, pids, seconds = _run_evaluate(tmp_path, {"FS_FAIL_RANK" if how == "raises" else "FS_KILL_RANK": "5"})
    assert out.returncode == (1 if how == "raises" else 137), (out.returncode, out.stderr[-2000:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    if how == "raises":
        assert "injected failure on rank 5" in out.stderr
    assert len(pids) == 8 and seconds < 120
    deadline = time.time() + 15
    while time.time() < deadline and any(_alive(p) for p in pids):
        time.sleep(0.2)
    assert not any(_alive(p) for p in pids), [p for p in pids if _alive(p)]


def test_evaluate_command_rejects_device_with_several_ranks(tmp_path):
    out = subprocess.run([sys.executable, "-m", "flingbot_amd.evaluate", "--tasks", "x.npz", "--gpus", "2", "--device", "0"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
    assert out.returncode == 2 and "--device names ONE HIP device" in out.stderr


def test_bench_survives_a_rank_failing_inside_the_c2_leg(tmp_path):
    """bench.c2_leg: rank 1's fling script raises (a movep that runs into its step limit) while rank 0's succeeds.  Both ranks
    must agree on the failure before the next collective, so nobody hangs; the headline and the 64-episode entry are printed
    as usual, the failed entry carries `error`, the second C2 entry (64 episodes, which does not fail) is measured normally."""
    import json

    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FS_ROOT=ROOT, FS_OUT=str(tmp_path), FS_C2_FAIL_RANK="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        outs.append(out)
    rec = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["configs"][1]["episodes_per_gpu"] == 64
    by_key = {c.get("key"): c for c in rec["configs"]}
    assert "error" in by_key["c2_fling_4"] and "value" not in by_key["c2_fling_4"]      # rank 0 itself did not fail: "another rank"
    assert "another rank failed" in by_key["c2_fling_4"]["error"]
    assert by_key["c2_fling_64"]["value"] > 0 and "ratio_to_crumpled_sheet" in by_key["c2_fling_64"]
    assert "fling_phase_ratio" not in rec
