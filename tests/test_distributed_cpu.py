"""N > 1 path on CPU: two `gloo` ranks shard episodes, gather per-episode coverage rewards and reduce the timing, using
the same helpers bench.py runs over RCCL on the GPU node (flingbot_amd/distributed.py)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
import numpy as np
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
sys.path.insert(0, os.environ["FS_ROOT"])
import torch
from flingbot_amd import distributed as fdist
from flingbot_amd.evaluate import run_episodes_sharded

def fake_runner(policy, env, tasks):           # stands in for the GPU loop: coverage derived from the task itself
    init = np.array([t["seed"] * 0.01 for t in tasks])
    return {"init_coverage": init, "final_coverage": init + 0.5, "n": len(tasks)}

tasks = [{"seed": g} for g in range(8)]
stats = run_episodes_sharded(None, None, tasks, episodes_per_rank=4, runner=fake_runner)
print(json.dumps({"rank": stats["rank"], "world": stats["world"], "n": stats["n"], "mine": stats["init_coverage"].tolist(),
                  "all_init": stats["all_init_coverage"].tolist(), "all_final": stats["all_final_coverage"].tolist()}))
fdist.barrier()
torch.distributed.destroy_process_group()
"""


def test_sharded_evaluation_gathers_by_global_episode(tmp_path):
    """evaluate.run_episodes_sharded on two gloo ranks: each rank runs only its own tasks, every rank ends with the
    coverages of all episodes in global order."""
    import json

    script = tmp_path / "sharded.py"
    script.write_text(SHARDED_WORKER)
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
    assert np.allclose(outs[0]["mine"], [0.0, 0.01, 0.02, 0.03]) and np.allclose(outs[1]["mine"], [0.04, 0.05, 0.06, 0.07])
    for o in outs:
        assert o["world"] == 2 and o["n"] == 4
        assert np.allclose(o["all_init"], np.arange(8) * 0.01) and np.allclose(o["all_final"], np.arange(8) * 0.01 + 0.5)


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
