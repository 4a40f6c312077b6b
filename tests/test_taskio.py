"""Task storage without HDF5: flingbot_amd/taskio.py (.npz interchange, Task, TaskLoader) and the
evaluation loop running from a stored file."""
import os
import sys
import random

import numpy as np
import pytest


def _generated_tasks():
    """The two seeded hard / easy tasks of tests/golden/task_golden.npz, regenerated on the CPU oracle by the product's
    generator (they are pinned to the reference's own generate_randomization there)."""
    from fling_helpers import oracle_generated_tasks

    res = oracle_generated_tasks()
    return [res[k] for k in sorted(res)]


def test_task_file_round_trip_and_loader_semantics(tmp_path, capsys):
    """save_tasks -> TaskLoader: every field of every task comes back bit for bit with the type the consumers expect, the keys
    are the reference writer's (sha1 of the running count, tasks.py:306), get_next_task wraps around like tasks.py:445-463
    (repeat) or stops (no repeat), and Task answers get_config / get_state / get_stats with the reference's key sets."""
    import hashlib
    from flingbot_amd import taskio

    tasks = _generated_tasks()
    assert len(tasks) >= 2
    path = str(tmp_path / "tasks.npz")
    assert taskio.save_tasks(path, [tasks[0], None, tasks[1]]) == 2            # a rejected task (None) is not stored
    loader = taskio.TaskLoader(path)
    assert loader.keys == [hashlib.sha1(f"{i}".encode()).hexdigest() for i in range(2)] and len(loader) == 2
    seen = [loader.get_next_task() for _ in range(5)]                           # 0 1 0 1 0: wrap-around
    assert [t.name for t in seen] == [loader.keys[i % 2] for i in range(5)]
    for t, src in zip(seen[:2], tasks[:2]):
        for f in taskio.ARRAY_FIELDS:
            a, b = np.asarray(t[f]), np.asarray(src[f])
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), f
        for f in taskio.SCALAR_FIELDS:
            assert t[f] == src[f] and type(t[f]) in (float, int, str), (f, type(t[f]))
        assert set(t.get_config()) == {"cloth_pos", "cloth_size", "cloth_stiff", "cloth_mass", "camera_name", "camera_params",
                                       "flip_mesh", "flatten_area", "mesh_verts", "mesh_stretch_edges", "mesh_bend_edges",
                                       "mesh_shear_edges", "mesh_faces"}
        assert set(t.get_state()) == {"particle_pos", "particle_vel", "shape_pos", "phase", "camera_params"}
        assert set(t.get_stats()) == {"task_name", "cloth_mass", "cloth_size", "cloth_stiff", "max_coverage", "task_difficulty",
                                      "init_coverage"}
        assert t.get_config()["camera_params"]["default_camera"]["pos"].tolist() == [0, 2, 0] and "[Task]" in str(t)
    once = taskio.TaskLoader(path, repeat=False)
    once.get_next_task(); once.get_next_task()
    with pytest.raises(StopIteration):
        once.get_next_task()
    # a Task loads into a simulator exactly like the dictionary it was made from (scene arguments + state)
    from flingbot_amd import tasks as ftasks
    for t, src in zip(seen[:2], tasks[:2]):
        a, b = ftasks.task_scene_arguments(t), ftasks.task_scene_arguments(src)
        assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))
    with pytest.raises(ValueError):
        np.savez(str(tmp_path / "other.npz"), format=np.array("something else"), names=np.array([]))
        taskio.TaskLoader(str(tmp_path / "other.npz"))
    # a mesh task has no grid size (tasks.py:357-358)
    m = taskio.Task("m", 1.0, 0.5, "hard", cloth_size=[10, 10], mesh_verts=np.zeros(9))
    assert m.cloth_size.tolist() == [-1, -1]


def test_converter_script_is_standalone():
    """scripts/convert_tasks_hdf5.py must run on a machine that has h5py and nothing of this repository or the reference:
    its imports are h5py, numpy and the standard library, and its field lists are the ones taskio reads."""
    import ast
    from flingbot_amd import taskio

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "convert_tasks_hdf5.py")).read()
    tree = ast.parse(src)
    mods = {n.names[0].name.split(".")[0] for n in ast.walk(tree) if isinstance(n, ast.Import)} | \
           {n.module.split(".")[0] for n in ast.walk(tree) if isinstance(n, ast.ImportFrom)}
    assert mods <= {"sys", "h5py", "numpy"}, mods
    consts = {t.targets[0].id: ast.literal_eval(t.value) for t in tree.body if isinstance(t, ast.Assign) and
              isinstance(t.targets[0], ast.Name) and t.targets[0].id in ("ARRAY_FIELDS", "SCALAR_FIELDS")}
    assert consts["ARRAY_FIELDS"] == taskio.ARRAY_FIELDS and consts["SCALAR_FIELDS"] == taskio.SCALAR_FIELDS
    assert '"flingbot_amd tasks v1"' in src and taskio.FORMAT == "flingbot_amd tasks v1"


def test_converter_script_runs_on_the_reference_writers_layout(tmp_path, monkeypatch):
    """scripts/convert_tasks_hdf5.py EXECUTED (h5py is not in this image, so the two calls it makes -- h5py.File as a context
    manager, groups that are mappings of datasets with an `attrs` mapping -- are served by an in-memory stand-in laid out the
    way the reference's writer lays a file out, environment/tasks.py:303-320: floats / ints / np.float64 / str as group
    attributes, everything else as datasets, group key = sha1 of the running count; h5py hands str attributes back as bytes or
    str depending on its version, both are covered).  The converted file loads through taskio and equals the source tasks."""
    import hashlib
    import importlib.util
    import types

    from flingbot_amd import taskio

    def ref_task(i):
        n = 3 + i
        return {"particle_pos": np.arange(4 * n, dtype=np.float32), "particle_vel": np.zeros(3 * n, np.float32),
                "shape_pos": np.arange(28, dtype=np.float32), "phase": np.full(n, 7, np.int32), "cloth_size": np.array([n, 1]),
                "cloth_stiff": np.array([0.9, 0.8, 0.7]), "mesh_verts": np.array([]), "mesh_stretch_edges": np.array([]),
                "mesh_bend_edges": np.array([]), "mesh_shear_edges": np.array([]), "mesh_faces": np.array([]),
                "flatten_area": np.float64(0.25 + i), "initial_coverage": 0.1 * (i + 1), "cloth_mass": 0.5 + i, "flip_mesh": 0,
                "task_difficulty": "hard" if i % 2 else "easy"}

    class Group(dict):
        def __init__(self):
            super().__init__()
            self.attrs = {}

    store = {}
    for i in range(3):                                            # the reference's writer loop
        g = store[hashlib.sha1(f"{len(store)}".encode()).hexdigest()] = Group()
        for key, value in ref_task(i).items():
            if type(value) in (float, int, np.float64, str):
                g.attrs[key] = value.encode() if isinstance(value, str) and i == 2 else value   # (old h5py: bytes)
            else:
                g[key] = value

    class File:
        def __init__(self, path, mode):
            assert path == "tasks.hdf5" and mode == "r"
        def __enter__(self):
            return store
        def __exit__(self, *exc):
            return False

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=File))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "convert_tasks_hdf5.py")
    spec = importlib.util.spec_from_file_location("convert_tasks_hdf5", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "set.npz")
    assert mod.convert("tasks.hdf5", out) == 3
    tasks = taskio.TaskLoader(out, repeat=False).all_tasks()
    assert [t.name for t in tasks] == list(store)                  # file order = the order the reference's TaskLoader walks
    for i, t in enumerate(tasks):
        want = ref_task(i)
        for f in taskio.ARRAY_FIELDS:
            assert np.array_equal(np.asarray(t[f]), want[f]), (i, f)
        assert float(t.flatten_area) == float(want["flatten_area"]) and float(t.initial_coverage) == want["initial_coverage"]
        assert float(t.cloth_mass) == want["cloth_mass"] and int(t.flip_mesh) == 0 and str(t.task_difficulty) == want["task_difficulty"]


@pytest.mark.gpu
def test_evaluation_loop_runs_from_a_stored_task_file(gpu_required, tmp_path):
    """generate -> save_tasks -> TaskLoader.all_tasks -> evaluate.run_tasks gives exactly the statistics the loop gives on the
    generator's dictionaries (the stored file is a faithful stand-in for the HDF5 sets the reference evaluates on)."""
    import torch
    from flingbot_amd import nets, sim as fsim, taskio, tasks as ftasks
    from flingbot_amd.env import BatchedFlingEnv
    from flingbot_amd.evaluate import run_tasks

    random.seed(2); np.random.seed(2); torch.manual_seed(2)
    n = 3
    gen = fsim.FlingSim(n_envs=n, solver=0)
    made = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters(min_cloth_size=24, strict_min_edge_length=24, max_cloth_size=32) for _ in range(n)])
    gen.close()
    made = [t for t in made if t is not None]
    path = str(tmp_path / "set.npz")
    taskio.save_tasks(path, made)
    stored = taskio.TaskLoader(path, repeat=False).all_tasks()

    def run(tasks):
        torch.manual_seed(5)
        ctx = fsim.FlingSim(n_envs=2, solver=0)
        env = BatchedFlingEnv(ctx, image_dim=128, episode_length=2)
        policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                         obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                         depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                         value_expl_decay=1.0, device="cuda:0")
        stats = run_tasks(policy, env, tasks)
        ctx.close()
        return stats

    a, b = run(made), run(stored)
    for k in ("init_coverage", "final_coverage", "episode_length", "coverage_steps"):
        assert np.array_equal(a[k], b[k]), k
    assert a["action_primitive_counts"] == b["action_primitive_counts"] and a["simulation_steps"] == b["simulation_steps"]


@pytest.mark.gpu
def test_evaluate_command_line_runs_a_stored_set(gpu_required, tmp_path, capsys):
    """`python -m flingbot_amd.evaluate --tasks set.npz` (run_sim.py --eval on a converted task set): generated tasks -> file ->
    the command's own main() -> one JSON line with the reference's summary statistics."""
    import json
    import torch
    from flingbot_amd import evaluate, sim as fsim, taskio, tasks as ftasks

    random.seed(4); np.random.seed(4); torch.manual_seed(4)
    gen = fsim.FlingSim(n_envs=3, solver=0)
    made = [t for t in ftasks.generate_tasks(gen, [ftasks.draw_task_parameters(min_cloth_size=24, strict_min_edge_length=24,
                                                                              max_cloth_size=30) for _ in range(3)]) if t is not None]
    gen.close()
    path = str(tmp_path / "set.npz")
    taskio.save_tasks(path, made)
    capsys.readouterr()
    replay = str(tmp_path / "replay.npz")
    evaluate.main(["--tasks", path, "--slots", "2", "--episode-length", "1", "--dump", replay])
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["tasks"] == len(made) and rec["episode_length"] == 1.0 and rec["simulation_steps"] > 0
    assert 0.0 < rec["init_coverage"] < 1.05 and set(rec["action_primitive_counts"]) == {"fling"}
    # --dump: the episode log SimEnv.on_episode_end would have dumped (taskio.save_replay), and the reference's statistics
    # over it (taskio.collect_stats = utils.collect_stats) agree with the line the command printed
    logged = np.load(replay)
    assert [str(k) for k in logged["keys"]] == ["%09d_step00_last" % i for i in range(len(made))]
    cs = taskio.collect_stats(replay)
    level = str(made[0]["task_difficulty"])
    assert cs[f"final_coverage/{level}/mean"] == pytest.approx(rec["final_coverage"], rel=1e-6)
    # (the log's init_coverage is the TASK's stored one, Task.get_stats() -- simEnv.py:447-450 -- not the one measured after the reset)
    assert cs[f"init_coverage/{level}/mean"] == pytest.approx(np.mean([t["initial_coverage"] / t["flatten_area"] for t in made]), rel=1e-6)
    # the same as ONE rank of a torch.distributed.run launch (process group on RCCL, LOCAL_RANK -> device): same statistics
    import subprocess
    import socket
    import sys
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "flingbot_amd.evaluate", "--tasks", path, "--slots", "2", "--episode-length", "1"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec2 = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    # (the policy is random-initialised per process, so the actions differ; the tasks and the statistics' shape do not)
    assert rec2["tasks"] == rec["tasks"] and rec2["init_coverage"] == rec["init_coverage"] and set(rec2) == set(rec)


def test_episode_log_statistics_equal_the_references_collect_stats(tmp_path):
    """f4's other half, the part the evaluation path needs: SimEnv.on_episode_end -> Memory.dump leaves one HDF5 group per
    action (learning/Memory.py:106-165) and utils.collect_stats (utils.py:186-390) turns the file into the numbers run_sim.py
    prints.  Here: taskio.save_replay writes the same per-action scalars under the same group names into a .npz, and
    taskio.collect_stats must return what the REFERENCE's collect_stats returned for the same log
    (tests/golden/replay_golden.npz, made by tests/golden/make_golden.py replay -- the reference's function run as is over
    the log): every key, exactly -- with the default window of the latest 128 entries and with all of them."""
    from flingbot_amd import taskio

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "replay_golden.npz"))
    keys = [str(k) for k in z["keys"]]
    log = {f: z["log:" + f] for f in ("preaction_coverage", "postaction_coverage", "max_coverage", "init_coverage", "task_difficulty",
                                      "action_primitive", "task_name")}
    # back to the shape evaluate.run_tasks hands over: one record per episode, one task per episode
    records, tasks = [], []
    for i, key in enumerate(keys):
        if "_step00" in key:
            records.append({"coverage": [float(log["preaction_coverage"][i])], "actions": [], "rewards": [], "preaction_coverage": []})
            tasks.append({"cloth_mass": 0.5, "flatten_area": float(log["max_coverage"][i]), "task_difficulty": str(log["task_difficulty"][i]),
                          "initial_coverage": float(log["init_coverage"][i])})
        r = records[-1]
        r["preaction_coverage"].append(float(log["preaction_coverage"][i]))
        r["coverage"].append(float(log["postaction_coverage"][i]))
        r["rewards"].append(float(log["postaction_coverage"][i] - log["preaction_coverage"][i]))
        r["actions"].append(str(log["action_primitive"][i]))
    path = str(tmp_path / "replay.npz")
    assert taskio.save_replay(path, records, tasks) == len(keys)
    assert [str(k) for k in np.load(path)["keys"]] == keys                       # the reference's group names, in its order
    assert len(keys) > 128                                                        # the default window really cuts
    for tag, kw in (("latest128", {}), ("all", {"num_points": 10 ** 6})):
        got = taskio.collect_stats(path, **kw)
        want = {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + ":")}
        assert set(got) == set(want), (sorted(set(got) ^ set(want)))
        for k, v in want.items():
            assert np.array_equal(np.asarray(got[k], np.float64), v), (tag, k, got[k], v)
    with pytest.raises(ValueError):
        taskio.collect_stats(str(tmp_path / "tasks_not_replay.npz")) if taskio.save_tasks(str(tmp_path / "tasks_not_replay.npz"), []) == 0 else None
