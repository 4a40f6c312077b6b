"""Task storage without HDF5: the interchange format of the evaluation loop's task sets.

The reference keeps task sets in HDF5 (environment/tasks.py:285-320 writes one group per task: scalars as attributes,
arrays as gzip datasets; `TaskLoader` / `Task`, tasks.py:323-463, read them back), and `h5py` is not part of this
repository's image.  The released sets (`flingbot-*-eval.hdf5`, README.md:138-145) are converted ONCE, on any machine that
has h5py, by `scripts/convert_tasks_hdf5.py` (pure h5py + numpy) into the `.npz` layout below; everything here reads and
writes that layout with numpy alone.

Layout ("flingbot_amd tasks v1"), one flat .npz:
    format            the string above
    names             str[n]: the HDF5 group keys, in file order (the order TaskLoader walks them in)
    <i>/<field>       for task i = 0 .. n-1 and every field of the reference's task dictionary (tasks.py:236-253):
        arrays   particle_pos float32[4N], particle_vel float32[3N], shape_pos float32[14 S], phase int32[N],
                 cloth_size int[2], cloth_stiff float[3], mesh_verts float[3V], mesh_stretch_edges / mesh_bend_edges /
                 mesh_shear_edges int[2 e], mesh_faces int[3 F]                  (the HDF5 datasets)
        scalars  flatten_area, initial_coverage, cloth_mass float; flip_mesh int; task_difficulty str   (the HDF5 attributes)
`Task` has the reference class's constructor, attributes and get_config / get_state / get_stats, and also answers
`task[field]` so that the functions of flingbot_amd.tasks (load_task_scene, ScenePrebuilder, ...) and evaluate.run_tasks take
it wherever they take the generator's dictionaries.
"""
import numpy as np

FORMAT = "flingbot_amd tasks v1"
ARRAY_FIELDS = ("particle_pos", "particle_vel", "shape_pos", "phase", "cloth_size", "cloth_stiff", "mesh_verts",
                "mesh_stretch_edges", "mesh_bend_edges", "mesh_shear_edges", "mesh_faces")
SCALAR_FIELDS = ("flatten_area", "initial_coverage", "cloth_mass", "flip_mesh", "task_difficulty")
_CAMERA = {"pos": np.array([0, 2, 0]), "angle": np.array([np.pi * 0.5, -np.pi * 0.5, 0]), "width": 720, "height": 720}


class Task:
    """One stored task (environment/tasks.py:323-434): same constructor arguments, attributes and accessors."""

    def __init__(self, name, flatten_area, initial_coverage, task_difficulty, cloth_size=None, flip_mesh=0, particle_pos=(),
                 particle_vel=(), shape_pos=(), mesh_verts=(), mesh_stretch_edges=(), mesh_bend_edges=(), mesh_shear_edges=(),
                 mesh_faces=(), phase=(), cloth_stiff=(), cloth_mass=0.5, cloth_pos=(0, 2, 0)):
        self.name = name
        self.flatten_area, self.initial_coverage = flatten_area, initial_coverage
        self.task_difficulty, self.cloth_mass, self.flip_mesh = task_difficulty, cloth_mass, flip_mesh
        for field, value in (("cloth_size", cloth_size), ("particle_pos", particle_pos), ("particle_vel", particle_vel),
                             ("shape_pos", shape_pos), ("phase", phase), ("cloth_pos", cloth_pos), ("cloth_stiff", cloth_stiff),
                             ("mesh_verts", mesh_verts), ("mesh_stretch_edges", mesh_stretch_edges),
                             ("mesh_bend_edges", mesh_bend_edges), ("mesh_shear_edges", mesh_shear_edges),
                             ("mesh_faces", mesh_faces)):
            setattr(self, field, np.array(value))
        if len(self.mesh_verts) > 0:      # a mesh cloth has no grid size (tasks.py:357-358)
            self.cloth_size = np.array([-1, -1])
        self.camera_pos, self.camera_angle = _CAMERA["pos"].copy(), _CAMERA["angle"].copy()   # the 'top_down' view
        self.camera_width, self.camera_height = _CAMERA["width"], _CAMERA["height"]

    # ---- dictionary face: what flingbot_amd.tasks / evaluate.run_tasks read from a generated task
    def __getitem__(self, field):
        if field in ARRAY_FIELDS or field in SCALAR_FIELDS or field in ("name", "cloth_pos"):
            return getattr(self, field)
        raise KeyError(field)

    def keys(self):
        return ARRAY_FIELDS + SCALAR_FIELDS

    def as_dict(self):
        return {k: self[k] for k in self.keys()}

    def _camera_params(self):
        return {"default_camera": {"pos": self.camera_pos, "angle": self.camera_angle, "width": self.camera_width,
                                   "height": self.camera_height}}

    def get_config(self):
        return {"cloth_pos": self.cloth_pos, "cloth_size": self.cloth_size, "cloth_stiff": self.cloth_stiff,
                "cloth_mass": self.cloth_mass, "camera_name": "default_camera", "camera_params": self._camera_params(),
                "flip_mesh": self.flip_mesh, "flatten_area": self.flatten_area, "mesh_verts": self.mesh_verts,
                "mesh_stretch_edges": self.mesh_stretch_edges, "mesh_bend_edges": self.mesh_bend_edges,
                "mesh_shear_edges": self.mesh_shear_edges, "mesh_faces": self.mesh_faces}

    def get_state(self):
        return {"particle_pos": self.particle_pos, "particle_vel": self.particle_vel, "shape_pos": self.shape_pos,
                "phase": self.phase, "camera_params": self._camera_params()}

    def get_stats(self):
        return {"task_name": self.name, "cloth_mass": self.cloth_mass, "cloth_size": self.cloth_size,
                "cloth_stiff": self.cloth_stiff, "max_coverage": self.flatten_area, "task_difficulty": self.task_difficulty,
                "init_coverage": self.initial_coverage}

    def __str__(self):
        return (f"[Task] {self.name}\n\ttask_difficulty: {self.task_difficulty}\n"
                f"\tinitial_coverage (%): {self.initial_coverage * 100 / self.flatten_area:.02f}\n"
                f"\tcloth_mass (kg): {self.cloth_mass:.04f}\n\tcloth_size: {self.cloth_size}\n"
                f"\tcloth_stiff: {self.cloth_stiff}\n\tflatten_area (m^2): {self.flatten_area:.04f}\n")


def save_tasks(path, tasks, names=None):
    """Write task dictionaries (flingbot_amd.tasks.generate_tasks) or Task objects; None entries (rejected tasks) are skipped.
    names: the group keys; default = what the reference's writer would have used, sha1 of the running count
    (tasks.py:306)."""
    import hashlib

    kept = [t for t in tasks if t is not None]
    if names is None:
        names = [getattr(t, "name", None) or hashlib.sha1(f"{i}".encode()).hexdigest() for i, t in enumerate(kept)]
    assert len(names) == len(kept)
    data = {"format": np.array(FORMAT), "names": np.array([str(n) for n in names])}
    for i, t in enumerate(kept):
        for f in ARRAY_FIELDS:
            data[f"{i}/{f}"] = np.asarray(t[f])
        for f in SCALAR_FIELDS:
            data[f"{i}/{f}"] = np.array(t[f])
    np.savez_compressed(path, **data)
    return len(kept)


class TaskLoader:
    """environment/tasks.py:436-463 over a converted file: walks the tasks in file order, wraps around when `repeat`.
    Where the reference (repeat=False) prints 'Out of tasks' and sleeps forever, this raises StopIteration."""

    def __init__(self, path: str, repeat: bool = True):
        self.path, self.repeat = path, repeat
        self._z = np.load(path, allow_pickle=False)
        if str(self._z["format"]) != FORMAT:
            raise ValueError(f"{path}: not a '{FORMAT}' file (convert the reference's HDF5 with scripts/convert_tasks_hdf5.py)")
        self.keys = [str(k) for k in self._z["names"]]
        print(f"[TaskLoader] Found {len(self.keys)} tasks from", path)
        self.curr_task_idx = 0

    def __len__(self):
        return len(self.keys)

    def task(self, i) -> Task:
        fields = {f: self._z[f"{i}/{f}"] for f in ARRAY_FIELDS}
        scal = {f: self._z[f"{i}/{f}"].item() for f in SCALAR_FIELDS}
        return Task(name=self.keys[i], **scal, **fields)

    def get_next_task(self) -> Task:
        if self.curr_task_idx >= len(self.keys):
            raise StopIteration("[TaskLoader] Out of tasks")
        t = self.task(self.curr_task_idx)
        self.curr_task_idx += 1
        if not self.repeat:
            print("[TaskLoader] {}/{}".format(self.curr_task_idx, len(self.keys)))
        if self.curr_task_idx >= len(self.keys) and self.repeat:
            self.curr_task_idx = 0
        return t

    def all_tasks(self):
        """Every task once, in file order: the list evaluate.run_tasks takes."""
        return [self.task(i) for i in range(len(self.keys))]


# ---- the episode log (the evaluation half of learning/Memory.py) -------------------------------------------------------------
REPLAY_FORMAT = "flingbot_amd replay v1"
REPLAY_SCALARS = ("preaction_coverage", "postaction_coverage", "rewards", "is_terminal", "action_primitive", "task_name",
                  "task_difficulty", "max_coverage", "init_coverage", "cloth_mass")


def save_replay(path, records, tasks, first_episode=0, episode_ids=None):
    """What SimEnv.on_episode_end -> Memory.dump (simEnv.py:783-805, learning/Memory.py:106-165) leaves of an EVALUATION
    episode: one entry per action, keyed like the reference's HDF5 groups -- '%09d_step%02d', the episode's last one with
    '_last' -- holding the scalars SimEnv.step / log_step_stats record (simEnv.py:433-452,477-503) and utils.collect_stats
    reads (utils.py:186-330): coverage before / after the action, reward, termination, the primitive, the task's get_stats().
    Not stored: observations, action masks and value maps -- the training set of run_sim.py's optimizer, out of scope here
    (DESIGN.md 8).  records: evaluate.run_tasks(...)['records']; tasks: the tasks they ran on (Task objects or generator
    dictionaries); one flat .npz, `keys` in the order the reference's file would list its groups.  Episode numbers are
    first_episode + position, or episode_ids[position] (a rank of a run with one shared task queue: the tasks' own indices)."""
    keys, data = [], {"format": np.array(REPLAY_FORMAT)}
    for i, (rec, task) in enumerate(zip(records, tasks)):
        n = len(rec["actions"])
        stats = task.get_stats() if hasattr(task, "get_stats") else {
            "task_name": str(i), "cloth_mass": task["cloth_mass"], "max_coverage": task["flatten_area"],
            "task_difficulty": task["task_difficulty"], "init_coverage": task["initial_coverage"]}
        for k in range(n):
            ep = first_episode + i if episode_ids is None else int(episode_ids[i])
            key = f"{ep:09d}_step{k:02d}" + ("_last" if k == n - 1 else "")
            keys.append(key)
            row = {"preaction_coverage": rec["preaction_coverage"][k], "postaction_coverage": rec["coverage"][k + 1],
                   "rewards": rec["rewards"][k], "is_terminal": float(k == n - 1),
                   "action_primitive": "none" if rec["actions"][k] is None else rec["actions"][k],
                   "task_name": stats["task_name"], "task_difficulty": stats["task_difficulty"],
                   "max_coverage": float(stats["max_coverage"]), "init_coverage": float(stats["init_coverage"]),
                   "cloth_mass": float(stats["cloth_mass"])}
            for f in REPLAY_SCALARS:
                data[f"{key}/{f}"] = np.array(row[f])
    data["keys"] = np.array(keys)
    np.savez_compressed(path, **data)
    return len(keys)


def collect_stats(path, num_points=128, action_primitives=("fling", "stretchdrag", "drag", "place")):
    """utils.collect_stats (utils.py:186-390) over a file written by save_replay, with the reference's key names
    ('<statistic>/<level>/mean|max|min|distribution', 'delta_coverage/<level>/percent_positive|negative|zero',
    'action_primitive/percent_fling|drag|place'): the LATEST `num_points` entries only (the reference's default window of 128
    -- an episode cut by the window contributes the steps that are inside it), entries whose postaction coverage is below 5 %
    of the flattened area skipped, best coverage tracked per episode through a running slot that an episode's '_last' entry
    closes.  Pinned to the reference's function by tests/golden/replay_golden.npz (the per-step dictionaries and the
    before / after images of the training dashboard are not produced)."""
    z = np.load(path, allow_pickle=False)
    if str(z["format"]) != REPLAY_FORMAT:
        raise ValueError(f"{path}: not a '{REPLAY_FORMAT}' file")
    keys = sorted(str(k) for k in z["keys"])          # HDF5 lists its groups by name
    if len(keys) > num_points:
        keys = keys[-num_points:]
    names = ("delta_coverage", "final_coverage", "init_coverage", "best_coverage", "episode_delta_coverage", "episode_length")
    stats = {k: {"easy": [], "hard": []} for k in names}
    for level in ("easy", "hard"):
        stats["best_coverage"][level] = [-1]
    counts = {ap: 0 for ap in action_primitives}
    for key in keys:
        g = {f: z[f"{key}/{f}"].item() for f in REPLAY_SCALARS}
        mx = g["max_coverage"]
        if g["postaction_coverage"] / mx < 0.05:
            continue
        level = str(g["task_difficulty"])
        stats["delta_coverage"][level].append((g["postaction_coverage"] - g["preaction_coverage"]) / mx)
        if g["action_primitive"] in counts:     # (an episode step without a valid action is stored as "none")
            counts[g["action_primitive"]] += 1
        stats["best_coverage"][level][-1] = max(stats["best_coverage"][level][-1], g["postaction_coverage"] / mx)
        if "last" in key:
            stats["episode_length"][level].append(int(key.split("step")[1].split("_")[0]))
            stats["final_coverage"][level].append(g["postaction_coverage"] / mx)
            stats["init_coverage"][level].append(g["init_coverage"] / mx)
            stats["best_coverage"][level].append(-1)
            stats["episode_delta_coverage"][level].append(stats["final_coverage"][level][-1] - g["init_coverage"] / mx)
    for level in ("easy", "hard"):
        del stats["best_coverage"][level][-1]
    out = {}
    for key in names:
        for level, values in stats[key].items():
            if len(values) == 0:
                continue
            v = np.array(values)
            out[f"{key}/{level}/distribution"] = v
            out[f"{key}/{level}/mean"], out[f"{key}/{level}/max"], out[f"{key}/{level}/min"] = v.mean(), v.max(), v.min()
            if key == "delta_coverage":
                out[f"{key}/{level}/percent_positive"] = np.count_nonzero(v > 0.0) / len(v)
                out[f"{key}/{level}/percent_negative"] = np.count_nonzero(v < 0.0) / len(v)
                out[f"{key}/{level}/percent_zero"] = np.count_nonzero(v == 0.0) / len(v)
    for ap in ("fling", "drag", "place"):
        out[f"action_primitive/percent_{ap}"] = counts[ap] / len(keys) if keys else float("nan")
    return out
