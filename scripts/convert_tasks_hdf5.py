#!/usr/bin/env python
"""Convert a FlingBot task set from the reference's HDF5 (environment/tasks.py:285-320; the released
flingbot-{normal,large}-rect-eval.hdf5 etc., README.md:138-145) to the .npz layout flingbot_amd/taskio.py reads.

    python scripts/convert_tasks_hdf5.py flingbot-normal-rect-eval.hdf5 flingbot-normal-rect-eval.npz

Needs h5py + numpy only (no import of the reference or of this repository): run it on any machine that has h5py -- this
repository's image does not -- and copy the .npz over.  Then:

    python -m flingbot_amd.evaluate --tasks flingbot-normal-rect-eval.npz
"""
import sys

import h5py
import numpy as np

ARRAY_FIELDS = ("particle_pos", "particle_vel", "shape_pos", "phase", "cloth_size", "cloth_stiff", "mesh_verts",
                "mesh_stretch_edges", "mesh_bend_edges", "mesh_shear_edges", "mesh_faces")
SCALAR_FIELDS = ("flatten_area", "initial_coverage", "cloth_mass", "flip_mesh", "task_difficulty")
DEFAULTS = {"flip_mesh": 0, "cloth_mass": 0.5}


def convert(src, dst):
    data = {"format": np.array("flingbot_amd tasks v1")}
    with h5py.File(src, "r") as f:
        names = [k for k in f]                       # the order TaskLoader walks them in (tasks.py:441-442)
        for i, key in enumerate(names):
            g = f[key]
            for field in ARRAY_FIELDS:
                data[f"{i}/{field}"] = np.array(g[field]) if field in g else np.array([])
            for field in SCALAR_FIELDS:
                if field in g.attrs:
                    v = g.attrs[field]
                elif field in g:                     # (a writer that stored a numpy scalar as a dataset)
                    v = np.array(g[field])[()]
                else:
                    v = DEFAULTS[field]
                data[f"{i}/{field}"] = np.array(v.decode() if isinstance(v, bytes) else v)
    data["names"] = np.array(names)
    np.savez_compressed(dst, **data)
    return len(names)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    print(f"{convert(sys.argv[1], sys.argv[2])} tasks -> {sys.argv[2]}")
