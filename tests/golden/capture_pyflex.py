#!/usr/bin/env python
"""Record a PyFleX fixture: the three canonical workloads of tests/scenarios.py on the reference's REAL `pyflex` module.

This is the kit that closes the one pin this repository cannot close itself: the solver arithmetic lives in closed-source
NVIDIA FleX (PyFlex/bindings/main.cpp:2244-2291 -> NvFlexUpdateSolver, NvFlex.h:476-481), which needs an NVIDIA GPU.  Run
this script on a machine where the reference's PyFleX is built (README of the reference: `PYFLEXROOT`, `import pyflex`
works); it needs numpy, that `pyflex`, and tests/scenarios.py next to it -- nothing else of this repository:

    python tests/golden/capture_pyflex.py --out tests/golden/external/pyflex_fixture.npz            # every frame
    python tests/golden/capture_pyflex.py --out fix.npz --scenarios c1,fling --every 5 --dim 64

and copy the file back to tests/golden/external/pyflex_fixture.npz (or point FLINGBOT_PYFLEX_FIXTURE at it).
tests/test_external_fixtures.py then replays the same inputs on the CPU oracle and on the HIP path and reports how far
they are from PyFleX, free-running and re-synchronised at every recorded frame.  It never runs here: there is no PyFleX in
this repository's build image or on its GPU box.

What is recorded per scenario <s> in the .npz:  <s>/frames int32[F] (1-based step index), <s>/positions float32[F][4N],
<s>/velocities float32[F][3N], <s>/shape_states float32[F][14 S], <s>/params float64[19], <s>/args (json), plus
meta (json: backend, every, the scenario list, numpy / platform versions).
"""
import argparse
import json
import os
import platform
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/: scenarios.py (numpy only)


class PyflexSim:
    """The duck-typed `sim` of tests/scenarios.py on the process-global pyflex module (pyflex.cpp:1135-1208)."""

    _initialised = False

    def __init__(self):
        import pyflex

        self.px = pyflex
        if not PyflexSim._initialised:
            pyflex.init(True, False, 720, 720)       # headless, no rendering (4 required positionals, pyflex.cpp:1138)
            PyflexSim._initialised = True

    def set_scene(self, scene_params):
        e_f, e_i = np.zeros(0, np.float32), np.zeros(0, np.int32)
        self.px.set_scene(scene_idx=0, scene_params=np.asarray(scene_params, np.float64), vertices=e_f, stretch_edges=e_i,
                          bend_edges=e_i, shear_edges=e_i, faces=e_i, thread_idx=0)      # flex_utils.py:343-351

    def step(self, n=1):
        for _ in range(n):
            self.px.step()

    def get_positions(self):
        return np.array(self.px.get_positions(), np.float32)

    def set_positions(self, p):
        self.px.set_positions(np.asarray(p).ravel())

    def get_velocities(self):
        return np.array(self.px.get_velocities(), np.float32)

    def set_velocities(self, v):
        self.px.set_velocities(np.asarray(v).ravel())

    def add_sphere(self, radius, pos, quat):
        self.px.add_sphere(float(radius), np.asarray(pos, np.float64), np.asarray(quat, np.float64))

    def get_shape_states(self):
        return np.array(self.px.get_shape_states(), np.float32)

    def set_shape_states(self, s):
        self.px.set_shape_states(np.asarray(s).ravel())


def scenario_args(name, dim, quick=False):
    """Keyword arguments of scenarios.CANONICAL[name] for a capture at cloth side `dim` (stored in the fixture and replayed)."""
    if name == "c1":
        return {"steps": 40 if quick else 200}
    if name == "c2":
        return ({"seed": 0, "dim": dim, "raise_steps": 30, "hold_steps": 10, "settle_steps": 20} if quick else
                {"seed": 0, "dim": dim, "raise_steps": 200, "hold_steps": 100, "settle_steps": 150})
    return {"dim": dim, "settle_steps": 30 if quick else 300}


def capture(make_sim, out_path, names=("c1", "c2", "fling"), every=1, dim=64, quick=False, backend="pyflex"):
    """Run the listed scenarios on make_sim() and write the fixture.  `make_sim` is PyflexSim for the real thing; the
    repository's own test proves the ingest path by passing its CPU oracle here."""
    import scenarios as sc

    data = {}
    for name in names:
        sim = make_sim()
        rec = sc.Recorder(every=every)
        args = scenario_args(name, dim, quick)
        sc.CANONICAL[name](sim, record=rec, **args)
        rec.close()
        data.update(rec.arrays(name + "/"))
        data[name + "/params"] = sc.survey_params(32 if name == "c1" else dim) if name != "fling" else \
            sc.cloth_params(dim, dim, pos=(0.0, -0.2, 0.0))
        data[name + "/args"] = np.array(json.dumps(args))
        print(f"{name}: {rec.count} steps, {len(rec.frames)} frames kept", flush=True)
    data["meta"] = np.array(json.dumps({"backend": backend, "every": every, "scenarios": list(names), "dim": dim,
                                        "numpy": np.__version__, "platform": platform.platform(),
                                        "format": "flingbot_amd pyflex fixture v1"}))
    np.savez_compressed(out_path, **data)
    return out_path


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", required=True)
    ap.add_argument("--scenarios", default="c1,c2,fling")
    ap.add_argument("--every", type=int, default=1, help="keep every k-th frame (1 = all: lets the replay measure ONE-step errors)")
    ap.add_argument("--dim", type=int, default=64, help="cloth side of c2 / fling (c1 is always 32)")
    ap.add_argument("--quick", action="store_true", help="short phases (smoke run)")
    a = ap.parse_args()
    capture(PyflexSim, a.out, tuple(a.scenarios.split(",")), a.every, a.dim, a.quick)
    print("wrote", a.out)
