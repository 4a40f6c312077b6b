#!/usr/bin/env python
"""bench_dropin.py -- what a user of the reference's API sees: the `pyflex` drop-in module (csrc/pyflex_module.cpp), one
process-global cloth per process, called exactly the way the reference calls PyFleX.

    python bench_dropin.py                       # both workloads, 1 process and 16 processes sharing the GPU -> one JSON line
    python bench_dropin.py --worker c1 ...       # (internal) one worker process

Workloads:
  c1      BASELINE.json configs[0] verbatim: 32 x 32 cloth, pyflex.init / set_scene (+ its step) / set_to_flatten, then 200
          pyflex.step() and one get_positions.
  picker  the call pattern of Picker.step around every simulation step (environment/flex_utils.py:104-205):
          get_shape_states + get_positions (Picker._get_pos), get_shape_states + set_shape_states + set_positions
          (Picker._set_pos), pyflex.step() -- on a 64 x 64 cloth with the two picker spheres.
The 16-process case is run three times: as an unmodified caller gets it (FLINGSIM_SHARED_GPU unset: the module's co-tenant
table, csrc/fs_tenants.cpp, finds the other workers and picks the back-end), with the detection overridden to "alone"
(FLINGSIM_SHARED_GPU=0: the streaming kernels, the default of rounds 1-5) and to "shared" (=1).
Processes: the reference runs one PyFleX per Ray worker (`--num_processes 16`, README.md:147-148, utils.py:144-157); here 16
fresh interpreters, each with its own pyflex.init, started together and released by a wall-clock start time, share one
MI355X.  The parent of the workers never touches the GPU.  Reported next to the batched face's numbers in bench.py.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def worker(mode, steps, start_at):
    sys.path.insert(0, os.path.join(ROOT, "flingbot_amd", "pyflex_native"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import pyflex
    import scenarios as sc

    pyflex.init(True, False, 720, 720)
    dim = 32 if mode == "c1" else (96 if mode == "picker96" else 64)   # picker96: a cloth the fused kernel cannot take (not in measure())
    e_f, e_i = np.zeros(0, np.float32), np.zeros(0, np.int32)
    pyflex.set_scene(scene_idx=0, scene_params=sc.survey_params(dim), vertices=e_f, stretch_edges=e_i, bend_edges=e_i,
                     shear_edges=e_i, faces=e_i, thread_idx=0)
    pyflex.step()
    pyflex.set_positions(sc.set_to_flatten_positions(dim, dim).flatten())
    if mode in ("picker", "picker96"):
        for c in ((0.04, 0.1, 0.0), (-0.04, 0.1, 0.0)):
            pyflex.add_sphere(0.02, np.array(c), np.array([1., 0., 0., 0.]))
    pyflex.step()
    pyflex.get_positions()                       # warm: everything allocated, the device idle
    tenants0 = pyflex._tenants()
    while time.time() < start_at:
        time.sleep(0.0005)
    t0 = time.time()
    if mode == "c1":
        for _ in range(steps):
            pyflex.step()
        pyflex.get_positions()
    else:
        delta = np.array([0.0, 1e-4, 0.0])
        for _ in range(steps):
            picker = np.array(pyflex.get_shape_states()).reshape(-1, 14)          # Picker._get_pos
            particles = np.array(pyflex.get_positions()).reshape(-1, 4)
            new_picker = picker[:, :3] + delta
            st = np.array(pyflex.get_shape_states()).reshape(-1, 14)              # Picker._set_pos
            st[:, 3:6] = st[:, :3]
            st[:, :3] = new_picker
            pyflex.set_shape_states(st)
            pyflex.set_positions(particles)
            pyflex.step()
        pyflex.get_positions()
    t1 = time.time()
    print(json.dumps({"mode": mode, "steps": steps, "t0": t0, "t1": t1, "tenants_at_start": tenants0[0], "backend_at_start": tenants0[1],
                      "tenants_at_end": pyflex._tenants()[0], "backend_at_end": pyflex._tenants()[1]}), flush=True)


def run(mode, n_procs, steps, timeout=300, shared_gpu=None):
    """n_procs fresh interpreters, released together; returns aggregate steps/s over [first start, last end].
    shared_gpu: None = FLINGSIM_SHARED_GPU unset, what an unmodified caller gets (csrc/pyflex_module.cpp asks the co-tenant
    table); False / True = the variable set to 0 / 1 (never / always the one-launch-per-frame kernel for a cloth that fits it)."""
    start_at = time.time() + 6.0 + 0.5 * n_procs          # enough for every child to import, init and warm up
    env = {k: v for k, v in os.environ.items() if k != "FLINGSIM_SHARED_GPU"}
    if shared_gpu is not None:
        env["FLINGSIM_SHARED_GPU"] = "1" if shared_gpu else "0"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", mode, "--steps", str(steps),
                               "--start-at", repr(start_at)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for _ in range(n_procs)]
    recs, errs = [], []
    for p in procs:
        try:
            out, err = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
        line = [ln for ln in out.splitlines() if ln.startswith("{")]
        if p.returncode == 0 and line:
            recs.append(json.loads(line[-1]))
        else:
            errs.append((err or out).strip().splitlines()[-1:] or ["?"])
    if not recs:
        return {"processes": n_procs, "error": str(errs[:2])}
    late = sum(1 for r in recs if r["t0"] > start_at + 0.05)
    span = max(r["t1"] for r in recs) - min(r["t0"] for r in recs)
    return {"processes": n_procs, "shared_gpu_switch": {None: "unset (detected)", False: "0", True: "1"}[shared_gpu],
            "backends_at_end": sorted({r.get("backend_at_end", "?") for r in recs}),
            "tenants_seen": [min(r.get("tenants_at_end", 0) for r in recs), max(r.get("tenants_at_end", 0) for r in recs)],
            "finished": len(recs), "steps_per_process": steps, "seconds": span,
            "steps_per_s": sum(r["steps"] for r in recs) / span,
            "slowest_process_steps_per_s": min(r["steps"] / (r["t1"] - r["t0"]) for r in recs),
            "late_starters": late, "failed": len(errs)}


def measure(procs=(1, 16)):
    out = {"module": "flingbot_amd/pyflex_native/pyflex (csrc/pyflex_module.cpp over libflingsim.so)",
           "note": "one cloth per process like the reference (Ray worker = PyFleX instance); every getter / setter is a "
                   "synchronous copy as in pyflex.cpp.  The batched face (FlingSim: all episodes of a process in one launch "
                   "sequence, movep on the device) is the supported throughput path; this is what unmodified callers get."}
    for key, mode in (("c1_32x32_200_steps", "c1"), ("picker_pattern_64x64", "picker")):
        out[key] = [run(mode, n, 200) for n in procs] + [run(mode, max(procs), 200, shared_gpu=False), run(mode, max(procs), 200, shared_gpu=True)]
    out["shared_gpu_switch"] = ("unset: pyflex.init registers the process in the device's co-tenant table (a per-user file under /dev/shm), "
                                "set_scene and every 64th step count the live tenants; with two or more the module steps a cloth that fits "
                                "it on the fused kernel (one launch per frame on one compute unit) instead of the streaming kernels (129 "
                                "launches per frame): slower for a lone process, but sixteen of them run side by side where sixteen launch "
                                "chains share the chip's dispatch rate.  FLINGSIM_SHARED_GPU=0 / 1 overrides the detection")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", default=None)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--start-at", type=float, default=0.0)
    a = ap.parse_args()
    if a.worker:
        worker(a.worker, a.steps, a.start_at)
    else:
        print(json.dumps(measure()))
