"""Generates the golden fixtures under tests/golden/ by running the importable parts of the REFERENCE in this
container (/root/reference is read-only and never travels to the GPU box; only the vectors written here do).

    python tests/golden/make_golden.py

Sources:  environment/flex_utils.py get_current_covered_area (stubs for pyflex, cv2),
          oracle/_ref/camera_ref  (compiled from the reference's PyFlex/core/maths.h by oracle/Makefile),
          learning/nets.py (stubs for cv2, ray), environment/utils.py (stubs for cv2, trimesh, ...).
"""
import json
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def coverage_vectors():
    stub("pyflex")
    stub("cv2")
    sys.path.insert(0, os.path.join(REF, "environment"))
    import flex_utils  # the reference module

    rng = np.random.RandomState(0)
    cases = {}

    def grid(dx, dz, sp=0.00625):
        xs, zs = np.meshgrid(np.arange(dx) * sp, np.arange(dz) * sp)
        p = np.zeros((dx * dz, 4), np.float32)
        p[:, 0], p[:, 2], p[:, 1], p[:, 3] = xs.ravel(), zs.ravel(), 0.005, 1.0
        return p

    cases["flat64"] = grid(64, 64)
    cases["flat32"] = grid(32, 32)
    cases["rect40x20"] = grid(40, 20)
    f = grid(64, 64); f[:, 0] = np.abs(f[:, 0] - 0.2); cases["folded64"] = f
    cases["random500"] = np.concatenate([rng.rand(500, 3) * [0.5, 0.2, 0.3], np.ones((500, 1))], 1).astype(np.float32)
    c = grid(32, 32); c[:, [0, 2]] *= 0.35; c[:, 0] += 0.01 * rng.randn(1024).astype(np.float32); cases["crumpled32"] = c
    cases["shifted"] = grid(48, 30) + np.array([-0.31, 0, 0.77, 0], np.float32)
    out = {}
    for k, p in cases.items():
        out["pos_" + k] = p
        out["area_" + k] = np.float64(flex_utils.get_current_covered_area(pos=p.ravel().copy()))
    np.savez_compressed(os.path.join(HERE, "coverage_golden.npz"), **out)
    print("coverage:", {k[5:]: float(v) for k, v in out.items() if k.startswith("area_")})


def camera_vectors():
    exe = os.path.join(ROOT, "oracle", "_ref", "camera_ref")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/camera_ref"])
    cases = [
        dict(cam=[0, 2, 0], ang=[np.pi / 2, -np.pi / 2, 0], w=720, h=720, lo=[-1.005, -2.005, -1.005], up=[1.005, 1.005, 1.005]),
        dict(cam=[0, 2, 0], ang=[np.pi / 2, -np.pi / 2, 0], w=720, h=720, lo=[-1.005, -0.205, -1.005], up=[1.005, 1.005, 1.005]),
        dict(cam=[0.3, 1.5, 0.7], ang=[0.4, -0.9, 0], w=640, h=480, lo=[-1.2, -0.3, -1.0], up=[1.0, 1.0, 2.5]),
    ]
    out = []
    for c in cases:
        f32 = lambda v: [float(np.float32(x)) for x in v]
        args = [repr(x) for x in (*f32(c["cam"]), *f32(c["ang"]), float(c["w"]), float(c["h"]), *f32(c["lo"]), *f32(c["up"]))]
        txt = subprocess.check_output([exe] + args).decode()
        rec = dict(c)
        rec["cam"], rec["ang"], rec["lo"], rec["up"] = f32(c["cam"]), f32(c["ang"]), f32(c["lo"]), f32(c["up"])
        for line in txt.strip().split("\n"):
            name, *vals = line.split()
            rec[name] = [float(v) for v in vals]
        out.append(rec)
    with open(os.path.join(HERE, "camera_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("camera:", len(out), "cases")


if __name__ == "__main__":
    which = sys.argv[1:] or ["coverage", "camera", "nets", "envutils"]
    if "coverage" in which:
        coverage_vectors()
    if "camera" in which:
        camera_vectors()
    if "nets" in which and "nets_vectors" in globals():
        nets_vectors()
    if "envutils" in which and "envutils_vectors" in globals():
        envutils_vectors()
