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


def nets_vectors():
    """Reference learning/nets.py (stubs: cv2, ray): state_dict layout, a seeded forward, policy rotations."""
    import torch

    stub("cv2")
    ray = stub("ray")
    ray.remote = lambda f: f
    sys.path.insert(0, os.path.join(REF, "learning"))
    import nets as refnets  # the reference module

    torch.manual_seed(0)
    kw = dict(action_primitives=["fling"], num_rotations=12, scale_factors=[1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75],
              obs_dim=64, pix_grasp_dist=16, pix_drag_dist=16, pix_place_dist=10, rgb_only=True, depth_only=False,
              action_expl_prob=0.0, action_expl_decay=0.9, value_expl_prob=0.0, value_expl_decay=0.9, device="cpu")
    pol = refnets.MaximumValuePolicy(**kw)
    g = torch.Generator().manual_seed(1)
    for m in pol.modules():  # make BatchNorm statistics non-trivial
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    pol.eval()
    sd = pol.state_dict()
    obs = torch.rand(3, 4, 24, 24, generator=g)
    obs[:, 3] = 1.9 + 0.1 * obs[:, 3]
    with torch.no_grad():
        out = pol.value_nets["fling"](obs)
        acted = pol.act([obs, obs[:2]])
    arrays = {"sd::" + k: v.numpy() for k, v in sd.items()}
    arrays.update(obs=obs.numpy(), out=out.numpy(), act0=acted[0]["fling"].numpy(), act1=acted[1]["fling"].numpy(),
                  rotations=np.array(pol.rotations), num_transforms=np.array(pol.num_transforms))
    np.savez_compressed(os.path.join(HERE, "nets_golden.npz"), **arrays)
    with open(os.path.join(HERE, "nets_state_dict_keys.json"), "w") as fh:
        json.dump({k: list(v.shape) for k, v in sd.items()}, fh, indent=0)
    # rotate stage of transform() (scipy only; cv2 is absent so pad/resize cannot be run from the reference)
    from scipy import ndimage as nd

    img = torch.rand(4, 40, 40, generator=g)
    rots = {}
    for ang in (-90.0, -40.909, 0.0, 16.3636, 57.27, 90.0):
        rots[f"rot_{ang}"] = nd.rotate(input=img.permute(2, 1, 0), angle=ang, reshape=False, mode="nearest")
    np.savez_compressed(os.path.join(HERE, "rotate_golden.npz"), img=img.numpy(), **rots)
    print("nets:", len(sd), "state_dict entries; out", tuple(out.shape))


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
