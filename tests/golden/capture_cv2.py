#!/usr/bin/env python
"""Record a cv2 / skimage (and optionally pyflex.render) fixture: the library calls of the reference's observation stage
on seeded inputs, for the pins this repository cannot close itself (neither package exists in its build image).

Run on any machine with the reference's Python environment (opencv-python, scikit-image, numpy; flingbot.yml):

    python tests/golden/capture_cv2.py --out tests/golden/external/cv2_fixture.npz
    python tests/golden/capture_cv2.py --out tests/golden/external/cv2_fixture.npz --render     # + one pyflex.render() frame

and copy the file back to tests/golden/external/cv2_fixture.npz (or point FLINGBOT_CV2_FIXTURE at it);
tests/test_external_fixtures.py compares oracle/observe.py, the prepare_image restatement and the HIP observation stage with
it.  Calls recorded, each exactly as the reference makes it:
  resize_u8 / resize_f32   cv2.resize(img, (dim, dim))                        flex_utils.py:425-426 (INTER_LINEAR default)
  hsv, inrange             cv2.cvtColor(rgb, COLOR_RGB2HSV), cv2.inRange(hsv, (0,0,0), (100,100,100))    simEnv.py:704-705
  label                    skimage.morphology.label(mask, return_num=True, background=0)                 utils.py:585-601
  nearest / pad            cv2.resize(img, (dim, dim), interpolation=INTER_NEAREST), cv2.copyMakeBorder(BORDER_REPLICATE)
                                                                                                         nets.py:149-171
  render_rgba / render_depth  pyflex.render() of the canonical flat 64 x 64 cloth with the two pickers   pyflex.cpp:1032-1054
Inputs are generated from numpy.random.RandomState(seed) and stored next to the outputs, so the fixture is self-contained.
"""
import argparse
import json
import platform

import numpy as np


def inputs(seed=0):
    """The seeded inputs (also what the ingest test regenerates to check the file was made by this script)."""
    rng = np.random.RandomState(seed)
    out = {}
    for k, (src, dst) in enumerate(((72, 40), (720, 400), (50, 77), (90, 90))):
        img = rng.randint(0, 256, (src, src, 3)).astype(np.uint8)
        # a blob structure so that colour test and labelling have something to find
        yy, xx = np.mgrid[:src, :src]
        for _ in range(4):
            cy, cx, r = rng.randint(0, src, 3)
            img[(yy - cy) ** 2 + (xx - cx) ** 2 < (r // 4 + 2) ** 2] = rng.randint(0, 256, 3)
        out[f"case{k}/rgb"] = img
        out[f"case{k}/depth"] = (rng.rand(src, src) * 2).astype(np.float32)
        out[f"case{k}/dst"] = np.array(dst)
    out["colours"] = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [100, 100, 100], [255, 255, 255], [0, 0, 0], [200, 100, 50],
                                [10, 200, 190], [101, 100, 100], [100, 100, 101]]], np.uint8)
    out["allcolours"] = rng.randint(0, 256, (256, 256, 3)).astype(np.uint8)
    out["nearest/img"] = rng.rand(4, 400, 400).astype(np.float32).transpose(2, 1, 0).copy()    # (W, H, C) like nets.transform
    return out


class Cv2Backend:
    """The library calls, exactly as the reference makes them."""

    def __init__(self):
        import cv2
        import skimage
        from skimage import morphology as morph

        self.cv2, self.morph = cv2, morph
        self.versions = {"cv2": cv2.__version__, "skimage": skimage.__version__}

    def resize(self, img, dim):                       # flex_utils.py:425-426
        return self.cv2.resize(img, (dim, dim))

    def rgb2hsv(self, rgb):                           # simEnv.py:704-705
        return self.cv2.cvtColor(rgb, self.cv2.COLOR_RGB2HSV)

    def inrange(self, hsv):
        return self.cv2.inRange(hsv, (0, 0, 0), (100, 100, 100))

    def label(self, mask):                            # environment/utils.py:587-590
        return self.morph.label(mask, return_num=True, background=0)

    def crop_center(self, img, crop):                 # nets.py:144-147
        startx = img.shape[1] // 2 - (crop // 2)
        starty = img.shape[0] // 2 - (crop // 2)
        return img[starty:starty + crop, startx:startx + crop, ...]

    def pad(self, img, size):                         # nets.py:150-152
        n = (size - img.shape[0]) // 2
        return self.cv2.copyMakeBorder(img, n, n, n, n, self.cv2.BORDER_REPLICATE)

    def resize_nearest(self, img, dim):               # nets.py:169-170
        return self.cv2.resize(np.ascontiguousarray(img), dsize=(dim, dim), interpolation=self.cv2.INTER_NEAREST)


SCALES = (0.75, 1.0, 1.5, 2.75)


def capture(out_path, seed=0, render=False, backend=None):
    """backend: Cv2Backend() for the real thing; the repository's own test proves the ingest path by passing a backend made
    of its restatements."""
    be = backend if backend is not None else Cv2Backend()
    data = inputs(seed)
    res = {}
    for k in range(4):
        rgb, d, dst = data[f"case{k}/rgb"], data[f"case{k}/depth"], int(data[f"case{k}/dst"])
        r8 = be.resize(rgb, dst)
        res[f"case{k}/resize_u8"] = r8
        res[f"case{k}/resize_f32"] = be.resize(d, dst)
        hsv = be.rgb2hsv(r8)
        res[f"case{k}/hsv"] = hsv
        inr = be.inrange(hsv)
        res[f"case{k}/inrange"] = inr
        mask = (inr == 0).astype(np.uint8)
        lab, num = be.label(mask)
        res[f"case{k}/label"] = np.asarray(lab).astype(np.int32)
        res[f"case{k}/num"] = np.array(num)
    res["colours_hsv"] = be.rgb2hsv(data["colours"])
    res["allcolours_hsv"] = be.rgb2hsv(data["allcolours"])
    img = data["nearest/img"]
    for scale in SCALES:
        new_dim = int(scale * img.shape[0])
        t = be.crop_center(img, new_dim) if scale < 1 else be.pad(img, new_dim) if scale > 1 else img
        res[f"nearest/scale{scale}/shape"] = np.array(t.shape)
        res[f"nearest/scale{scale}/out"] = be.resize_nearest(t, 64)
    meta = {"numpy": np.__version__, "platform": platform.platform(), "seed": seed, "format": "flingbot_amd cv2 fixture v1"}
    meta.update(getattr(be, "versions", {}))
    if render:
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import scenarios as sc
        from capture_pyflex import PyflexSim
        import pyflex

        sim = PyflexSim()
        sc.canonical_flat(sim, 64)
        for p in ((0.5, 0.5, -0.5), (-0.5, 0.5, -0.5)):
            sim.add_sphere(0.02, p, [1, 0, 0, 0])
        sim.step(20)
        res["render/positions"] = sim.get_positions()
        res["render/shape_states"] = sim.get_shape_states()
        rgba, depth = pyflex.render()
        res["render/rgba"], res["render/depth"] = np.array(rgba, np.uint8), np.array(depth, np.float32)
    data.update(res)
    data["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(out_path, **data)
    return out_path


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--render", action="store_true", help="also record one pyflex.render() frame (needs the reference's pyflex)")
    a = ap.parse_args()
    print("wrote", capture(a.out, a.seed, a.render))
