"""Smoke test of the composed act -> fling loop (scripts/eval_loop_demo.py, BASELINE.json configs[4] in miniature): every
stage is pinned on its own elsewhere in this suite; here they only have to compose on the device and leave a sane state."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_act_fling_loop_composes_on_device(gpu_required):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import eval_loop_demo

    out = eval_loop_demo.main(E=4, S=96, D=64, n_actions=1, seed=0)
    assert np.isfinite(out["cov0"]).all() and np.isfinite(out["cov1"]).all()
    assert (out["cov1"] > 0).all() and (out["cov1"] < 0.2).all()  # a 0.4 m x 0.4 m cloth covers at most 0.16 m^2 (+ margin)
    assert 0 <= out["flung"] <= 4
    assert set(out["stages"]) == {"render", "prepare", "net", "select", "fling", "settle"}
