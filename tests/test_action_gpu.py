"""Device action selection (csrc/fs_action.hip) against (a) the golden answers of the REFERENCE's
SimEnv.get_max_value_valid_action (tests/golden/action_golden.npz) and (b) the exhaustive numpy restatement
(oracle/action.py) on larger random cases where most of the best-valued candidates are invalid.  Integer / index work:
the selected flattened index must be identical; the returned 3-D points are computed by the same numpy expressions as the
reference and must be bit-equal."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _selector(prims, rotations, D, gd, dd, pd, reach, sdist=0.3, gh=0.02):
    from flingbot_amd.action import ActionSelector

    return ActionSelector(prims, rotations, D, gd, dd, pd, reach, stretchdrag_dist=sdist, grasp_height=gh)


def test_device_selection_matches_reference_golden(gpu_required):
    g = np.load(os.path.join(GOLD, "action_golden.npz"))
    for ci in range(4):
        D, S, gd, dd, pd, _ = g[f"c{ci}_cfg"].tolist()
        reach, sdist, gh = g[f"c{ci}_reach"].tolist()
        prims = g[f"c{ci}_prims"].tolist()
        sel = _selector(prims, g[f"c{ci}_rotations"].tolist(), D, gd, dd, pd, reach, sdist, gh)
        values = torch.tensor(g[f"c{ci}_values"]).cuda()
        action, params = sel.select(values, g[f"c{ci}_scales"], g[f"c{ci}_depth"])
        want = str(g[f"c{ci}_action"])
        assert (action or "") == want, ci
        if want:
            assert np.array_equal(params["p1"], g[f"c{ci}_p1"]) and np.array_equal(params["p2"], g[f"c{ci}_p2"]), ci


@pytest.mark.parametrize("seed,prims", [(0, ["fling"]), (1, ["fling", "stretchdrag", "drag", "place"]), (2, ["place", "drag"])])
def test_device_selection_matches_exhaustive_oracle(gpu_required, seed, prims):
    from oracle import action as oa

    rng = np.random.default_rng(seed)
    D, S, gd, dd, pd = 48, 160, 8, 10, 6
    num_rot = 12
    rotations = [(2 * i / (num_rot - 1) - 1) * 90 for i in range(num_rot)]
    if "fling" not in prims:
        rotations = [(2 * i / num_rot - 1) * 180 for i in range(num_rot)]
    scales = np.array([1.0, 1.5, 2.0, 2.75])
    yy, xx = np.mgrid[0:S, 0:S]
    depth = np.full((S, S), 2.0, np.float32)
    blob = ((xx - S * 0.4) ** 2 + (yy - S * 0.6) ** 2) < (S * 0.25) ** 2
    depth[blob] = (1.97 - 0.04 * rng.random(blob.sum())).astype(np.float32)
    T = num_rot * len(scales)
    values = rng.random((len(prims), T, D, D)).astype(np.float32)
    values = np.round(values * 4096) / 4096  # plenty of exact ties
    reach = 0.62
    cfg = dict(obs_dim=D, pix_grasp_dist=gd, pix_drag_dist=dd, pix_place_dist=pd, scales=scales, rotations=rotations,
               depth=depth, reach_distance_limit=reach, stretchdrag_dist=0.3, grasp_height=0.02,
               left_arm_base=np.array([0.765, 0, 0]), right_arm_base=np.array([-0.765, 0, 0]))
    want_action, want, want_k = oa.get_max_value_valid_action(values, prims, cfg)
    sel = _selector(prims, rotations, D, gd, dd, pd, reach)
    action, params = sel.select(torch.tensor(values).cuda(), scales, depth)
    assert action == want_action
    assert want_k > 50, "the case must force the walk past many invalid candidates"
    assert params["flat_index"] == want_k
    assert np.array_equal(params["p1"], want["p1"]) and np.array_equal(params["p2"], want["p2"])
    assert np.array_equal(params["pretransform_pixels"], want["pretransform_pixels"])
