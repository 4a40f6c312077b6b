"""flingbot_amd.nets against vectors generated from the reference's learning/nets.py (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KW = dict(action_primitives=["fling"], num_rotations=12, scale_factors=[1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75],
          obs_dim=64, pix_grasp_dist=16, pix_drag_dist=16, pix_place_dist=10, rgb_only=True, depth_only=False,
          action_expl_prob=0.0, action_expl_decay=0.9, value_expl_prob=0.0, value_expl_decay=0.9)
TOL = 1e-5  # fp32 CNN forward: absolute tolerance on O(1) outputs


def _policy(device):
    from flingbot_amd import nets

    g = np.load(os.path.join(GOLD, "nets_golden.npz"))
    pol = nets.MaximumValuePolicy(device=device, **KW)
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    pol.load_state_dict(sd, strict=True)  # the reference checkpoint layout loads unchanged
    return pol, g


def test_state_dict_layout_matches_reference():
    from flingbot_amd import nets

    with open(os.path.join(GOLD, "nets_state_dict_keys.json")) as fh:
        ref = json.load(fh)
    assert len(ref) == 108
    pol = nets.MaximumValuePolicy(device="cpu", **KW)
    mine = {k: list(v.shape) for k, v in pol.state_dict().items()}
    assert mine == ref
    assert list(pol.state_dict().keys()) == list(ref.keys())  # same order too
    net = pol.value_nets["fling"]
    n_params = sum(p.numel() for p in net.parameters())  # 37 985 including the `steps` scalar
    assert n_params == 37985, n_params
    assert net.input_channels == 3 and pol.num_transforms == 96
    assert hasattr(net, "mean") and hasattr(net, "std") and hasattr(net, "steps") and hasattr(net, "device")


def test_forward_matches_reference_cpu():
    pol, g = _policy("cpu")
    obs = torch.from_numpy(g["obs"])
    with torch.no_grad():
        out = pol.value_nets["fling"](obs)
    assert out.shape == (3, 1, 24, 24)
    assert np.abs(out.numpy() - g["out"]).max() < TOL
    acted = pol.act([obs, obs[:2]])
    assert np.abs(acted[0]["fling"].numpy() - g["act0"]).max() < TOL
    assert np.abs(acted[1]["fling"].numpy() - g["act1"]).max() < TOL
    assert np.allclose(pol.rotations, g["rotations"])
    # folded-BatchNorm inference path gives the same numbers and leaves the state_dict layout alone
    keys = list(pol.state_dict().keys())
    pol.value_nets["fling"].fold_batchnorm()
    with torch.no_grad():
        out2 = pol.value_nets["fling"](obs)
    assert np.abs(out2.numpy() - g["out"]).max() < 5 * TOL
    assert list(pol.state_dict().keys()) == keys


def test_folded_copy_follows_new_weights():
    """The BatchNorm-folded inference copy is derived data: loading another checkpoint into an already folded policy (also
    through the parent module), a train() phase, or .to() must not leave the OLD weights in the inference path."""
    pol, g = _policy("cpu")
    net = pol.value_nets["fling"]
    obs = torch.from_numpy(g["obs"])
    net.fold_batchnorm()
    with torch.no_grad():
        before = net(obs).clone()
    torch.manual_seed(7)
    sd = {k: (torch.randn_like(v) * 0.1 + (1.0 if k.endswith("running_var") else 0.0)).abs() if v.dtype.is_floating_point and v.dim() > 0
          else v for k, v in pol.state_dict().items()}
    pol.load_state_dict(sd, strict=True)          # parent load -> child hook
    with torch.no_grad():
        after = net(obs).clone()                   # eval mode, folded path (re-folded lazily)
        object.__setattr__(net, "_folded", None)   # the module graph itself, same weights
        plain = net(obs)
    assert not torch.allclose(before, after)
    assert (after - plain).abs().max() < 5 * TOL * max(1.0, float(plain.abs().max()))
    # train() phase with a weight update, then eval(): folded again from the updated weights
    net.fold_batchnorm()
    net.train()
    with torch.no_grad():
        for p_ in net.net.parameters():
            p_.mul_(0.5)
    net.eval()
    with torch.no_grad():
        after2 = net(obs).clone()
        object.__setattr__(net, "_folded", None)
        plain2 = net(obs)
    assert (after2 - plain2).abs().max() < 5 * TOL * max(1.0, float(plain2.abs().max()))


def test_exploration_and_bookkeeping():
    from flingbot_amd import nets

    kw = dict(KW)
    kw.update(action_primitives=["fling", "place"], value_expl_prob=1.0, action_expl_prob=1.0)
    pol = nets.MaximumValuePolicy(device="cpu", **kw)
    assert len(pol.rotations) == 12 and abs(pol.rotations[0] + 90) < 1e-9 and abs(pol.rotations[-1] - 90) < 1e-9
    obs = torch.rand(96, 4, 16, 16)
    pol.obs_dim = 16
    maps = pol.act([obs])[0]
    assert set(maps) == {"fling", "place"} and maps["fling"].shape == (96, 16, 16)
    # action exploration keeps one primitive and floors the other at the kept one's minimum
    consts = [k for k, v in maps.items() if float(v.max() - v.min()) == 0.0]
    assert len(consts) == 1
    pol.decay_exploration()
    assert abs(float(pol.value_expl_prob) - 0.9) < 1e-6
    assert int(pol.steps()) == 0
    # rotations without fling span the full circle
    kw["action_primitives"] = ["place"]
    assert nets.MaximumValuePolicy(device="cpu", **kw).rotations[0] == -180


def test_rotate_stage_matches_reference_scipy_vectors():
    """transform(): the rotate stage equals the reference's call on its own vectors; crop/pad/resize follow OpenCV's
    documented conventions (the reference's cv2 is absent here, so that boundary is unpinned)."""
    from flingbot_amd import nets

    g = np.load(os.path.join(GOLD, "rotate_golden.npz"))
    img = torch.from_numpy(g["img"])  # [4, 40, 40]
    for key in [k for k in g.files if k.startswith("rot_")]:
        ang = float(key[4:])
        out = nets.transform(img, ang, 1.0, 40)  # scale 1, dim == size: only permute + rotate + permute back
        ref = np.swapaxes(g[key], -1, 0)
        assert out.shape == (4, 40, 40)
        assert np.abs(out.numpy() - ref).max() < 1e-6, key
    # scale > 1: replicate pad to int(scale*40) then nearest resize; scale < 1: centre crop
    big = nets.transform(img, 0.0, 1.5, 20)
    small = nets.transform(img, 0.0, 0.5, 20)
    assert big.shape == (4, 20, 20) and small.shape == (4, 20, 20)
    src = img.permute(2, 1, 0).numpy()
    padded = np.pad(src, [(10, 10), (10, 10), (0, 0)], mode="edge")
    assert np.array_equal(big.numpy(), np.swapaxes(padded[(np.arange(20) * 3)][:, (np.arange(20) * 3)], -1, 0))
    assert np.array_equal(small.numpy(), np.swapaxes(src[10:30, 10:30], -1, 0))
    stack = nets.prepare_image(img, [(0.0, 1.0), (30.0, 1.25), (-45.0, 2.0)], 16)
    assert stack.shape == (3, 4, 16, 16) and stack.dtype == torch.float32


def test_resize_nearest_convention():
    from flingbot_amd import nets

    a = np.arange(400 * 400, dtype=np.float32).reshape(400, 400)
    r = nets.resize_nearest(a, 64)
    idx = np.floor(np.arange(64) * (400 / 64)).astype(int)
    assert np.array_equal(r, a[idx][:, idx])
    assert np.array_equal(nets.pad(np.ones((4, 4, 2)), 8).shape, (8, 8, 2))


@pytest.mark.gpu
def test_forward_matches_reference_gpu(gpu_required):
    """Same vectors on the MI355X (PyTorch-ROCm / MIOpen convolutions), plain and folded-BN channels-last paths."""
    pol, g = _policy("cuda")
    obs = torch.from_numpy(g["obs"])
    with torch.no_grad():
        out = pol.value_nets["fling"](obs.cuda()).cpu()
    assert np.abs(out.numpy() - g["out"]).max() < 5 * TOL
    pol.value_nets["fling"].fold_batchnorm()
    acted = pol.act([obs, obs[:2]])
    assert acted[0]["fling"].device.type == "cpu"
    assert np.abs(acted[0]["fling"].numpy() - g["act0"]).max() < 1e-4
    assert np.abs(acted[1]["fling"].numpy() - g["act1"]).max() < 1e-4
    # full-size observation stack: 96 x 4 x 64 x 64, batched over 3 environments
    big = [torch.rand(96, 4, 64, 64) for _ in range(3)]
    res = pol.act(big)
    assert len(res) == 3 and res[0]["fling"].shape == (96, 64, 64)
    single = pol.act([big[1]])[0]["fling"]
    assert np.abs(single.numpy() - res[1]["fling"].numpy()).max() < 1e-4  # batching does not change results


def test_value_net_pack_layout():
    """fs_value_net_pack (host-only C-ABI entry): the packed block holds the folded weights where the kernels expect
    them -- head [ic][tap][oc], per convolution the MFMA B operand of k-step tap*4+cg for lane l = W[l&15][4cg+(l>>4)][tap]."""
    from flingbot_amd import nets

    torch.manual_seed(7)
    net = nets.SpatialValueNet(rgb_only=True, device="cpu").eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape) * 0.2)
            m.running_var.copy_(torch.rand(m.running_var.shape) + 0.5)
    net.fold_batchnorm(hip=True)
    lib, packed = net._hip
    p = packed.numpy()
    assert p.shape == (int(lib.fs_value_net_param_floats()),) and p.shape[0] == 8 + 576 + 16 + 16 * (36 * 64 + 16) + 144
    f = net._folded
    assert np.allclose(p[0:3], 0.18) and np.allclose(p[4:7], 0.1) and p[3] == 0 and p[7] == 1
    w0 = f[0].weight.numpy()
    head = p[8:8 + 576].reshape(4, 9, 16)
    assert np.array_equal(head[:3], w0.reshape(16, 3, 9).transpose(1, 2, 0)) and not head[3].any()
    assert np.array_equal(p[584:600], f[0].bias.numpy())
    convs = [c for b in list(f)[2:-1] for c in (b.c1, b.c2)]
    lanes = np.arange(64)
    for ci, conv in enumerate(convs):
        blk = p[600 + ci * 2320: 600 + (ci + 1) * 2320]
        w = conv.weight.numpy().reshape(16, 16, 9)
        for tap in (0, 4, 8):
            for cg in range(4):
                assert np.array_equal(blk[(tap * 4 + cg) * 64:(tap * 4 + cg + 1) * 64], w[lanes & 15, 4 * cg + (lanes >> 4), tap])
        assert np.array_equal(blk[2304:], conv.bias.numpy())
    assert np.array_equal(p[600 + 16 * 2320:], f[-1].weight.numpy().ravel())
    # CPU observations keep using the folded PyTorch modules
    with torch.no_grad():
        assert net(torch.rand(2, 4, 64, 64)).shape == (2, 1, 64, 64)
