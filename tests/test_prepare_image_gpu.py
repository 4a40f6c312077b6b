"""Device prepare_image (csrc/fs_image.hip) against the host restatement of the reference's
transform() -- scipy.ndimage.rotate (cubic spline, mode='nearest') + centre crop / replicate pad + nearest resize
(learning/nets.py:155-193).  Floating point (float64 spline arithmetic rounded to float32): tolerance 2e-6 absolute on
images in [0, 2]; the index chain (crop / pad / nearest resize / axis swaps) must agree exactly, which a wrong pixel
would violate by orders of magnitude on the random image."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-6


def _policy_transforms(num_rotations=12, scales=(0.75, 1.0, 1.5, 2.0, 2.75)):
    rotations = [(2 * i / (num_rotations - 1) - 1) * 90 for i in range(num_rotations)]  # nets.py:213-214
    return [(r, s) for r in rotations for s in scales]


@pytest.mark.parametrize("size,dim", [(40, 16), (97, 64)])
def test_device_prepare_image_matches_host_transform(gpu_required, size, dim):
    from flingbot_amd import nets

    g = torch.Generator().manual_seed(size)
    img = torch.rand(4, size, size, generator=g)
    img[3] = 1.9 + 0.1 * img[3]  # depth-like channel
    yy, xx = np.mgrid[0:size, 0:size]
    img[1] = torch.tensor(((xx // 5 + yy // 7) % 2).astype(np.float32))  # hard edges: spline overshoot paths
    tf = _policy_transforms()
    ref = nets.prepare_image(img, tf, dim)                    # host path: scipy + numpy
    dev = nets.prepare_image(img.cuda(), tf, dim)             # device path
    assert dev.is_cuda and dev.shape == ref.shape == (len(tf), 4, dim, dim) and dev.dtype == torch.float32
    err = (dev.cpu() - ref).abs().max().item()
    assert err < TOL, err


def test_device_prepare_image_golden_rotations(gpu_required):
    """Against rotate_golden.npz (scipy run by tests/golden/make_golden.py): scale 1, dim == size keeps every pixel."""
    import os
    from flingbot_amd import nets

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rotate_golden.npz"))
    img = torch.tensor(g["img"])
    size = img.shape[-1]
    angles = [float(k[4:]) for k in g.files if k.startswith("rot_")]
    dev = nets.prepare_image_device(img.cuda(), [(a, 1.0) for a in angles], size).cpu().numpy()
    for k, a in enumerate(angles):
        want = np.swapaxes(g[f"rot_{a}"], -1, 0)  # (W,H,C) -> (C,H,W), transform()'s last step
        assert np.abs(dev[k] - want).max() < TOL, a
