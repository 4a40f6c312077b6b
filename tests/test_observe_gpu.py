"""fs_observe (csrc/fs_observe.hip) against oracle/observe.py: the observation stage between pyflex.render and prepare_image
-- flip, cv2.resize (INTER_LINEAR), HSV cloth mask, largest connected component, bounding box, preprocess_obs -- computed on
the device from the frame the rasteriser just produced.  Bit-exact: integer pixel arithmetic, float32 depth taps in the
oracle's operation order.  (cv2 / skimage are absent here: the oracle restates their documented algorithms; parity with the
reference's own cv2 build is unpinned, see oracle/observe.py.)"""
import numpy as np
import pytest

from conftest import cloth_params

pytestmark = pytest.mark.gpu


def _scene(ctx, e, dims, seed, crumple=True):
    env = ctx.env(e)
    env.set_scene(cloth_params(dims[0], dims[1], pos=(0.0, 0.3 if crumple else 0.02, 0.0)))
    if crumple:
        rng = np.random.RandomState(seed)
        p = env.get_positions().reshape(-1, 4)
        p[:, :3] += rng.randn(*p[:, :3].shape).astype(np.float32) * 0.004
        env.set_positions(p.ravel())
    return env


@pytest.mark.parametrize("render_dim,image_dim", [(720, 400), (720, 128), (256, 256), (300, 77), (128, 200)])
def test_observe_matches_oracle(gpu_required, render_dim, image_dim):
    from flingbot_amd import sim as fsim
    from oracle import observe as oo

    ctx = fsim.FlingSim(n_envs=2, solver=0)
    for e, dims in enumerate([(40, 30), (64, 64)]):
        env = _scene(ctx, e, dims, seed=e)
        ctx.step(25 + 20 * e)
        cp = ctx.get_camera_params(e)
        ctx.set_camera_params(e, [*cp[2:8], render_dim, render_dim])
        rgba, depth = ctx.render(e)
        obs, bbox, mask = ctx.observe(e, image_dim, want_mask=True)
        ref_obs, rgb, d, ref_mask, crop = oo.get_obs(rgba, depth, render_dim, image_dim)
        got = obs.cpu().numpy()
        assert got.shape == (4, image_dim, image_dim)
        assert np.array_equal(got[:3], ref_obs[:3]), "rgb planes"
        assert np.array_equal(got[3], ref_obs[3]), "depth plane"
        assert ref_mask is not None and ref_mask.any()
        assert np.array_equal(mask.cpu().numpy(), ref_mask)
        x, y = np.where(ref_mask)
        assert bbox.tolist() == [x.min(), x.max(), y.min(), y.max(), int(ref_mask.sum())]
    ctx.close()


def test_largest_component_of_several_and_empty(gpu_required):
    """Several blobs in view (cloth + gripper spheres): the biggest one wins; an episode whose cloth left the view reports no component."""
    from flingbot_amd import sim as fsim
    from oracle import observe as oo

    ctx = fsim.FlingSim(n_envs=1, solver=0)
    env = _scene(ctx, 0, (48, 48), seed=3, crumple=False)
    ctx.step(5)
    q = env.get_positions().reshape(-1, 4).copy()
    env.add_sphere(0.03, [0.45, 0.05, 0.3], [1, 0, 0, 0])   # a gripper sphere away from the cloth: a second blob
    env.add_sphere(0.03, [-0.5, 0.05, -0.35], [1, 0, 0, 0])
    cp = ctx.get_camera_params(0)
    ctx.set_camera_params(0, [*cp[2:8], 360, 360])
    rgba, depth = ctx.render(0)
    obs, bbox, mask = ctx.observe(0, 200, want_mask=True)
    ref_obs, rgb, d, ref_mask, crop = oo.get_obs(rgba, depth, 360, 200)
    raw = oo.cloth_mask_raw(rgb)
    assert raw.sum() > ref_mask.sum() > 0          # more than one component in the raw mask
    assert np.array_equal(mask.cpu().numpy(), ref_mask) and bbox[4] == ref_mask.sum()
    assert np.array_equal(obs.cpu().numpy(), ref_obs)
    q[:, 0] += 50.0                                  # cloth out of view: a gripper sphere is the largest blob left
    env.set_positions(q.ravel())
    obs, bbox = ctx.observe(0, 200)
    assert 0 < bbox[4] < 200 and bbox[1] - bbox[0] < 20 and bbox[3] - bbox[2] < 20
    env.clear_shapes()                               # nothing left that passes the colour test
    obs, bbox = ctx.observe(0, 200)
    assert bbox.tolist() == [-1, -1, -1, -1, 0]
    ctx.close()


def test_observe_batch_equals_single_calls(gpu_required):
    """fs_observe_batch (one set of host round trips for all episodes) returns, per episode, exactly what fs_observe does:
    episodes whose labelling converges after different numbers of rounds, one without any cloth in view."""
    from flingbot_amd import sim as fsim

    n = 5
    ctx = fsim.FlingSim(n_envs=n, solver=0)
    dims = [(40, 30), (64, 64), (24, 50), (32, 32), (48, 20)]
    for e in range(n):
        env = _scene(ctx, e, dims[e], seed=10 + e, crumple=e != 3)
        cp = ctx.get_camera_params(e)
        ctx.set_camera_params(e, [*cp[2:8], 360, 360])
    ctx.step(20)
    q = ctx.get_positions(3).reshape(-1, 4).copy()
    q[:, 0] += 50.0   # episode 3: the cloth is out of view
    ctx.set_positions(3, q.ravel())
    singles = [ctx.observe(e, 160, want_mask=True) for e in range(n)]
    order = [4, 0, 3, 1, 2]
    obs, bbox, mask = ctx.observe_batch(order, 160, want_mask=True)
    assert tuple(obs.shape) == (n, 4, 160, 160) and bbox.shape == (n, 5)
    for k, e in enumerate(order):
        o1, b1, m1 = singles[e]
        assert np.array_equal(obs[k].cpu().numpy(), o1.cpu().numpy()), e
        assert bbox[k].tolist() == b1.tolist(), e
        assert np.array_equal(mask[k].cpu().numpy(), m1.cpu().numpy()), e
    assert bbox[2].tolist() == [-1, -1, -1, -1, 0] and (bbox[[0, 1, 3, 4], 4] > 0).all()
    ctx.close()
