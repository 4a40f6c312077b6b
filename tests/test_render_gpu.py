"""GPU tests of the observation side of the hot path: coverage reward, vertex normals, rasteriser."""
import os

import numpy as np
import pytest

from conftest import cloth_params

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _grid_for(n):
    """A (dimx, dimz) grid with exactly n particles."""
    for dx in range(int(np.sqrt(n)), 0, -1):
        if n % dx == 0:
            return dx, n // dx
    raise AssertionError


def test_coverage_matches_reference_vectors(gpu_required):
    """fs_coverage == the reference's get_current_covered_area on its own golden vectors (exact float64)."""
    from flingbot_amd import sim as fsim
    from oracle.coverage import covered_area

    g = np.load(os.path.join(GOLD, "coverage_golden.npz"))
    names = [k[4:] for k in g.files if k.startswith("pos_")]
    ctx = fsim.FlingSim(n_envs=len(names) + 1)  # last env has no scene -> 0
    for e, k in enumerate(names):
        pos = g["pos_" + k]
        ctx.set_scene(e, cloth_params(*_grid_for(pos.shape[0])))
        ctx.set_positions(e, pos.ravel())
    cov = ctx.coverage()
    for e, k in enumerate(names):
        ref = float(g["area_" + k])
        assert covered_area(g["pos_" + k].ravel()) == ref  # oracle pinned to the reference vector
        assert cov[e] == ref, (k, cov[e], ref)
    assert cov[-1] == 0.0


def test_coverage_after_simulation(gpu_required):
    from flingbot_amd import sim as fsim
    from oracle.coverage import covered_area
    import scenarios as sc

    ctx = fsim.FlingSim(n_envs=2)
    sc.scenario_crumple(ctx.env(0), 32, 32, seed=5, lift_steps=20, settle_steps=30)
    sc.scenario_drop(ctx.env(1), 24, 40, steps=30)
    cov = ctx.coverage()
    for e in range(2):
        assert cov[e] == covered_area(ctx.get_positions(e))


def test_vertex_normals_match_oracle(gpu_required):
    from flingbot_amd import sim as fsim
    from oracle import OracleSim
    import scenarios as sc

    ctx, orc = fsim.FlingSim(n_envs=1), OracleSim()
    sc.scenario_crumple(ctx.env(0), 32, 32, seed=2, lift_steps=20, settle_steps=10)
    sc.scenario_crumple(orc, 32, 32, seed=2, lift_steps=20, settle_steps=10)
    nh, no = ctx.get_normals(0), orc.get_normals()
    assert np.abs(np.linalg.norm(nh.reshape(-1, 4)[:, :3], axis=1) - 1.0).max() < 1e-5
    assert np.array_equal(nh.view(np.uint32), no.view(np.uint32)), np.abs(nh - no).max()


def test_render_flat_cloth_geometry(gpu_required):
    """Geometric pins of the rasteriser: flat cloth at height h under the top-down FlingBot camera
    (pos (0,2,0), fov 39.5978 deg): depth = 2 - h inside the projected rectangle, ~2.0 on the ground; the cloth covers
    (side * 720 / (2 * d * tan(fov/2)))^2 pixels; rows are bottom-up, world +x -> -y_ndc, +z -> -x_ndc."""
    from flingbot_amd import sim as fsim
    import scenarios as sc

    ctx = fsim.FlingSim(n_envs=1)
    env = ctx.env(0)
    env.set_scene(cloth_params(64, 64, pos=(0.0, 2.0, 0.0)))
    h = 0.05
    w = env.get_positions().reshape(-1, 4)[0, 3]
    p = sc.flat_positions(64, 64, y=h, inv_mass=w)
    p[:, 0] += 0.1  # shift +x to check the axis convention
    env.set_positions(p.ravel())
    rgba, depth = env.render()
    assert rgba.shape == (720 * 720 * 4,) and depth.shape == (720 * 720,)
    img = rgba.reshape(720, 720, 4)
    d = depth.reshape(720, 720)
    ground = np.abs(d - 2.0) < 1e-4
    cloth = np.abs(d - (2.0 - h)) < 1e-4
    assert ground.sum() + cloth.sum() == 720 * 720
    side = 63 * 0.00625
    px_per_m = 720 / (2 * (2.0 - h) * np.tan(np.radians(39.5978 / 2)))
    expect = (side * px_per_m) ** 2
    assert abs(cloth.sum() - expect) / expect < 0.02, (cloth.sum(), expect)
    rows, cols = np.nonzero(cloth)
    # world +x maps to -y_ndc: the cloth (shifted to +x) sits BELOW the image centre in the bottom-up buffer
    assert rows.mean() < 359.5 - 0.1 * px_per_m * 0.8
    assert abs(cols.mean() - 359.5) < 1.5
    # colours: cloth is the pink g_colors[4] * 1.5, ground is the near-black plane; alpha 255 everywhere
    c = img[cloth][:, :3].astype(float).mean(0)
    assert c[0] > c[2] > c[1] and c[0] > 100, c
    gcol = img[ground][:, :3].astype(float).mean(0)
    assert gcol.max() < 40, gcol
    assert (img[..., 3] == 255).all()
    # the cloth throws a shadow: some ground pixels are darker than the lit ground
    gl = img[ground][:, 0].astype(int)
    assert gl.min() < gl.max()


def test_render_pickers_visible_and_idempotent(gpu_required):
    from flingbot_amd import sim as fsim
    import scenarios as sc

    ctx = fsim.FlingSim(n_envs=1)
    env = ctx.env(0)
    env.set_scene(cloth_params(32, 32, pos=(0.0, 2.0, 0.0)))
    w = env.get_positions().reshape(-1, 4)[0, 3]
    env.set_positions(sc.flat_positions(32, 32, y=0.01, inv_mass=w).ravel())
    env.add_sphere(0.02, [0.4, 0.5, -0.4], [1, 0, 0, 0])
    env.add_sphere(0.02, [-0.4, 0.5, -0.4], [1, 0, 0, 0])
    before = env.get_positions().copy()
    rgba1, depth1 = env.render()
    rgba2, depth2 = env.render()
    assert np.array_equal(rgba1, rgba2) and np.array_equal(depth1, depth2)
    assert np.array_equal(before, env.get_positions())  # render never ticks the solver (pyflex.cpp:1079-1087)
    d = depth1.reshape(720, 720)
    sph = (d > 1.47) & (d < 1.53)
    # two discs of radius 0.02 m at distance 1.5: r_px = 0.02 * 720 / (2 * 1.5 * tan(fov/2))
    r_px = 0.02 * 720 / (2 * 1.5 * np.tan(np.radians(39.5978 / 2)))
    assert abs(sph.sum() - 2 * np.pi * r_px ** 2) / (2 * np.pi * r_px ** 2) < 0.15, sph.sum()
    img = rgba1.reshape(720, 720, 4)
    assert img[sph][:, :3].mean() > 120  # 0.9 grey spheres


def test_sphere_mesh_equals_reference_mesh(gpu_required):
    """The picker meshes the rasteriser draws == the reference's own CreateSphere(20, 20, r) + Mesh::Transform
    (core/mesh.cpp:858-902, 650-657; main.cpp:1739-1751), recorded from the reference's mesh.cpp compiled here
    (oracle/_ref/sphere_ref -> tests/golden/sphere_golden.json): 2 400 indices, 441 positions and 441 normals, all exact.
    The mesh follows the shape's PREVIOUS position and rotation; FlingBot's pickers carry the quaternion [1, 0, 0, 0]
    (flex_utils.py:82-83), so their vertex 0 is the south pole."""
    import json

    from flingbot_amd import sim as fsim
    from oracle.render import sphere_mesh as orc_sphere_mesh

    with open(os.path.join(GOLD, "sphere_golden.json")) as fh:
        gold = json.load(fh)
    ctx = fsim.FlingSim(n_envs=1)
    env = ctx.env(0)
    env.set_scene(cloth_params(8, 8, pos=(0.0, 2.0, 0.0)))
    for c in gold:  # current transform = something else: only the previous one may show
        env.add_sphere(c["radius"], [0.9, 0.8, 0.7], [0.0, 0.0, 0.0, 1.0])
    st = env.get_shape_states().reshape(-1, 14).copy()
    for q, c in enumerate(gold):
        st[q, 3:6], st[q, 10:14] = c["pos"], c["quat"]
    env.set_shape_states(st.ravel())
    verts, nrms, tris = env.sphere_mesh()
    o_verts, o_nrms, o_tris = orc_sphere_mesh(env.get_shape_states(), [c["radius"] for c in gold])
    for q, c in enumerate(gold):
        gp = np.array(c["positions"], np.float32).reshape(441, 3)
        gn = np.array(c["normals"], np.float32).reshape(441, 3)
        gi = np.array(c["indices"], np.int32).reshape(800, 3)
        sl = slice(441 * q, 441 * (q + 1))
        assert np.array_equal(tris[800 * q:800 * (q + 1)] - 441 * q, gi)
        assert np.array_equal(verts[sl, :3], gp), np.abs(verts[sl, :3] - gp).max()
        assert np.array_equal(nrms[sl, :3], gn), np.abs(nrms[sl, :3] - gn).max()
        assert (verts[sl, 3] == 1.0).all() and (nrms[sl, 3] == 0.0).all()
    assert np.array_equal(verts, o_verts) and np.array_equal(nrms, o_nrms) and np.array_equal(tris, o_tris)
    # the pickers exactly as Picker.reset adds them: add_sphere(radius, pos, [1, 0, 0, 0]) -> half a turn about x
    env.clear_shapes()
    env.add_sphere(gold[0]["radius"], gold[0]["pos"], gold[0]["quat"])
    verts, nrms, _ = env.sphere_mesh()
    assert np.array_equal(verts[:, :3], np.array(gold[0]["positions"], np.float32).reshape(441, 3))
    assert nrms[0, 1] == -1.0  # south pole first


@pytest.mark.parametrize("side,seed", [(32, 7), (64, 3)])
def test_render_matches_raster_oracle(gpu_required, side, seed):
    """HIP rasteriser vs the scalar C restatement (oracle/raster_oracle.c) on a crumpled cloth with both pickers in view:
    depth bit-exact, alpha exact, colour within 1 LSB (device expf/powf vs libm differ in the last ulp).  side = 64 is
    BASELINE.json configs[1] ("64x64 cloth + depth render") at its own size."""
    from flingbot_amd import sim as fsim
    from oracle.render import render as orc_render
    import scenarios as sc

    ctx = fsim.FlingSim(n_envs=1)
    env = ctx.env(0)
    sc.scenario_crumple(env, side, side, seed=seed, lift_steps=25, settle_steps=20)
    env.add_sphere(0.02, [0.35, 0.4, -0.3], [1, 0, 0, 0])
    env.add_sphere(0.02, [-0.2, 0.1, 0.25], [1, 0, 0, 0])
    st = env.get_shape_states().reshape(-1, 14).copy()
    st[:, 3:6] = st[:, 0:3] + [0.01, 0.0, -0.02]  # prev != current: the renderer must draw the PREVIOUS position
    env.set_shape_states(st.ravel())
    rgba, depth = env.render()
    cam = env.get_camera_params()  # [w, h, px, py, pz, ax, ay, az]
    lo, up = env.get_scene_bounds()
    m = fsim.camera_matrices(cam[2:5], cam[5:8], int(cam[0]), int(cam[1]), lo, up)
    mats = np.concatenate([m["view"].ravel(), m["proj"].ravel(), m["light"].ravel(), m["lightpos"], m["lightdir"]])
    ref_rgba, ref_depth = orc_render(mats, cam[2:5], int(cam[0]), int(cam[1]), env.get_positions(), env.get_normals(),
                                     env.get_faces(), env.get_shape_states(), [0.02, 0.02])
    assert np.array_equal(depth.view(np.uint32), ref_depth.view(np.uint32)), \
        f"{(depth != ref_depth).sum()} depth pixels differ"
    a, b = rgba.reshape(-1, 4).astype(int), ref_rgba.reshape(-1, 4).astype(int)
    assert np.array_equal(a[:, 3], b[:, 3])
    diff = np.abs(a[:, :3] - b[:, :3])
    assert diff.max() <= 1, f"max colour difference {diff.max()}"
    assert (diff > 0).mean() < 0.02
    # ... and with NOTHING of the product on the checker's side: the camera / light matrices printed by the reference's own
    # core/maths.cpp (oracle/_ref/camera_ref, compiled by oracle/Makefile where the reference lies; the binary travels) and the
    # vertex normals the oracle computes from the positions (main.cpp:904-919).  The product's matrices agree with the
    # reference's to 2e-6, so a handful of silhouette pixels may land on the other side of an edge: tolerances, not bits -- but a
    # wrong camera, light or normal would move every pixel.
    import os
    import subprocess
    from oracle import OracleSim
    from conftest import cloth_params

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "camera_ref")
    if os.path.exists(exe):
        out = subprocess.run([exe] + [repr(float(v)) for v in (*cam[2:8], cam[0], cam[1], *lo, *up)], capture_output=True, text=True,
                             check=True).stdout
        rows = {ln.split()[0]: np.array(ln.split()[1:], np.float64).astype(np.float32) for ln in out.splitlines() if ln.strip()}
        ref_mats = np.concatenate([rows["view"], rows["proj"], rows["light"], rows["lightpos"], rows["lightdir"]])
        assert np.abs(ref_mats - mats).max() <= 2e-5 * max(1.0, float(np.abs(ref_mats).max()))
        orc = OracleSim()
        orc.set_scene(cloth_params(side, side, pos=(0.0, -0.2, 0.0)))
        orc.set_positions(env.get_positions())
        ind_rgba, ind_depth = orc_render(ref_mats, cam[2:5], int(cam[0]), int(cam[1]), orc.get_positions(), orc.get_normals(),
                                         orc.get_faces(), env.get_shape_states(), [0.02, 0.02])
        dd = np.abs(depth - ind_depth)
        assert (dd <= 2e-5).mean() >= 0.999, f"{(dd > 2e-5).sum()} depth pixels off against the independent render"
        c2 = np.abs(a[:, :3] - ind_rgba.reshape(-1, 4).astype(int)[:, :3]).max(1)
        assert (c2 <= 2).mean() >= 0.995, f"{(c2 > 2).sum()} colour pixels off against the independent render"
    # the scene really contains all three kinds of primitives
    d = depth.reshape(720, 720)
    assert ((d > 1.55) & (d < 1.65)).sum() > 50 and ((d > 1.85) & (d < 1.95)).sum() > 50  # spheres at y = 0.4 and 0.1
    assert (d < 1.999).sum() > 1500 * (side // 32) ** 2
