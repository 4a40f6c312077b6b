"""CPU tests (no GPU): the oracle against known answers, golden vectors and physics invariants; the product's host
logic (scene builder, camera set-up) against the oracle / reference vectors; the C-ABI surface.

The solver arithmetic of the reference is closed source and absent, so the solver oracle is
"parity unpinned" against PyFleX positions; what IS pinned here: bit-exact integer topology against closed-form counts
and a hand-derived small grid, the exact parameter table, coverage against the reference's own function, camera
matrices against the reference's own maths.h, and the invariants listed in SURVEY.md 8(c).
"""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import cloth_params

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(dimx, dimz, **kw):
    from oracle import OracleSim

    s = OracleSim()
    s.set_scene(cloth_params(dimx, dimz, **kw))
    return s


# ---------------------------------------------------------------- topology (SURVEY.md 8a-4 + table)
@pytest.mark.parametrize("dx,dz,n,stretch,bend,shear,t", [(32, 32, 1024, 1984, 1920, 1922, 1922),
                                                          (64, 64, 4096, 8064, 7936, 7938, 7938)])
def test_grid_counts_known_answers(dx, dz, n, stretch, bend, shear, t):
    s = _oracle(dx, dz, stiff=(0.8, 1.0, 0.9))
    assert (s.n, s.m, s.t) == (n, stretch + bend + shear, t)
    k = s.get_spring_stiffness()
    assert (np.isclose(k, 0.8).sum(), np.isclose(k, 1.0).sum(), np.isclose(k, 0.9).sum()) == (stretch, bend, shear)


def test_grid_3x2_hand_derived():
    """helpers.h:838-924 walked by hand for dx=3, dy=2."""
    s = _oracle(3, 2, pos=(0.0, 0.0, 0.0), stiff=(0.8, 1.0, 0.9))
    # row-major pass y=0: x=1 stretch(1,0); x=2 stretch(2,1) bend(2,0)
    #                y=1: x=0 shear(3,1); x=1 stretch(4,3) shear(4,2) shear(4,0); x=2 stretch(5,4) bend(5,3) shear(5,1)
    # column-major pass: x=0 y=1 stretch(3,0); x=1 y=1 stretch(4,1); x=2 y=1 stretch(5,2)
    expect = [1, 0, 2, 1, 2, 0, 3, 1, 4, 3, 4, 2, 4, 0, 5, 4, 5, 3, 5, 1, 3, 0, 4, 1, 5, 2]
    assert s.get_edges().tolist() == expect
    assert s.get_faces().tolist() == [0, 1, 4, 0, 4, 3, 1, 2, 5, 1, 5, 4]
    k = s.get_spring_stiffness()
    assert np.allclose(k, [0.8, 0.8, 1.0, 0.9, 0.8, 0.9, 0.9, 0.8, 1.0, 0.9, 0.8, 0.8, 0.8])
    r = np.float32(0.00625)
    L = s.get_spring_lengths()
    assert L[0] == r and L[2] == np.float32(2) * r and abs(L[3] - np.sqrt(2) * 0.00625) < 1e-9
    p = s.get_positions().reshape(-1, 4)
    assert np.array_equal(p[:, 0], np.array([0, r, 2 * r, 0, r, 2 * r], np.float32))
    assert np.array_equal(p[:, 2], np.array([0, 0, 0, r, r, r], np.float32))
    assert (p[:, 3] == np.float32(1.0) / (np.float32(0.5) / np.float32(6))).all()
    assert (s.get_phases() == 0x7f300000).all()


def test_parameter_table():
    """Effective NvFlexParams of the cloth scene."""
    p = _oracle(8, 8).get_params()
    f = np.float32
    expect = {0: 30, 1: 4, 2: f(1) / f(100), 3: 0, 4: f(-9.8), 5: 0, 6: f(0.00625) * f(1.8), 7: f(0.00625) * f(1.8),
              8: f(0.005), 9: f(0.04), 10: 0, 11: f(0.75), 12: 0, 13: 1, 14: 1, 15: f(0.02), 16: 1, 17: 100,
              18: np.finfo(np.float32).max, 19: 0, 20: 0, 21: 0, 22: 1, 23: 0, 24: 1, 25: 0, 26: 0, 27: 96, 28: 6, 29: 1}
    for k, v in expect.items():
        assert p[k] == f(v), (k, p[k], v)


def test_cloth_pos_y_is_negated_and_bounds():
    s = _oracle(4, 4, pos=(0.25, 1.5, -0.5))
    p = s.get_positions().reshape(-1, 4)
    assert p[0, 0] == np.float32(0.25) and p[0, 1] == np.float32(-1.5) and p[0, 2] == np.float32(-0.5)
    lo, up = s.get_scene_bounds()  # (-1,1) box merged with the particles, grown by collisionDistance
    assert np.allclose(lo, [-1.005, -1.505, -1.005]) and np.allclose(up, [1.005, 1.005, 1.005])


def test_flip_mesh_swaps_winding_only():
    from oracle import OracleSim

    a, b = OracleSim(), OracleSim()
    a.set_scene(cloth_params(17, 9, flip=0))
    b.set_scene(cloth_params(17, 9, flip=1))
    assert np.array_equal(a.get_edges(), b.get_edges())
    fa, fb = a.get_faces().reshape(-1, 3), b.get_faces().reshape(-1, 3)
    assert not np.array_equal(fa, fb)
    assert np.array_equal(np.sort(fa, axis=1), np.sort(fb, axis=1))


# ---------------------------------------------------------------- product host logic vs oracle (bit-exact)
@pytest.mark.parametrize("dims", [(32, 32), (64, 64), (7, 3), (1, 5), (1, 1)])
def test_host_scene_builder_matches_oracle_bitwise(dims):
    from flingbot_amd import sim as fsim

    p = cloth_params(*dims, pos=(0.3, 1.7, -0.2), stiff=(0.8, 1.0, 0.9), mass=0.37, flip=1 if dims[0] > 16 else 0)
    h = fsim.host_scene(p)
    o = _oracle(*dims, pos=(0.3, 1.7, -0.2), stiff=(0.8, 1.0, 0.9), mass=0.37, flip=1 if dims[0] > 16 else 0)
    assert (h["n"], h["m"], h["t"]) == (o.n, o.m, o.t)
    assert np.array_equal(h["springs"], o.get_edges())
    assert np.array_equal(h["triangles"], o.get_faces())
    assert np.array_equal(h["phases"], o.get_phases())
    for a, b in ((h["positions"], o.get_positions()), (h["spring_lengths"], o.get_spring_lengths()),
                 (h["spring_stiffness"], o.get_spring_stiffness()), (h["params"], o.get_params())):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    lo, up = o.get_scene_bounds()
    assert np.array_equal(h["bounds"], np.concatenate([lo, up]))
    assert h["max_deg"] <= 12
    # CSR adjacency: ascending spring id per particle
    e = h["springs"].reshape(-1, 2)
    for i in range(min(h["n"], 50)):
        ids = np.nonzero((e == i).any(axis=1))[0]
        other = np.where(e[ids, 0] == i, e[ids, 1], e[ids, 0])
        assert np.array_equal(h["adj_neighbors"][h["adj_offsets"][i]:h["adj_offsets"][i + 1]], other)


def test_host_scene_mesh_path_and_errors():
    from flingbot_amd import sim as fsim
    from oracle import OracleSim

    verts = np.array([[0, 0, 0], [0.1, 0, 0], [0, 0, 0.1], [0.1, 0.02, 0.1]], np.float32)
    faces, stretch, bend, shear = [0, 1, 2, 1, 3, 2], [0, 1, 0, 2, 1, 3, 2, 3], [0, 3], [1, 2]
    p = cloth_params(0, 0, pos=(0.1, 0.3, 0.2), mass=0.01)
    h = fsim.host_scene(p, verts.ravel(), stretch, bend, shear, faces)
    o = OracleSim()
    o.set_scene(p, verts.ravel(), stretch, bend, shear, faces)
    assert h["n"] == 4 and h["m"] == 6 and h["t"] == 2
    assert np.array_equal(h["positions"].view(np.uint32), o.get_positions().view(np.uint32))
    assert np.array_equal(h["spring_lengths"].view(np.uint32), o.get_spring_lengths().view(np.uint32))
    assert np.array_equal(h["springs"], o.get_edges())
    with pytest.raises(fsim.FlingSimError):
        fsim.host_scene(p, verts.ravel(), stretch, bend, shear, [0, 1, 9])  # face index out of range
    with pytest.raises(fsim.FlingSimError):
        fsim.host_scene(p[:10])  # too few scene params


def test_camera_matrices_match_reference_maths():
    """fs_camera_matrices vs matrices printed by oracle/_ref/camera_ref (built from the reference's core/maths.h)."""
    from flingbot_amd import sim as fsim

    with open(os.path.join(GOLD, "camera_golden.json")) as fh:
        cases = json.load(fh)
    assert len(cases) >= 3
    for c in cases:
        m = fsim.camera_matrices(c["cam"], c["ang"], c["w"], c["h"], c["lo"], c["up"])
        for key in ("view", "proj", "light"):
            ref = np.array(c[key], np.float32).reshape(4, 4)
            assert np.allclose(m[key], ref, rtol=2e-6, atol=2e-5), (key, np.abs(m[key] - ref).max())
        assert np.allclose(m["lightpos"], c["lightpos"], rtol=1e-6)
        assert np.allclose(m["lightdir"], c["lightdir"], rtol=1e-6, atol=1e-7)
    # FlingBot's top-down camera: world +x -> -y_ndc, +z -> -x_ndc, ground at eye z = -2
    c = cases[0]
    vp = np.array(c["proj"]).reshape(4, 4) @ np.array(c["view"]).reshape(4, 4)
    for world, sx, sy in (([0.1, 0, 0], 0, -1), ([0, 0, 0.1], -1, 0)):
        clip = vp @ np.array(world + [1.0])
        ndc = clip[:3] / clip[3]
        assert np.sign(round(ndc[0], 6)) == sx and np.sign(round(ndc[1], 6)) == sy
    assert abs((np.array(c["view"]).reshape(4, 4) @ np.array([0, 0, 0, 1.0]))[2] + 2.0) < 1e-5


# ---------------------------------------------------------------- coverage (reference's own function)
def test_coverage_oracle_matches_reference_vectors():
    from oracle.coverage import covered_area

    g = np.load(os.path.join(GOLD, "coverage_golden.npz"))
    names = [k[4:] for k in g.files if k.startswith("pos_")]
    assert len(names) >= 7
    for k in names:
        assert covered_area(g["pos_" + k].ravel().copy()) == float(g["area_" + k]), k
    # SURVEY.md [probed]: flat 64x64 -> 0.16, 32x32 -> 0.04 (to the discretisation)
    assert abs(float(g["area_flat64"]) - 0.16) < 0.006 and abs(float(g["area_flat32"]) - 0.04) < 0.003


# ---------------------------------------------------------------- solver invariants (SURVEY.md 8c iii)
def test_free_fall_matches_closed_form():
    """One unconstrained particle: v_{k+1} = v_k + h (g - damping v_k), x_{k+1} = x_k + h v_{k+1}."""
    s = _oracle(1, 1, pos=(0.0, -1.0, 0.0))
    s.step(10)
    h, g, d = np.float32(0.01) / np.float32(4), np.float32(-9.8), np.float32(1.0)
    x, v = np.float32(1.0), np.float32(0.0)
    for _ in range(40):
        v = np.float32(v + h * (g - d * v))
        xn = np.float32(x + h * v)
        v = np.float32((xn - x) * (np.float32(1.0) / h))  # velocity re-derived from the displacement (finalize)
        x = xn
    p, vel = s.get_positions(), s.get_velocities()
    assert abs(p[1] - x) < 1e-6 and abs(vel[1] - v) < 1e-4
    assert p[0] == 0.0 and p[2] == 0.0


def test_pinned_particle_never_moves_and_cloth_hangs():
    s = _oracle(8, 8, pos=(0.0, -0.5, 0.0))
    p = s.get_positions().reshape(-1, 4).copy()
    p[0, 3] = 0.0
    p[7, 3] = 0.0
    s.set_positions(p.ravel())
    s.step(60)
    q = s.get_positions().reshape(-1, 4)
    assert np.array_equal(q[[0, 7], :3], p[[0, 7], :3])
    assert q[-1, 1] < 0.5 - 0.02  # the free edge sags
    assert np.isfinite(q).all()
    assert (s.get_velocities().reshape(-1, 3)[[0, 7]] == 0).all()


def test_ground_contact_rest_and_sleep():
    """A cloth dropped onto the plane ends at y ~= collisionDistance, asleep (v == 0), and then does not move at all."""
    s = _oracle(16, 16, pos=(0.0, -0.03, 0.0))
    s.step(80)
    p1 = s.get_positions().copy()
    y = p1.reshape(-1, 4)[:, 1]
    assert y.min() > 0.005 - 3e-4 and y.max() < 0.005 + 3e-4
    assert np.abs(s.get_velocities()).max() == 0.0
    s.step(20)
    assert np.array_equal(p1, s.get_positions())


def test_spring_pair_converges_to_rest_length():
    """Two free particles + one spring (mesh path), stretched: the distance relaxes towards the rest length."""
    from oracle import OracleSim

    s = OracleSim()
    verts = np.array([[0, 0, 0], [0.01, 0, 0]], np.float32)
    s.set_scene(cloth_params(0, 0, pos=(0.0, -1.0, 0.0), stiff=(0.9, 0.9, 0.9), mass=0.001), verts.ravel(), [0, 1], [], [], [])
    L = s.get_spring_lengths()[0]
    p = s.get_positions().reshape(-1, 4).copy()
    p[1, 0] = 0.02
    s.set_positions(p.ravel())
    s.step(1)
    q = s.get_positions().reshape(-1, 4)
    d = np.linalg.norm(q[1, :3] - q[0, :3])
    assert abs(d - L) < 1e-4 * L + 1e-7  # 120 sweeps at stiffness 0.9 leave nothing of a 100% stretch
    assert abs((q[0, 0] + q[1, 0]) / 2 - 0.01) < 1e-6  # equal masses: the midpoint stays


def test_self_collision_filter_and_separation():
    """Rest-pose neighbours (closer than `radius` in the rest pose) never become contacts; folded layers do and end up
    separated by ~solidRestDistance."""
    import scenarios as sc
    from oracle import OracleSim

    s = OracleSim()
    sc.scenario_crumple(s, 32, 32, seed=3)
    cnt, lists = s.get_last_neighbors()
    assert cnt.sum() > 100 and cnt.max() <= 96
    rest = s.get_restPositions().reshape(-1, 4)[:, :3]
    pos = s.get_positions().reshape(-1, 4)[:, :3]
    r = 0.00625 * 1.8
    for i in np.nonzero(cnt)[0][:200]:
        js = lists[i, :cnt[i]]
        assert (np.diff(js) > 0).all()  # ascending, unique
        assert (np.linalg.norm(rest[js] - rest[i], axis=1) >= r - 1e-9).all()
        assert np.linalg.norm(pos[js] - pos[i], axis=1).min() > 0.5 * r  # no deep interpenetration at rest
    assert pos[:, 1].min() > 0.005 - 1.5e-3


def test_sphere_pushes_cloth_and_friction_drags_it():
    """A kinematic sphere sweeping through a resting cloth displaces particles (contact) along its motion (friction)."""
    import scenarios as sc
    from oracle import OracleSim

    s = OracleSim()
    s.set_scene(cloth_params(16, 16, pos=(0.0, -0.2, 0.0)))
    w = s.get_positions().reshape(-1, 4)[0, 3]
    s.set_positions(sc.flat_positions(16, 16, y=0.005, inv_mass=w).ravel())
    s.add_sphere(0.02, [-0.1, 0.02, 0.0], [1, 0, 0, 0])
    before = s.get_positions().reshape(-1, 4)[:, 0].mean()
    for k in range(40):
        st = s.get_shape_states().reshape(-1, 14).copy()
        st[:, 3:6] = st[:, 0:3]
        st[:, 0] += 0.004
        s.set_shape_states(st.ravel())
        s.step()
    after = s.get_positions().reshape(-1, 4)
    assert after[:, 0].mean() > before + 0.002
    c = s.get_shape_states().reshape(-1, 14)[0, :3]
    d = np.linalg.norm(after[:, :3] - c, axis=1)
    assert d.min() > 0.02  # nothing left inside the sphere radius


def test_oracle_is_deterministic():
    import scenarios as sc
    from oracle import OracleSim

    a, b = OracleSim(), OracleSim()
    sc.scenario_fling(a, 16, 16, settle_steps=10)
    sc.scenario_fling(b, 16, 16, settle_steps=10)
    assert np.array_equal(a.get_positions().view(np.uint32), b.get_positions().view(np.uint32))


def test_rsqrt_table_is_within_one_ulp_and_rescales():
    """The oracle's reciprocal square root is data measured on an MI355X (oracle/v_rsq_f32_gfx950.npz: gfx950's v_rsq_f32).
    What can be checked without the chip: every entry is -1 / 0 / +1 ulp from float32(1 / sqrt(float64(x))), the function is
    within 1 ulp of the exact reciprocal root over every normal exponent (AMD documents 1 ulp for the instruction), zero and
    denormal inputs take the FLT_MIN clamp (2^63, finite), inf gives 0, and the C implementation equals a numpy
    restatement of table + rescaling.  (tests/test_parity_gpu.py re-reads the whole table from the GPU it runs on.)"""
    import oracle

    tab = oracle.rsqrt_table()
    fields = np.stack([(tab >> s) & 3 for s in (0, 2, 4, 6)], 1).ravel().astype(np.int64) - 2
    assert fields.size == 1 << 24 and set(np.unique(fields)) <= {-1, 0, 1}
    assert (fields == 0).mean() > 0.8
    rng = np.random.RandomState(3)
    xb = rng.randint(1 << 23, 255 << 23, size=1 << 18).astype(np.uint32)
    x = xb.view(np.float32)
    y = oracle.eval_rsqrt(x)
    exact = 1.0 / np.sqrt(x.astype(np.float64))
    assert (np.abs(y.astype(np.float64) - exact) / np.spacing(exact.astype(np.float32)).astype(np.float64)).max() <= 1.0
    e = (xb >> 23).astype(np.int64) - 127
    par = e & 1
    xr = (((127 + par) << 23) | (xb & 0x7fffff)).astype(np.uint32).view(np.float32)
    base = ((1.0 / np.sqrt(xr.astype(np.float64))).astype(np.float32).view(np.int32) + fields[(par << 23) | (xb & 0x7fffff)]).astype(np.int32)
    pred = np.ldexp(base.view(np.float32), (-(e - par) // 2).astype(np.int32)).astype(np.float32)
    assert np.array_equal(pred.view(np.uint32), y.view(np.uint32))
    sp = oracle.eval_rsqrt(np.array([0.0, 1e-45, 1e-40, 1.17549435e-38, np.inf, 4.0], np.float32))
    assert list(sp) == [2.0 ** 63] * 4 + [0.0, 0.5]


def test_static_friction_branch_is_redundant():
    """flex_oracle.c friction_scale: 'full stick below mu_s x depth, else clamp to mu_k x depth'.  With mu_s <= mu_k -- the
    cloth scene has (0, 0.75) against shapes and (1, 1) between particles -- the static branch can never change a result:
    whenever it fires the kinetic clamp would have returned 1 as well.  So 'is there a separate static branch in FleX' is
    not a parity risk for this scene (PARITY.md)."""
    rng = np.random.RandomState(0)
    tl = np.concatenate([rng.rand(20000) * 0.01, [0.0]])
    pen = np.concatenate([rng.rand(20000) * 0.01, [0.0]])
    for mu_s, mu_k in ((0.0, 0.75), (1.0, 1.0), (0.3, 0.75)):
        with_static = np.where(tl < mu_s * pen, 1.0, np.where(tl > mu_k * pen, mu_k * pen / np.maximum(tl, 1e-30), 1.0))
        without = np.where(tl > mu_k * pen, mu_k * pen / np.maximum(tl, 1e-30), 1.0)
        assert np.array_equal(with_static, without)


def test_model_alternatives_are_live_and_bounded(capsys):
    """oracle/flex_oracle.c "MODEL switches": every alternative reading of an inferred choice builds, runs the canonical
    workloads, and changes what it claims to change and nothing else -- the by-distance truncation and the end-pose spheres
    are exact no-ops where lists stay below 96 / no sphere exists, every other switch moves the trajectory.  The full table
    (64 x 64, all frames) is PARITY.md, made by tests/parity_table.py."""
    import parity_table as pt

    out = pt.table(quick=True, jobs=8, scenarios=("c2", "fling"))
    for s in ("c2", "fling"):
        assert out[(s, "alt_neighbors_by_distance")]["divergence"] == [0.0] * len(out[(s, "exact")]["frames"])
        assert out[(s, "exact")]["max_list"][0] < 96
        assert out[(s, "exact")]["local"][0] <= 1e-6           # arithmetic choices: one step from the same state
        for v in ("alt_friction_post", "alt_apply_per_type", "alt_stiffness_iter", "alt_damping_mult"):
            assert max(out[(s, v)]["divergence"]) > 1e-6, (s, v)
        for v in ("alt_sleep_velocity_only", "alt_sleep_at_predict", "alt_no_sleep"):
            assert max(out[(s, v)]["divergence"]) > 0, (s, v)
    assert max(out[("c2", "alt_shape_end_pose")]["divergence"]) == 0.0       # no spheres in the crumple
    assert max(out[("fling", "alt_shape_end_pose")]["divergence"]) > 1e-6    # pickers move in the fling
    # the finalize clamp's readings (NvFlex.h:112-113; round 6): the clamp DOES fire in both workloads (white-box counter), every
    # reading of it moves the trajectory, dropping it altogether sends the sheet flying at the pop-out of flex_utils.set_scene's
    # step (metres after one frame of the crumple), and no contact normal ever falls back to (0,1,0)
    for s in ("c2", "fling"):
        clamps, degenerate, particle_substeps = out[(s, "exact")]["white"]
        assert 0 < clamps < particle_substeps and degenerate == 0, (s, out[(s, "exact")]["white"])
        for v in ("alt_no_maxaccel", "alt_maxaccel_per_frame", "alt_maxaccel_position"):
            assert max(out[(s, v)]["divergence"]) > 1e-4, (s, v)
        # a picked particle's velocity at finalize cannot matter: the reference grasps a settled cloth and never writes velocities
        assert out[(s, "alt_kinematic_velocity_kept")]["divergence"] == [0.0] * len(out[(s, "exact")]["frames"])
    assert out[("c2", "alt_no_maxaccel")]["divergence"][0] > 1.0
    # every switch of the oracle has its row in the table and its paragraph in PARITY.md
    from oracle.flex import MODEL_ALTERNATIVES
    rows = {v for v, _ in pt.ROWS}
    parity_md = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "PARITY.md")).read()
    for name in MODEL_ALTERNATIVES:
        assert "alt_" + name in rows and "`alt_" + name + "`" in parity_md, name
    with capsys.disabled():
        print("\n" + pt.render(out, scenarios=("c2", "fling")))


# ---------------------------------------------------------------- how far the oracle's spelled-out arithmetic moves a step
def _clone_state(src, variant, params):
    """A fresh oracle of build `variant` in exactly the state of `src` (particles, velocities, phases, shapes)."""
    from oracle import OracleSim

    o = OracleSim(variant)
    o.set_scene(params)
    if src.get_n_shapes():
        for row in src.get_shape_states().reshape(-1, 14):
            o.add_sphere(0.02, row[:3], [1, 0, 0, 0])
        o.set_shape_states(src.get_shape_states())
    o.set_positions(src.get_positions())
    o.set_velocities(src.get_velocities())
    o.set_phases(src.get_phases())
    return o


def _approximation_cases():
    """(name, oracle in a recorded mid-motion state, its scene parameters): the regimes of the hot path."""
    import scenarios as sc
    from oracle import OracleSim

    class _Stop(Exception):
        pass

    def until(n_steps):
        seen = [0]

        def rec(_):
            seen[0] += 1
            if seen[0] >= n_steps:
                raise _Stop
        return rec

    P = cloth_params
    o = OracleSim()
    sc.scenario_drop(o, 32, 32, height=0.3, steps=10)
    yield "free fall", o, P(32, 32, pos=(0, -0.3, 0))
    # tilted sheet sliding into the ground: plane contact + dynamic friction while most of it still moves
    o, p = OracleSim(), P(32, 32, pos=(0, -0.05, 0))
    o.set_scene(p)
    pos = o.get_positions().reshape(-1, 4).copy()
    pos[:, 1] = 0.006 + 0.3 * (pos[:, 0] - pos[:, 0].min())
    o.set_positions(pos.ravel())
    o.set_velocities(np.tile(np.array([0.3, -0.5, 0.1], np.float32), pos.shape[0]))
    o.step(5)
    yield "ground contact with friction", o, p
    o = OracleSim()
    sc.scenario_crumple(o, 32, 32, seed=3, lift_steps=40, settle_steps=12)
    yield "crumple, falling onto itself", o, P(32, 32, pos=(0, -0.2, 0))
    o = OracleSim()
    try:
        sc.scenario_fling(o, 32, 32, record=until(70))
    except _Stop:
        pass
    yield "fling, pickers moving", o, P(32, 32, pos=(0, -0.2, 0))
    o = OracleSim()
    sc.scenario_fling(o, 32, 32, settle_steps=6)
    yield "fling, just released", o, P(32, 32, pos=(0, -0.2, 0))
    o, p = OracleSim(), P(32, 32, pos=(0, 0.3, 0))
    o.set_scene(p)
    rng = np.random.RandomState(5)
    pos = o.get_positions().reshape(-1, 4).copy()
    pos[:, :3] = (rng.rand(1024, 3) * 0.04).astype(np.float32) + np.array([0, 0.3, 0], np.float32)
    o.set_positions(pos.ravel())
    o.step(2)
    yield "dense ball (83 contacts / particle)", o, p


def test_approximations_stay_within_1e_4_of_exact_math(capsys):
    """The solver oracle spells two arithmetic choices out so CPU and GPU agree bit for bit: 1/sqrt as an integer seed +
    three Newton steps (<= 2 ulp) and explicit fused multiply-adds in dot products / accumulations.
    This bounds them: from six recorded mid-motion states, ONE pyflex.step() (4 substeps x 30 iterations) of the oracle
    against the plain IEEE restatement of the same step (oracle/Makefile liboracle_exact.so: correctly rounded sqrtf and
    divisions, every multiply-add rounded twice) and against each choice alone -- north_star's tolerance is 1e-4 relative
    on positions; measured <= 1.3e-6."""
    rows = []
    for name, src, params in _approximation_cases():
        moving = int((src.get_velocities().reshape(-1, 3) != 0).any(1).sum())
        assert moving > 100, f"{name}: the recorded state must be in motion"
        for variant in ("exact", "exact_rsqrt", "nofma", "newton"):
            a, b = _clone_state(src, None, params), _clone_state(src, variant, params)
            a.step(1)
            b.step(1)
            pa, pb = a.get_positions().reshape(-1, 4)[:, :3], b.get_positions().reshape(-1, 4)[:, :3]
            va, vb = a.get_velocities().reshape(-1, 3), b.get_velocities().reshape(-1, 3)
            # The sleep rule (NvFlex.h:110) is a discontinuity of the MODEL: a particle whose speed sits on sleepThreshold
            # = 0.02 in the last substep keeps its position on one side and moves by <= 0.02 x 2.5 ms on the other, whatever
            # the arithmetic.  Such particles are counted and bounded by exactly that; everything else by the tolerances.
            flip = (va == 0).all(1) != (vb == 0).all(1)
            assert flip.sum() <= 2, (name, variant, int(flip.sum()))
            if flip.any():
                assert np.abs(pa - pb)[flip].max() <= 0.02 * 0.0025 * 1.01 and np.abs(va - vb)[flip].max() <= 0.02 * 1.01
            pscale, vscale = max(1.0, float(np.abs(pa).max())), max(1.0, float(np.abs(va).max()))
            pos_all = float(np.abs(pa - pb).max() / pscale)
            pos_rel = float(np.abs(pa - pb)[~flip].max() / pscale)
            vel_rel = float(np.abs(va - vb)[~flip].max() / vscale)
            rows.append((name, variant, pos_rel, vel_rel, int(flip.sum())))
            assert pos_all <= 1e-4, (name, variant, pos_all)      # north_star's bar, every particle
            assert pos_rel <= 2e-6, (name, variant, pos_rel)      # two orders inside it away from the sleep threshold
            # velocities are position differences / 2.5 ms, so they amplify a last-bit position change 400-fold; a
            # stick / slip decision that flips on the packed ball (never seen in a cloth scene) costs 3e-4 there
            assert vel_rel <= (1e-3 if name.startswith("dense ball") else 1e-4), (name, variant, vel_rel)
    with capsys.disabled():
        for r in rows:
            print("\n  one-step effect  %-36s %-12s positions %.2e  velocities %.2e (relative)  sleep flips %d" % r, end="")


# ---------------------------------------------------------------- C-ABI surface
def test_library_exports_every_declared_symbol():
    """libflingsim.so loads and exports every function include/flingsim.h declares (no compute call without a GPU)."""
    from flingbot_amd import build, sim as fsim

    lib = fsim.load_library()
    hdr = open(os.path.join(ROOT, "include", "flingsim.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 45
    raw = C.CDLL(build.LIB)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in flingsim.h but not exported"
    for name in lib._fs_symbols:
        assert name in declared, f"{name} bound in sim.py but not declared in the header"
    assert lib.fs_version() >= 100


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly (this container has no GPU)."""
    import torch
    from flingbot_amd import sim as fsim

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fsim.FlingSimError, match="no HIP device|hip"):
        fsim.FlingSim(n_envs=1)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under flingbot_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "flingbot_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt and "flex_oracle" not in txt, f
    # development helpers under scripts/ are not tests either: the ones that compare against the oracle live in tests/soak/
    for f in os.listdir(os.path.join(ROOT, "scripts")):
        if f.endswith(".py"):
            txt = open(os.path.join(ROOT, "scripts", f), errors="ignore").read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


# ---------------------------------------------------------------- picker / movep restatement vs the reference classes
def replay_picker_program(sim, picker_cls=None, golden=None, record=None):
    """Replays the recorded movep program of tests/golden/picker_golden.npz on `sim` with the restated picker."""
    from oracle.picker import OraclePicker

    g = golden
    sim.set_scene(g["scene_params"])
    sim.step(1)
    sim.set_positions(g["init_pos"].ravel())
    sim.set_velocities(np.zeros(3 * g["init_pos"].shape[0], np.float32))
    tool = (picker_cls or OraclePicker)(sim)
    # Picker.reset(center) puts two pickers at center -/+ r on x (flex_utils.py:64-72): r = sqrt(1) * 0.02 * 2
    r = np.sqrt(2 - 1) * 0.02 * 2.
    tool.reset([[0.0 + np.cos(2 * np.pi * i / 2) * r, 0.1, 0.0 + np.sin(2 * np.pi * i / 2) * r] for i in range(2)])
    iters = []
    for target, speed, ms, gs in zip(g["targets"], g["speeds"], g["min_steps"], g["grasp"]):
        iters.append(tool.movep(target, list(gs), speed=float(speed), min_steps=None if ms < 0 else int(ms)))
        if record is not None:
            record(sim, tool)
    return tool, iters


def test_picker_restatement_matches_reference_classes():
    """oracle/picker.py == the reference's Picker/PickerPickPlace + movep, both driving the CPU oracle: same picked
    particle ids, same shape states and particle arrays bit for bit over 92 simulation steps."""
    from oracle import OracleSim

    g = np.load(os.path.join(GOLD, "picker_golden.npz"))
    sim = OracleSim()
    shapes, picked, poss = [], [], []
    from oracle.picker import OraclePicker

    class Rec(OraclePicker):
        def step(self, action):
            super().step(action)

        def pick_place_step(self, action):
            n = super().pick_place_step(action)
            if n:
                shapes.append(self.sim.get_shape_states().copy())
                picked.append([-1 if q is None else q for q in self.picked_particles])
                poss.append(self.sim.get_positions().copy())
            return n

    tool, iters = replay_picker_program(sim, Rec, g)
    assert iters == g["iters"].tolist()
    assert np.array_equal(np.array(picked), g["picked"])
    assert np.array_equal(np.array(shapes).view(np.uint32), g["shapes"].view(np.uint32))
    assert np.array_equal(np.array(poss[::10]).view(np.uint32), g["pos_every_10"].view(np.uint32))
    assert np.array_equal(poss[-1].view(np.uint32), g["pos_last"].view(np.uint32))
    assert g["picked"][-1].tolist() == [0, -1] and g["picked"][5].tolist() == [0, 23]


def test_fling_primitive_host_logic_reproduces_reference_golden():
    """flingbot_amd.primitives.FlingPrimitives is pure host logic over a simulator interface: on the CPU oracle (with the
    numpy restatement of the picker) it retraces the trajectories the REFERENCE's SimEnv methods produced
    (tests/golden/fling_golden.npz), bit for bit."""
    from fling_helpers import OracleBatch, load_fling_golden
    from flingbot_amd.primitives import FlingPrimitives

    g = load_fling_golden()
    cases = [0, 1, 2, 3]  # two-handed (full stretch loop), one-handed (single-grasp exit), missed grasp (terminate), no grasp
    sim = OracleBatch(len(cases), g["scene_params"], g["init_pos"])
    prim = FlingPrimitives(sim, range(len(cases)))
    out = prim.pick_and_fling(g["p1"][cases], g["p2"][cases], g["g1"][cases], g["g2"][cases])
    for k, c in enumerate(cases):
        assert out[k]["terminated"] == bool(g["terminate"][c])
        if not np.isnan(g["stretch_ret"][c]):
            assert out[k]["dist"] == g["stretch_ret"][c] and out[k]["fling_height"] == g["lift_ret"][c]
        assert np.array_equal(sim.get_positions(k).view(np.uint32), g["pos_fling"][c].view(np.uint32)), c
    assert out[3]["skipped"]


def test_scheduled_fling_programs_reproduce_reference_golden():
    """The same golden through flingbot_amd/schedule.py: every case is its own program (the reference's straight-line code
    as a coroutine), advanced in short resumable chunks next to the others -- including the no-op case -- and followed by
    wait_until_stable inside the same scheduling run for the case the golden recorded it for."""
    from fling_helpers import OracleBatch, load_fling_golden
    from flingbot_amd.primitives import FlingPrimitives

    g = load_fling_golden()
    cases = [0, 1, 2, 3]
    sim = OracleBatch(len(cases), g["scene_params"], g["init_pos"])
    prim = FlingPrimitives(sim, range(len(cases)))
    out, _ = prim.act_scheduled({k: ("fling", g["p1"][c], g["p2"][c], g["g1"][c], g["g2"][c]) for k, c in enumerate(cases)},
                                settle=False, cap_min=5, cap=11)
    assert sim.advance_calls > 60  # the trajectories really were cut into many chunks
    for k, c in enumerate(cases):
        assert out[k]["terminated"] == bool(g["terminate"][c])
        if not np.isnan(g["stretch_ret"][c]):
            assert out[k]["dist"] == g["stretch_ret"][c] and out[k]["fling_height"] == g["lift_ret"][c]
        assert np.array_equal(sim.get_positions(k).view(np.uint32), g["pos_fling"][c].view(np.uint32)), c
    assert out[3]["skipped"] and prim.sim_steps > 0  # (the device test compares the step count with the lock-step run's)


def test_collide_shapes_stage_lists_candidates_once_per_substep_and_never_changes_a_result():
    """NvFlex.h:205 (collideShapes, once per substep) + :145-147 (collisionDistance + shapeCollisionMargin) + :361 / main.cpp:828
    (maxContactsPerParticle = 6): the oracle lists, per particle, the planes / spheres within 0.005 + 0.04 of the predicted
    position and its iterations test only those.  (a) the list is exactly that set, (b) capped at 6 -- plane first, then the
    lowest-numbered spheres, (c) no iteration of a fling ever finds a violated contact outside the list, so (d) the trajectory
    equals the build without the stage (-DORC_ALT_SHAPE_EVERY_ITERATION = rounds 1-4) bit for bit."""
    from oracle import OracleSim

    import scenarios as sc

    # (a) one step of a resting sheet under two spheres: recompute the lists from the state before the step
    s = OracleSim()
    s.set_scene(cloth_params(16, 16, pos=(0.0, -0.03, 0.0)))
    centres = np.array([(0.02, 0.06, 0.03), (0.3, 0.5, 0.3)], np.float64)
    for c in centres:
        s.add_sphere(0.02, c, [1, 0, 0, 0])
    s.set_shape_states(np.array(s.get_shape_states(), np.float32))
    s.step(3)
    pos = s.get_positions().reshape(-1, 4).astype(np.float64)
    m = s.get_last_shape_candidates()
    near_ground = pos[:, 1] < 0.045 - 2e-3            # well inside / outside the reach (the lists are built on the last
    far_ground = pos[:, 1] > 0.045 + 2e-3             # substep's PREDICTED positions, a fraction of a millimetre away)
    assert ((m[near_ground] & 1) == 1).all() and ((m[far_ground] & 1) == 0).all()
    d0 = np.linalg.norm(pos[:, :3] - centres[0], axis=1) - 0.02
    assert ((m[d0 < 0.045 - 2e-3] & 0x100) != 0).all() and ((m[d0 > 0.045 + 2e-3] & 0x100) == 0).all()
    assert (m & 0x100).any() and not (m & 0x200).any() and not (m >> 10).any()
    # (b) the cap
    s = OracleSim()
    s.set_scene(cloth_params(8, 8, pos=(0.0, -0.03, 0.0)))
    for q in range(8):
        s.add_sphere(0.02, (0.02 + 0.002 * q, 0.04, 0.02), [1, 0, 0, 0])
    s.set_shape_states(np.array(s.get_shape_states(), np.float32))
    s.step(1)
    m = s.get_last_shape_candidates()
    assert (m == 0x1f01).all(), np.unique(m)
    assert s.get_params()[28] == 6.0 and abs(s.get_params()[9] - 0.04) < 1e-9
    # (c), (d): a fling with grasped corners
    a, b = OracleSim(), OracleSim("alt_shape_every_iteration")
    ra, rb = [], []
    sc.scenario_fling(a, 16, 16, settle_steps=20, record=lambda sim: ra.append(sim.get_positions().copy()))
    sc.scenario_fling(b, 16, 16, settle_steps=20, record=lambda sim: rb.append(sim.get_positions().copy()))
    assert a.missed_shape_contacts() == 0
    assert len(ra) == len(rb) > 100 and np.array_equal(np.array(ra).view(np.uint32), np.array(rb).view(np.uint32))


def test_action_selection_restatement_matches_reference_golden():
    """oracle/action.py against tests/golden/action_golden.npz -- SimEnv.get_max_value_valid_action of the REFERENCE run
    on synthetic value maps and depth images (fling only / three primitives with tied values / stretchdrag / nothing
    reachable)."""
    from oracle import action as oa

    g = np.load(os.path.join(GOLD, "action_golden.npz"))
    for ci in range(4):
        D, S, gd, dd, pd, _ = g[f"c{ci}_cfg"].tolist()
        reach, sdist, gh = g[f"c{ci}_reach"].tolist()
        cfg = dict(obs_dim=D, pix_grasp_dist=gd, pix_drag_dist=dd, pix_place_dist=pd, scales=g[f"c{ci}_scales"],
                   rotations=g[f"c{ci}_rotations"].tolist(), depth=g[f"c{ci}_depth"], reach_distance_limit=reach,
                   stretchdrag_dist=sdist, grasp_height=gh, left_arm_base=np.array([0.765, 0, 0]),
                   right_arm_base=np.array([-0.765, 0, 0]))
        action, res, _ = oa.get_max_value_valid_action(g[f"c{ci}_values"], g[f"c{ci}_prims"].tolist(), cfg)
        want = str(g[f"c{ci}_action"])
        assert (action or "") == want, ci
        if want:
            assert np.array_equal(res["p1"], g[f"c{ci}_p1"]) and np.array_equal(res["p2"], g[f"c{ci}_p2"]), ci


def test_action_selector_host_evaluation_matches_reference_golden():
    """The product's host-side re-evaluation of the winning candidate (flingbot_amd/action.py, the reference's numpy
    expressions) returns the reference's own p1 / p2 for the golden winners; no GPU involved."""
    from flingbot_amd.action import ActionSelector
    from oracle import action as oa

    g = np.load(os.path.join(GOLD, "action_golden.npz"))
    for ci in range(3):
        D, S, gd, dd, pd, _ = g[f"c{ci}_cfg"].tolist()
        reach, sdist, gh = g[f"c{ci}_reach"].tolist()
        prims = g[f"c{ci}_prims"].tolist()
        rotations = g[f"c{ci}_rotations"].tolist()
        cfg = dict(obs_dim=D, pix_grasp_dist=gd, pix_drag_dist=dd, pix_place_dist=pd, scales=g[f"c{ci}_scales"],
                   rotations=rotations, depth=g[f"c{ci}_depth"], reach_distance_limit=reach, stretchdrag_dist=sdist,
                   grasp_height=gh, left_arm_base=np.array([0.765, 0, 0]), right_arm_base=np.array([-0.765, 0, 0]))
        action, _, k = oa.get_max_value_valid_action(g[f"c{ci}_values"], prims, cfg)
        P, T = g[f"c{ci}_values"].shape[:2]
        pidx, x, yy, zz = np.unravel_index(k, (P, T, D - 2 * gd, D - 2 * gd))
        sel = ActionSelector(prims, rotations, D, gd, dd, pd, reach, stretchdrag_dist=sdist, grasp_height=gh)
        res = sel._candidate(prims[pidx], int(x), int(yy) + gd, int(zz) + gd, g[f"c{ci}_scales"], g[f"c{ci}_depth"])
        assert res is not None and prims[pidx] == str(g[f"c{ci}_action"])
        assert np.array_equal(res["p1"], g[f"c{ci}_p1"]) and np.array_equal(res["p2"], g[f"c{ci}_p2"]), ci


def test_action_host_evaluation_equals_oracle_on_random_candidates():
    """flingbot_amd/action.py formulates the host geometry from the math (its own matrix construction, one vectorised
    back-projection, a table for the grasp pixels); oracle/action.py restates the reference statement by statement.  Over
    random candidates of all four primitives, random depth planes, scales and rotations both must agree on skip / keep and,
    for kept candidates, on every float64 bit of p1 / p2 and on the pre-transform pixels and the chosen arm."""
    from flingbot_amd.action import ActionSelector, get_transform_matrix, get_action_params
    from oracle import action as oa

    rng = np.random.default_rng(77)
    kept = {p: 0 for p in ("fling", "stretchdrag", "drag", "place")}
    for trial in range(12):
        D, S = int(rng.choice([32, 48, 64])), int(rng.choice([96, 200, 400]))
        gd, dd, pd = int(rng.integers(2, 9)), int(rng.integers(2, 12)), int(rng.integers(2, 12))
        scales = np.sort(rng.uniform(0.75, 3.0, size=4))
        rotations = list(np.linspace(-180, 180, 7)[:-1] + rng.uniform(-3, 3))
        depth = (2.0 - rng.uniform(0.0, 0.3, size=(S, S)) * (rng.random((S, S)) < 0.6)).astype(np.float32)
        reach = float(rng.uniform(0.7, 1.3))
        prims = list(kept)
        cfg = dict(obs_dim=D, pix_grasp_dist=gd, pix_drag_dist=dd, pix_place_dist=pd, scales=scales, rotations=rotations,
                   depth=depth, reach_distance_limit=reach, stretchdrag_dist=0.3, grasp_height=0.02,
                   left_arm_base=np.array([0.765, 0, 0]), right_arm_base=np.array([-0.765, 0, 0]))
        sel = ActionSelector.__new__(ActionSelector)          # the host half only: no library, no device
        sel.rotations, sel.obs_dim = rotations, D
        sel.pix_grasp_dist, sel.pix_drag_dist, sel.pix_place_dist = gd, dd, pd
        sel.reach_distance_limit, sel.stretchdrag_dist, sel.grasp_height = reach, 0.3, 0.02
        sel.left_arm_base, sel.right_arm_base = cfg["left_arm_base"].astype(np.float64), cfg["right_arm_base"].astype(np.float64)
        sel.pose = oa.compute_pose(pos=[0, 2, 0], lookat=[0, 0, 0], up=[0, 0, 1])
        for _ in range(240):
            action = prims[int(rng.choice(4, p=[0.3, 0.4, 0.15, 0.15]))]
            x, y, z = int(rng.integers(len(rotations) * len(scales))), int(rng.integers(D)), int(rng.integers(D))
            want = oa.evaluate_candidate(action, x, y, z, cfg)
            got = sel._candidate(action, x, y, z, scales, depth)
            assert (want is None) == (got is None), (trial, action, x, y, z)
            if want is None:
                continue
            kept[action] += 1
            for key in ("p1", "p2"):
                assert got[key].dtype == want[key].dtype and np.array_equal(got[key], want[key]), (trial, action, key)
            assert np.array_equal(got["pretransform_pixels"], want["pretransform_pixels"])
            assert got["left_or_right"] == want["left_or_right"]
            a, b = get_action_params(action, (x, y, z), gd, dd, pd), oa.get_action_params(action, (x, y, z), gd, dd, pd)
            assert all(np.array_equal(p, q) and p.dtype == q.dtype for p, q in zip(a, b))
        for r in rotations:
            for s in scales:
                assert np.array_equal(get_transform_matrix(S, D, -r, s), oa.get_transform_matrix(S, D, -r, s))
    assert all(v >= 20 for v in kept.values()), kept
    with pytest.raises(Exception):
        get_action_params("fold", (0, 1, 2), 4, 4, 4)


def test_envutils_host_mirror_matches_reference_vectors():
    """environment/utils.py:161-276, 579-582 (compute_pose, compute_intrinsics, get_transform_matrix, pixel_to_3d,
    pixels_to_3d_positions, preprocess_obs) run by tests/golden/make_golden.py::envutils_vectors on seeded inputs: the
    product's host mirror (flingbot_amd/action.py) and the oracle's restatements reproduce every output EXACTLY."""
    from flingbot_amd import action as fa
    from oracle import action as oa
    from oracle import observe as oo

    g = np.load(os.path.join(GOLD, "envutils_golden.npz"))
    for mod in (fa, oa):
        for k, (pos, lookat, up) in enumerate(g["pose_in"]):
            assert np.array_equal(mod.compute_pose(pos=list(pos), lookat=list(lookat), up=list(up)), g["pose_out"][k])
        for k, (fov, size) in enumerate(g["intrinsics_in"]):
            assert np.array_equal(mod.compute_intrinsics(fov, size), g["intrinsics_out"][k])
        for k, (a, b, rot, sc_) in enumerate(g["tm_in"]):
            assert np.array_equal(mod.get_transform_matrix(original_dim=int(a), resized_dim=int(b), rotation=rot, scale=sc_),
                                  g["tm_out"][k])
        for pk in range(2):
            for q, (x, y) in enumerate(g["p3d_xy"]):
                got = mod.pixel_to_3d(g["p3d_depth"].copy(), int(x), int(y), pose_matrix=g["pose_out"][pk])
                assert np.array_equal(got, g["p3d_out"][pk][q]) and got.dtype == g["p3d_out"].dtype
        assert np.array_equal(mod.pixel_to_3d(g["p3d_depth"].copy(), 20, 30, pose_matrix=g["pose_out"][0], fov=50.0,
                                              depth_scale=0.5), g["p3d_scaled"])
    assert np.allclose(fa.pixel_to_3d(np.full((400, 400), 2.0, np.float32), 300, 200, g["pose_out"][0]), [0.36, 0, 0],
                       atol=1e-6)  # the known answer SURVEY.md 8c quotes
    for k in range(int(g["pp_n"])):
        v = g[f"pp{k}_in"]
        pix, scale, rot, only = v[:4].astype(int).reshape(2, 2), v[4], v[5], bool(v[6])
        r = fa.pixels_to_3d_positions(pixels=pix, scale=scale, rotation=rot, pretransform_depth=g["pp_depth"].copy(),
                                      transformed_depth=np.zeros((32, 32), np.float32), pose_matrix=g["pose_out"][0],
                                      pretransform_pix_only=only)
        assert bool(r["valid_action"]) == bool(g[f"pp{k}_valid"])
        assert np.array_equal(r["pretransform_pixels"], g[f"pp{k}_pixels"])
        assert (r.get("p1") is not None) == bool(g[f"pp{k}_has_points"])
        if r.get("p1") is not None:
            assert np.array_equal(r["p1"], g[f"pp{k}_p1"]) and np.array_equal(r["p2"], g[f"pp{k}_p2"])
    assert sum(bool(g[f"pp{k}_valid"]) for k in range(int(g["pp_n"]))) >= 3 and not bool(g["pp4_valid"])
    obs = fa.preprocess_obs(g["obs_rgb"].copy(), g["obs_d"].copy())
    assert obs.dtype.is_floating_point and np.array_equal(obs.numpy(), g["obs_out"])
    # the oracle of the device observation stage composes its tensor with the same expression
    ref = np.concatenate([g["obs_rgb"].astype(np.float32) / np.float32(255), g["obs_d"][:, :, None]], 2).transpose(2, 0, 1)
    assert np.array_equal(ref, g["obs_out"]) and oo.get_obs is not None


def test_drag_place_stretchdrag_host_logic_reproduces_reference_golden():
    """primitives.pick_and_drag / pick_and_place / pick_stretch_drag on the CPU oracle retrace what the REFERENCE's
    SimEnv.pick_and_drag_primitive / pick_and_place_primitive / pick_stretch_drag_primitive did
    (tests/golden/primitives_golden.npz), bit for bit."""
    from fling_helpers import OracleBatch, load_primitives_golden, run_primitives_golden

    g = load_primitives_golden()
    run_primitives_golden(lambda n: OracleBatch(n, g["scene_params"], g["init_pos"]),
                          lambda sim, k: sim.get_positions(k), lambda sim, k: sim.get_shape_states(k))


def test_scheduled_drag_place_stretchdrag_programs_reproduce_reference_golden():
    """All five cases of primitives_golden.npz -- three different primitives -- as per-episode programs in ONE scheduling
    run (flingbot_amd/schedule.py), in chunks of 3..7 simulation steps."""
    from fling_helpers import OracleBatch, load_primitives_golden, run_primitives_golden

    g = load_primitives_golden()
    run_primitives_golden(lambda n: OracleBatch(n, g["scene_params"], g["init_pos"]),
                          lambda sim, k: sim.get_positions(k), lambda sim, k: sim.get_shape_states(k), scheduled=True)


def test_task_generator_host_logic_reproduces_reference_golden():
    """flingbot_amd.tasks.generate_hard_tasks on the CPU oracle retraces the REFERENCE's generate_randomization
    (tests/golden/task_golden.npz: two seeded hard tasks), including the random draws, bit for bit."""
    from fling_helpers import oracle_generated_tasks

    assert len(oracle_generated_tasks()) >= 2      # (the checks are inside; the result is shared with tests/test_taskio.py)


def test_observe_oracle_known_answers():
    """oracle/observe.py (cv2 / skimage conventions restated; unpinned against the reference's own cv2 build): colour
    conversion against the analytic HSV of exactly representable colours, resize invariants, component selection."""
    import colorsys
    from oracle import observe as oo

    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [100, 100, 100], [255, 255, 255], [0, 0, 0], [200, 100, 50],
                    [10, 200, 190], [101, 100, 100], [100, 100, 101]]], np.uint8)
    hsv = oo.rgb2hsv_u8(px)[0].astype(int)
    for p, (h, s, v) in zip(px[0], hsv):
        eh, es, ev = colorsys.rgb_to_hsv(*(p / 255.0))
        assert abs(h - eh * 180) <= 0.51 and abs(s - es * 255) <= 0.51 and v == round(ev * 255), (p, h, s, v)
    # cloth test: dark / grey pixels (all of H, S, V <= 100) are background, anything brighter or more saturated is cloth
    assert oo.cloth_mask_raw(px)[0].tolist() == [1, 1, 1, 0, 1, 0, 1, 1, 1, 1]
    # bilinear resize: constants stay constant, identity size is a copy, a horizontal ramp stays monotone, range preserved
    assert np.unique(oo.resize_linear_u8(np.full((72, 72, 3), 137, np.uint8), 40)).tolist() == [137]
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (72, 72, 3)).astype(np.uint8)
    assert np.array_equal(oo.resize_linear_u8(img, 72), img)
    ramp = np.repeat(np.arange(72, dtype=np.uint8)[None, :, None] * 3, 72, 0).repeat(3, 2)
    out = oo.resize_linear_u8(ramp, 40)
    assert (np.diff(out[:, :, 0].astype(int), axis=1) >= 0).all() and out.min() >= 0 and out.max() <= 213
    d = rng.rand(72, 72).astype(np.float32)
    r = oo.resize_linear_f32(d, 40)
    assert r.dtype == np.float32 and r.min() >= d.min() and r.max() <= d.max()
    assert np.allclose(oo.resize_linear_f32(np.full((72, 72), 2.0, np.float32), 40), 2.0)
    # taps of cv::resize: destination 0 of a 720 -> 400 resize samples source 0.4 -> index 0, fraction 0.4
    s_, f_ = oo._linear_taps(400, 720)
    assert s_[0] == 0 and abs(f_[0] - 0.4) < 1e-6 and s_[-1] == 718 and abs(f_[-1] - 0.6) < 1e-4  # float32 (718.6 - 718)
    # largest component: 8-connectivity, ties go to the component met first in raster order, empty mask -> None
    m = np.zeros((10, 10), np.uint8)
    m[1:3, 1:3] = 1; m[3, 3] = 1            # 5 pixels, diagonal contact counts
    m[6:8, 6:8] = 1; m[8, 5] = 1            # 5 pixels as well
    big = oo.largest_component(m)
    assert big.sum() == 5 and big[1, 1] == 1 and big[6, 6] == 0
    m[8, 8] = 1                              # second blob now has 6
    assert oo.largest_component(m)[6, 6] == 1
    assert oo.largest_component(np.zeros((4, 4), np.uint8)) is None and oo.adaptive_crop(None) is None
    # SimEnv.get_obs crop: a centred 10 x 10 blob in a 100 x 100 image -> max(100 - 2 * 45, 100 - 2 * (100 - 54)) * 1.5
    c = np.zeros((100, 100), np.uint8); c[45:55, 45:55] = 1
    assert oo.adaptive_crop(c) == int(max(100 - 2 * 45, 100 - 2 * (100 - 54)) * 1.5) == 15


def test_load_cloth_matches_reference_on_quad_mesh(tmp_path):
    """tasks.load_cloth against the reference's own function run on a synthetic notched quad sheet
    (tests/golden/task_golden.npz obj_*): vertices, triangles and the three edge lists IN THE REFERENCE'S ORDER (the order of
    its Python sets, which becomes the solver's spring order)."""
    from flingbot_amd import tasks as ftasks

    g = np.load(os.path.join(GOLD, "task_golden.npz"), allow_pickle=True)
    path = tmp_path / "sheet_processed.obj"
    path.write_text(str(g["obj_text"]))
    verts, faces, stretch, bend, shear = ftasks.load_cloth(str(path))
    assert np.array_equal(verts, g["obj_vertices"]) and np.array_equal(faces, g["obj_faces"])
    assert np.array_equal(stretch, g["obj_stretch"]) and np.array_equal(bend, g["obj_bend"])
    assert np.array_equal(shear, g["obj_shear"])
    # structure: 16 quads -> 32 triangles and 32 diagonals; every edge ascending and unique
    assert faces.shape == (32, 3) and shear.shape == (32, 2)
    for e in (stretch, bend, shear):
        assert (e[:, 0] < e[:, 1]).all() and len({tuple(r) for r in e.tolist()}) == len(e)
    # the arrays feed the oracle's mesh scene (pyflex.set_scene with cloth_size -1, softgym_cloth.h:69-132)
    from oracle import OracleSim
    sp = np.array([0, 0.2, 0, -1, -1, 0.9, 0.9, 0.9, 2, 0, 2, 0, np.pi / 2, -np.pi / 2, 0, 720, 720, 0.5, 0], np.float32)
    orc = OracleSim()
    orc.set_scene(sp, verts.reshape(-1), stretch.reshape(-1), bend.reshape(-1), shear.reshape(-1), faces.reshape(-1))
    assert orc.n == 30
    orc.step(5)
    assert np.isfinite(orc.get_positions()).all()


def test_observe_resize_agrees_with_an_independent_bilinear():
    """The sampling convention of the restated cv2.resize (pixel centres: src = (dst + 0.5) * scale - 0.5, clamped) is the one
    torch.nn.functional.interpolate(mode='bilinear', align_corners=False) uses: the float path agrees to the rounding of the tap fractions, the
    8-bit fixed-point path to one grey level."""
    import torch
    import torch.nn.functional as F
    from oracle import observe as oo

    rng = np.random.RandomState(3)
    for src, dst in ((72, 40), (90, 90), (50, 77), (720, 400)):
        d = rng.rand(src, src).astype(np.float32) * 2
        ref = F.interpolate(torch.from_numpy(d)[None, None], size=(dst, dst), mode="bilinear", align_corners=False)[0, 0].numpy()
        # (cv2 rounds the tap fraction to float32 from a double expression, torch evaluates it in float32: ~1e-5 apart)
        assert np.abs(oo.resize_linear_f32(d, dst) - ref).max() < 5e-4, (src, dst)
        img = rng.randint(0, 256, (src, src, 3)).astype(np.uint8)
        reff = F.interpolate(torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None], size=(dst, dst), mode="bilinear",
                             align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(oo.resize_linear_u8(img, dst).astype(np.float32) - reff).max() <= 1.0, (src, dst)


def test_env_step_bookkeeping_reproduces_reference_golden():
    """SimEnv.step (environment/simEnv.py:464-515) around the primitives -- preaction / postaction, the "cloth did not move"
    early end, episode_length, the coverage reward, grasp flags that survive an aborted fling -- through
    BatchedFlingEnv.step_actions with the CPU oracle standing in for the device (host logic only), against the golden
    recorded from the reference's own SimEnv.step (tests/golden/make_golden.py step)."""
    from fling_helpers import OracleBatch, load_step_golden, run_step_golden

    g = load_step_golden()
    run_step_golden(lambda n: OracleBatch(n, g["scene_params"], g["init_pos"], pickers=False),
                    lambda sim, k: sim.get_positions(k), lambda sim, k: sim.get_shape_states(k))


@pytest.mark.parametrize("mode", ["blocking", "pipelined", "pipelined-run-ahead", "pipelined-short-chunks"])
def test_scheduler_requests_and_services_on_the_oracle(mode):
    """schedule.run_programs itself, no GPU: programs that mix plain steps, wait_until_stable with DIFFERENT tolerances and
    budgets, a movep and a host-side service, against the same operations done one episode after the other -- through the
    blocking scheduler and through the pipelined one (chunks queued ahead with fs_advance_begin / fs_advance_end; the CPU
    stand-in checks the protocol: no episode is touched while it is a live part of an open chunk, host-side work runs on
    the service lane while chunks are open, chunks are queued on the main lane)."""
    from fling_helpers import OracleBatch, load_fling_golden
    from flingbot_amd import schedule as sch
    from flingbot_amd.primitives import FlingPrimitives

    g = load_fling_golden()

    def lifted(sim, n):
        for e in range(n):
            p = sim.sims[e].get_positions().reshape(-1, 4).copy()
            p[:, 1] += np.float32(0.05 + 0.02 * e)
            sim.sims[e].set_positions(p.ravel())

    n = 3
    a, b = OracleBatch(n, g["scene_params"], g["init_pos"]), OracleBatch(n, g["scene_params"], g["init_pos"])
    lifted(a, n); lifted(b, n)
    prim = FlingPrimitives(a, range(n))
    seen = []

    def program(ep, k):
        yield ("step", 3 + k)
        got = yield ("tag", k)                       # a service of the caller's
        assert got == 10 * k
        yield from sch._movep(ep, [[0.1 * k, 0.2, 0.0], [-0.1, 0.2, 0.05 * k]], speed=2e-2)
        res = yield ("wait", 20 + 5 * k, 0.3 / (k + 1))
        stats = yield ("stats",)
        return res, float(stats[1])

    def tag(reqs):
        seen.append(sorted(e for e, _ in reqs))
        return [10 * args[0] for _, args in reqs]

    kw = {"blocking": dict(cap_min=2, cap=5), "pipelined": dict(cap_min=2, cap=5, pipeline=True),
          "pipelined-run-ahead": dict(cap_min=2, cap=5, pipeline=True, run_ahead=True),
          "pipelined-short-chunks": dict(cap_min=1, cap=2, pipeline=True, run_ahead=True, depth=3)}[mode]
    out = sch.run_programs(prim, {e: program(sch.Episode(prim, e), e) for e in range(n)}, services={"tag": tag}, **kw)
    assert sum(len(s) for s in seen) == n
    assert a.advance_in_flight() == 0 and not getattr(a, "_lane", False)
    steps = 0
    for e in range(n):  # the same, sequentially, on the second batch
        b.step_list([e], 3 + e)
        steps += 3 + e
        b.movep([e], np.array([[[0.1 * e, 0.2, 0.0], [-0.1, 0.2, 0.05 * e]]]), [[False, False]], speed=2e-2)
        steps += b.last_movep_steps
        stable, st = b.wait_until_stable([e], max_steps=20 + 5 * e, tolerance=0.3 / (e + 1))
        steps += int(st[0])
        assert out[e][0] == (bool(stable[0]), int(st[0])), e
        assert out[e][1] == float(b.cloth_stats([e])[0, 1])
        assert np.array_equal(a.get_positions(e).view(np.uint32), b.get_positions(e).view(np.uint32)), e
        assert np.array_equal(np.asarray(a.get_shape_states(e), np.float32).view(np.uint32),
                              np.asarray(b.get_shape_states(e), np.float32).view(np.uint32)), e
    assert prim.sim_steps == steps and len({out[e][0][1] for e in range(n)}) > 1

    # a program that ENDS after a service while another episode stands at a service that is served later in the same pass
    def short(ep, k):
        if k == 0:
            got = yield ("stats",)
            return float(got[0])
        if k == 2:
            yield ("step", 1)
        got = yield ("tag", k)
        return got
    out2 = sch.run_programs(prim, {e: short(sch.Episode(prim, e), e) for e in range(n)}, services={"tag": tag}, **kw)
    assert out2[1] == 10 and out2[2] == 20 and out2[0] == float(a.cloth_stats([0])[0, 0])

    def bad():
        yield ("nonsense",)

    with pytest.raises(ValueError):
        sch.run_programs(prim, {0: bad()})

    # an exception in the middle of a run (a program's own, or a service's) leaves nothing in flight and the lane switched back
    def failing(ep, k):
        yield ("step", 2)
        if k == 1:
            raise RuntimeError("program failed")
        yield ("wait", 30, 1e-9)

    with pytest.raises(RuntimeError, match="program failed"):
        sch.run_programs(prim, {e: failing(sch.Episode(prim, e), e) for e in range(n)}, **kw)
    assert a.advance_in_flight() == 0 and not getattr(a, "_lane", False)


def test_host_scene_derived_tables_match_brute_force():
    """The kernels' derived topology tables, built by fs_scene.cpp (no GPU): the rest-near ids of the SelfCollideFilter test
    (NvFlex.h:166: particles closer than the collision radius in the rest pose) against an all-pairs computation, and the
    streaming kernels' one-byte spring codes + dictionary against the adjacency they encode -- for a grid cloth, a cloth with
    per-type stiffnesses and the .obj mesh of the task golden."""
    from flingbot_amd import sim as fsim

    def check(h, what, expect_mesh_neighbours=True):
        n = h["n"]
        pos = h["positions"].reshape(-1, 4)[:, :3].astype(np.float32)
        par = h["params"]
        r = np.float32(par[6]) + np.float32(par[10])  # radius + particleCollisionMargin
        rn = h["restnear"].reshape(8, n)
        ids = np.stack([rn & 0xffff, rn >> 16], axis=1).reshape(16, n)  # slot q of particle i: word q // 2, half q % 2
        d = pos[:, None, :] - pos[None, :, :]
        e2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        near = e2 < r * r
        np.fill_diagonal(near, False)
        for i in range(n):
            got = {int(v) for v in ids[:, i] if v != 0xffff}
            assert got == set(np.nonzero(near[i])[0].tolist()), (what, i)
        if expect_mesh_neighbours:
            assert near.sum(axis=1).max() >= 8  # interior particles of a grid cloth see their 8 mesh neighbours
        # spring codes: slot s of particle i -> dictionary entry (j - i, length, stiffness) = its s-th incident spring
        codes = h["stream_codes"].reshape(n, 4)
        dic = h["stream_dict"].reshape(-1, 4)
        springs, lens, ks = h["springs"].reshape(-1, 2), h["spring_lengths"], h["spring_stiffness"]
        off, adj = h["adj_offsets"], h["adj_neighbors"]
        by_pair = {}
        for sid, (a, b) in enumerate(springs):
            by_pair.setdefault((int(a), int(b)), []).append(sid)
            by_pair.setdefault((int(b), int(a)), []).append(sid)
        used = 0
        for i in range(0, n, max(1, n // 300)):
            deg = off[i + 1] - off[i]
            for s_ in range(16):
                c = (int(codes[i, s_ // 4]) >> (8 * (s_ % 4))) & 255
                if s_ >= deg:
                    assert c == 255, (what, i, s_)
                    continue
                j = int(adj[off[i] + s_])
                ent = dic[c]
                assert int(ent[0].view(np.int32)) == j - i, (what, i, s_)
                assert any(lens[sid].view(np.uint32) == ent[1] and ks[sid].view(np.uint32) == ent[2] for sid in by_pair[(i, j)]), (what, i, s_)
                used += 1
        assert used > 100

    def stencil_flag(h, dims=None):
        """flags['restnear_ok'] = 2 <=> the host says "every rest-near set is exactly the particle's in-grid 8-neighbourhood"
        (the kernels then test two index differences instead of the packed ids): checked here against all pairs."""
        flag = h["flags"]["restnear_ok"]
        if dims is None:
            return flag
        n, (dx, dz) = h["n"], dims
        pos = h["positions"].reshape(-1, 4)[:, :3].astype(np.float32)
        par = h["params"]
        r = np.float32(par[6]) + np.float32(par[10])
        d = pos[:, None, :] - pos[None, :, :]
        near = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]) < r * r
        np.fill_diagonal(near, False)
        ix, iz = np.arange(n) % dx, np.arange(n) // dx
        eight = (np.abs(ix[:, None] - ix[None, :]) <= 1) & (np.abs(iz[:, None] - iz[None, :]) <= 1)
        np.fill_diagonal(eight, False)
        assert (flag == 2) == bool(np.array_equal(near, eight)), dims
        return flag

    from conftest import cloth_params
    h = fsim.host_scene(cloth_params(37, 23))
    check(h, "grid 37 x 23")
    assert stencil_flag(h, (37, 23)) == 2
    h = fsim.host_scene(cloth_params(24, 30, stiff=(0.8, 1.0, 0.6)))
    check(h, "grid 24 x 30, per-type stiffness")
    assert stencil_flag(h, (24, 30)) == 2
    g = np.load(os.path.join(GOLD, "task_golden.npz"))
    sp = cloth_params(0, 0)
    h = fsim.host_scene(sp, g["obj_vertices"].reshape(-1), g["obj_stretch"].reshape(-1), g["obj_bend"].reshape(-1),
                        g["obj_shear"].reshape(-1), g["obj_faces"].reshape(-1))
    check(h, "obj mesh", expect_mesh_neighbours=False)
    assert stencil_flag(h) != 2  # an irregular mesh keeps the packed-id test


def test_sphere_mesh_pinned_to_reference_mesh():
    """oracle/_ref/sphere_ref -- the reference's own core/mesh.cpp CreateSphere(20, 20, r) + Mesh::Transform, compiled where
    it lies -- recorded five spheres in tests/golden/sphere_golden.json (FlingBot's pickers with their [1, 0, 0, 0]
    quaternion, an identity rotation, two general ones).  The oracle's restatement (raster_oracle.c orc_sphere_mesh) and
    the product's host code (the functions fs_render's kernel runs, through fs_host_sphere_mesh) both give the same 2 400
    indices, 441 positions and 441 normals EXACTLY."""
    import json

    from flingbot_amd import sim as fsim
    from oracle.render import sphere_mesh as orc_sphere_mesh

    with open(os.path.join(GOLD, "sphere_golden.json")) as fh:
        gold = json.load(fh)
    assert len(gold) == 5
    for c in gold:
        assert c["counts"] == [441, 441, 2400]
        gp = np.array(c["positions"], np.float32).reshape(441, 3)
        gn = np.array(c["normals"], np.float32).reshape(441, 3)
        gi = np.array(c["indices"], np.int32).reshape(800, 3)
        st = np.zeros(14, np.float32)
        st[0:3], st[3:6], st[6:10], st[10:14] = [7.0, 8.0, 9.0], c["pos"], [0.0, 0.0, 0.0, 1.0], c["quat"]
        for name, (v, n, t) in (("oracle", orc_sphere_mesh(st, [c["radius"]])),
                                ("product", fsim.host_sphere_mesh(c["radius"], c["pos"], c["quat"]))):
            assert np.array_equal(t, gi), name
            assert np.array_equal(v[:, :3], gp), (name, np.abs(v[:, :3] - gp).max())
            assert np.array_equal(n[:, :3], gn), (name, np.abs(n[:, :3] - gn).max())
    # FlingBot's pickers are drawn upside down: vertex 0 is the south pole (quaternion x = 1 is half a turn about x)
    assert gold[0]["quat"] == [1.0, 0.0, 0.0, 0.0] and gold[0]["normals"][:3] == [0.0, -1.0, 0.0]


def test_prebuilt_scenes_host_side():
    """The host half of fs_set_scene built ahead (sim.PrebuiltScene / tasks.ScenePrebuilder: no GPU, any thread): the worker
    thread hands out one scene per task in order, a scene is consumed once, and what it holds is what fs_host_scene_build
    returns for the task's arguments (same counts and arrays as flingbot_amd.sim.host_scene)."""
    from flingbot_amd import sim as fsim, tasks as ftasks

    rng = np.random.RandomState(4)
    tasks = []
    for k in range(5):
        dx, dz = int(rng.randint(8, 20)), int(rng.randint(8, 20))
        tasks.append(dict(mesh_verts=np.zeros(0, np.float32), mesh_stretch_edges=np.zeros(0, np.int32),
                          mesh_bend_edges=np.zeros(0, np.int32), mesh_shear_edges=np.zeros(0, np.int32),
                          mesh_faces=np.zeros(0, np.int32), cloth_size=(dx, dz), cloth_stiff=(0.9, 1.0, 0.8), cloth_mass=0.5,
                          flip_mesh=0))
    pre = ftasks.ScenePrebuilder(tasks, ahead=2)
    lib = fsim.load_library()
    try:
        for i in (0, 1, 2, 4):  # (3 is never asked for: close() frees it)
            scene = pre.get(i)
            cnt = [C.c_int(0) for _ in range(4)]
            assert lib.fs_host_scene_counts(scene.take(), *[C.byref(c) for c in cnt]) == 0
            ref = fsim.host_scene(*ftasks.task_scene_arguments(tasks[i]))
            assert [c.value for c in cnt] == [ref["n"], ref["m"], ref["t"], ref["max_deg"]]
            edges = np.zeros(2 * ref["m"], np.int32)
            assert lib.fs_host_scene_copy(scene.take(), 3, edges.ctypes.data_as(C.c_void_p), edges.size) == 0
            assert np.array_equal(edges, ref["springs"])
            scene.free()
            with pytest.raises(fsim.FlingSimError):
                scene.take()
    finally:
        pre.close()
    assert not pre.futures
    # a queue that learns its tasks as it goes (evaluate.run_tasks with a shared counter across ranks): nothing is built until the
    # queue says what it expects, then in THAT order, and a task it never announced is built on the spot
    pre = ftasks.ScenePrebuilder(tasks, ahead=2, order=[])
    try:
        assert not pre.futures
        pre.expect([4, 1])
        assert set(pre.futures) == {4, 1}
        pre.expect([0])
        assert set(pre.futures) == {4, 1}                     # `ahead` = 2 scenes beyond the ones handed out
        for i in (4, 1, 3, 0):
            scene = pre.get(i)
            cnt = [C.c_int(0) for _ in range(4)]
            assert lib.fs_host_scene_counts(scene.take(), *[C.byref(c) for c in cnt]) == 0
            assert cnt[0].value == tasks[i]["cloth_size"][0] * tasks[i]["cloth_size"][1]
            scene.free()
    finally:
        pre.close()
    assert not pre.futures


def test_run_tasks_shared_claim_logic_on_the_oracle():
    """evaluate.run_tasks with claim= (the shared task queue of a multi-rank evaluation, distributed.SharedTaskCounter) without a
    GPU: the scheduler runs on the oracle-backed stand-in, the environment's episode program is a two-request stub.  Two calls
    share one counter -- the first stops claiming after four tasks -- and together they run every task exactly once; the up-front
    claim respects claim_first, later claims come a sixteenth of the slots (here: one) at a time; a call that gets nothing returns
    empty statistics instead of failing."""
    from fling_helpers import OracleBatch, load_fling_golden
    from flingbot_amd import distributed as fdist, evaluate
    from flingbot_amd.primitives import FlingPrimitives

    g = load_fling_golden()
    n_tasks, slots = 9, 2
    tasks = [{"flatten_area": 0.5 + 0.01 * i, "mesh_verts": np.zeros(0)} for i in range(n_tasks)]
    counter, asked = fdist.SharedTaskCounter(n_tasks), []

    class Env:
        actions, device = ["fling"], None
        unpaid_steps = 0

        def __init__(self):
            self.sim = OracleBatch(slots, g["scene_params"], g["init_pos"])
            self.sim.n_envs = slots
            self.prim = FlingPrimitives(self.sim, range(slots))

        def open_slots(self, s):
            pass

        def episode_program(self, slot, task, max_actions=None, prebuilt=None):
            yield ("step", 1 + int(round(100 * (task["flatten_area"] - 0.5))) % 3)
            cov = yield ("coverage",)
            return {"coverage": [0.1, float(task["flatten_area"]) / 2], "actions": ["fling"], "rewards": [0.0], "preaction_coverage": [0.1]}

    class Policy:
        value_nets = {}

    def limited(k):
        k = min(k, max(0, 4 - len(counter.claimed)))
        asked.append(k)
        return counter.claim(k)

    parts = []
    for claim in (limited, counter.claim, counter.claim):
        env = Env()
        env.sim.coverage = lambda: np.zeros(slots)
        parts.append(evaluate.run_tasks(Policy(), env, tasks, fold=False, pipeline=False, prebuild=False, claim=claim, claim_first=2))
    assert asked[0] == 2 and set(asked[1:]) <= {0, 1}                       # fair share up front, then one at a time
    got = [p["task_indices"].tolist() for p in parts]
    assert got[0] == [0, 1, 2, 3] and got[1] == [4, 5, 6, 7, 8] and got[2] == []
    assert np.allclose(parts[1]["final_coverage"], 0.5) and parts[1]["action_primitive_counts"] == {"fling": 5}
    assert parts[2]["records"] == [] and np.isnan(parts[2]["mean"]["final_coverage"]) and parts[2]["coverage_steps"].shape[1] == 0
