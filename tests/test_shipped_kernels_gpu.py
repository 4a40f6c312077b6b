"""GPU parity of the kernels that are SHIPPED AND TIMED, in the launch shapes they are timed in.

tests/test_parity_gpu.py walks the physics (drop / crumple / fling / dense ball / tethers / phases) mostly one episode at
a time; a one-episode launch, however, runs neither the bench configuration (256 episodes, one fused workgroup each,
blockIdx > 0) nor the throughput forms of the streaming iterate kernel, which the launch size selects (fs_solver.hip).
Here every form is launched at a size that selects it -- asserted through the white-box getter fs_last_kernel_form --
on distinct-seed episodes, and sampled episodes are compared with the CPU oracle bit for bit:

    fs_k_fused_grid64     8 and 256 episodes of bench.py's own workload (BASELINE.json configs[1] / the bench line):
                          what AUTO / FUSED run for 64-wide grid cloths; plus its general path (picked particles,
                          coincident particles), fewer than 64 rows, and per-type stiffnesses
    fs_k_fused_step<12>   the same batches with FS_SOLVER_FUSED_CODED (every other cloth that fits the fused kernel)
    fs_k_fused_step<0>    FS_SOLVER_FUSED_GENERIC, 9 episodes
    fs_k_fused_step<16>   a mesh cloth with 16 springs per particle, 3 episodes
    fs_k_iterate_gridl    64 x 64x64 episodes, AUTO (BASELINE.json configs[2] and configs[3]'s per-GPU share); 8 episodes;
                          16 x 104x104 episodes (the upper end of the reference's cloth sizes, environment/tasks.py:108-121)
    fs_k_iterate<true>    64 x 64x64 episodes, FS_SOLVER_STREAM_CODED (non-canonical cloths at that size)
    fs_k_iterate<false>   64 x 64x64 episodes, FS_SOLVER_STREAM_ELL
    fs_k_boundary         every streaming case above (finalize + predict + bucket sort in one launch per substep boundary);
                          FS_SOLVER_STREAM_SPLIT (7) runs the 64-episode case with the four separate kernels that cloths
                          above 16384 particles take
    fs_k_iterate_eager    8 x 64x64 episodes, FS_SOLVER_STREAM_CODED
    fs_k_iterate_grid     112 x 64x64 distinct episodes, FS_SOLVER_STREAM_CODED
"""
import os
import threading

import numpy as np
import pytest

import bench
from conftest import cloth_params

pytestmark = pytest.mark.gpu


def _oracle_runs(setups, steps):
    """Run one oracle episode per entry of `setups` (callables taking the OracleSim) for `steps` frames, in parallel
    threads (ctypes releases the GIL around orc_step)."""
    from oracle import OracleSim

    sims = [OracleSim() for _ in setups]

    def work(k):
        setups[k](sims[k])
        sims[k].step(steps)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(setups))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return sims


def _assert_bits(ctx, e, orc, what):
    ph, po = ctx.get_positions(e), orc.get_positions()
    vh, vo = ctx.get_velocities(e), orc.get_velocities()
    assert np.isfinite(po).all()
    scale = max(1.0, float(np.abs(po).max()))
    assert np.abs(ph - po).max() <= 1e-4 * scale, f"{what}: positions outside north_star's 1e-4 bar"
    assert np.array_equal(ph.view(np.uint32), po.view(np.uint32)), \
        f"{what}: positions not bit-exact (max abs diff {np.abs(ph - po).max():.3e})"
    assert np.array_equal(vh.view(np.uint32), vo.view(np.uint32)), \
        f"{what}: velocities not bit-exact (max abs diff {np.abs(vh - vo).max():.3e})"


def _bench_batch(n_envs, solver, steps, sample, expect_form, groups=0, expect_groups=None):
    from flingbot_amd import sim as fsim

    ctx = fsim.FlingSim(n_envs=n_envs, solver=solver)
    ctx.set_stream_groups(groups)
    for e in range(n_envs):
        bench.setup_episode(ctx.env(e), seed=e)  # exactly the episodes bench.py times
    ctx.step(steps)  # one launch (fused) / one launch sequence (streaming) for all episodes
    assert ctx.last_kernel_form() == expect_form, ctx.last_kernel_form()
    if expect_groups is not None:
        assert ctx.last_stream_groups() == expect_groups, ctx.last_stream_groups()
    orcs = _oracle_runs([lambda o, s=s: bench.setup_episode(o, seed=s) for s in sample], steps)
    for s, o in zip(sample, orcs):
        _assert_bits(ctx, s, o, f"{n_envs} episodes, form {expect_form}, episode {s}")
    co, _ = orcs[0].get_last_neighbors()
    ctx.close()
    return int(co.sum())


@pytest.mark.parametrize("n_envs", [8, 256])
@pytest.mark.parametrize("solver,form", [("FS_SOLVER_FUSED", "FS_FORM_FUSED_GRID64"), ("FS_SOLVER_FUSED_CODED", "FS_FORM_FUSED_12")])
def test_fused_kernel_bench_batch_bit_exact(gpu_required, n_envs, solver, form):
    """The bench configuration itself: `n_envs` distinct-seed 64x64 episodes of bench.py's workload in ONE launch of the
    fused kernel (256 = one workgroup per CU) -- fs_k_fused_grid64, the form a 64-wide grid cloth gets, and
    fs_k_fused_step<12>, the dictionary-coded form of every other cloth; first, second, middle and last episode equal the
    oracle after 40 frames (free fall, ground contact with friction, the lower rows folding onto each other)."""
    from flingbot_amd import sim as fsim

    sample = sorted({0, 1, n_envs // 2, n_envs - 1})
    contacts = _bench_batch(n_envs, getattr(fsim, solver), 40, sample, getattr(fsim, form))
    assert contacts > 50, "the sampled episode must have particle contacts"


def _folded_sheet(sim, shift):
    """A 64x64 cloth folded in half on the ground: rows 32..63 lie 10.5 / 11 mm above rows 31..0, so ~3 500 particles have
    exactly one contact candidate (shift 0: the particle below) or two (shift half a pitch in x: the two below)."""
    sim.set_scene(cloth_params(64, 64, pos=(0.0, -0.005, 0.0)))
    g = sim.get_positions().reshape(64, 64, 4).copy()
    for iz in range(32, 64):
        g[iz, :, 2] = g[63 - iz, :, 2]
        g[iz, :, 1] = g[63 - iz, :, 1] + (0.0105 if shift else 0.011)
        g[iz, :, 0] += shift
    sim.set_positions(g.reshape(-1))
    sim.set_velocities(np.zeros(3 * 4096, np.float32))


@pytest.mark.parametrize("shift", [0.0, 0.003125])
@pytest.mark.parametrize("solver,form", [("FS_SOLVER_FUSED", "FS_FORM_FUSED_GRID64"), ("FS_SOLVER_FUSED_CODED", "FS_FORM_FUSED_12")])
def test_fused_kernels_overflow_queue_beyond_capacity(gpu_required, solver, form, shift):
    """The fused kernels finish particle contacts in three places: the contact set (the 1024 longest lists), the overflow queue
    (1536 entries, finished in pass 2) and, beyond both, inline in the main loop.  A folded sheet gives ~3 500 particles one
    candidate each (or ~2 900 two each): more than set + queue hold, so all three run, the queue also with second candidates."""
    from flingbot_amd import sim as fsim

    ctx = fsim.FlingSim(n_envs=2, solver=getattr(fsim, solver))
    for e in range(2):
        _folded_sheet(ctx.env(e), shift)
    ctx.step(8)
    assert ctx.last_kernel_form() == getattr(fsim, form), ctx.last_kernel_form()
    orc = _oracle_runs([lambda o: _folded_sheet(o, shift)], 8)[0]
    co, _ = orc.get_last_neighbors()
    assert (co > 0).sum() > 1024 + 1536 and co.max() == (2 if shift else 1), ((co > 0).sum(), co.max())
    for e in range(2):
        _assert_bits(ctx, e, orc, f"folded sheet, shift {shift}, episode {e}")
    ch, _ = ctx.get_last_neighbors(0)
    assert np.array_equal(ch, co)
    ctx.close()


def test_grid64_kernel_general_paths_bit_exact(gpu_required):
    """fs_k_fused_grid64 beyond the free sheet: (a) a two-picker fling of a 64x64 cloth -- picked particles have inverse
    mass 0, so the waves around them leave the equal-mass spring form for the general one (ELL adjacency) while the rest
    of the cloth stays on the fast one; (b) two particles placed on EXACTLY the same point (squared spring length 0: the
    spring is skipped and not counted, fs_spring's `length > 0`), which the fast form detects and hands to the general
    path; (c) a 64 x 40 cloth (rows past the cloth, the pairs whose second row does not exist) with different stretch /
    bend / shear stiffnesses."""
    import scenarios as sc
    from flingbot_amd import sim as fsim
    from oracle import OracleSim

    # (a)
    ctx = fsim.FlingSim(n_envs=1, solver=fsim.FS_SOLVER_FUSED)
    hip, orc = ctx.env(0), OracleSim()
    ph = sc.scenario_fling(hip, 64, 64, settle_steps=10)
    assert ctx.last_kernel_form() == fsim.FS_FORM_FUSED_GRID64
    po = sc.scenario_fling(orc, 64, 64, settle_steps=10)
    assert ph.picked == po.picked
    _assert_bits(ctx, 0, orc, "grid64, fling with pickers")
    ctx.close()
    # (b)
    ctx = fsim.FlingSim(n_envs=2, solver=fsim.FS_SOLVER_FUSED)
    orcs = [OracleSim(), OracleSim()]
    p = cloth_params(64, 64, pos=(0.0, -0.1, 0.0))
    for e in range(2):
        for s_ in (ctx.env(e), orcs[e]):
            s_.set_scene(p)
            pos = s_.get_positions().reshape(-1, 4).copy()
            i = 64 * (20 + 17 * e) + 31
            pos[i + 1, :3] = pos[i, :3]          # stretch neighbours (i, i+1) coincide
            pos[i + 64, :3] = pos[i, :3]         # and a z-direction neighbour as well
            s_.set_positions(pos.ravel())
    ctx.step(6)
    assert ctx.last_kernel_form() == fsim.FS_FORM_FUSED_GRID64
    for e in range(2):
        orcs[e].step(6)
        _assert_bits(ctx, e, orcs[e], f"grid64, coincident particles, episode {e}")
    ctx.close()
    # (c)
    ctx = fsim.FlingSim(n_envs=3, solver=fsim.FS_SOLVER_AUTO)
    ctx.set_solver(fsim.FS_SOLVER_FUSED)
    orcs = [OracleSim() for _ in range(3)]
    p = cloth_params(64, 40, pos=(0.1, -0.08, -0.2), stiff=(0.8, 1.0, 0.6), mass=0.3)
    for e in range(3):
        rng = np.random.RandomState(40 + e)
        for s_ in (ctx.env(e), orcs[e]):
            s_.set_scene(p)
        pos = orcs[e].get_positions().reshape(-1, 4).copy()
        pos[:, :3] += (rng.rand(pos.shape[0], 3).astype(np.float32) - 0.5) * 0.004
        pos[: 64 * 6, 1] += 0.008            # the first six rows lie 8 mm above rows 8..13: particle contacts from
        pos[: 64 * 6, 2] += 8 * 0.00625      # the first substep on (not rest neighbours: eight rows apart)
        for s_ in (ctx.env(e), orcs[e]):
            s_.set_positions(pos.ravel())
    ctx.step(1)
    orcs[0].step(1)
    co, _ = orcs[0].get_last_neighbors()
    assert co.sum() > 50, "the two layers must be in contact while they are pushed apart"
    _assert_bits(ctx, 0, orcs[0], "grid64, 64 x 40 cloth, first frame")
    ctx.step(29)
    orcs[0].step(29)
    assert ctx.last_kernel_form() == fsim.FS_FORM_FUSED_GRID64
    for e in range(3):
        if e:
            orcs[e].step(30)
        _assert_bits(ctx, e, orcs[e], f"grid64, 64 x 40 cloth, episode {e}")
    ctx.close()


def test_grid64_eligibility(gpu_required):
    """Cloths the grid-64 form must NOT take: tethers (negative stiffness), other widths, meshes -> coded kernel."""
    from flingbot_amd import sim as fsim

    cases = [(cloth_params(64, 64, stiff=(0.9, -0.6, 0.9)), fsim.FS_FORM_FUSED_12),
             (cloth_params(32, 64), fsim.FS_FORM_FUSED_12), (cloth_params(63, 64), fsim.FS_FORM_FUSED_12),
             (cloth_params(64, 64), fsim.FS_FORM_FUSED_GRID64), (cloth_params(64, 5), fsim.FS_FORM_FUSED_GRID64)]
    for p, form in cases:
        ctx = fsim.FlingSim(n_envs=1, solver=fsim.FS_SOLVER_FUSED)
        ctx.set_scene(0, p)
        ctx.step(1)
        assert ctx.last_kernel_form() == form, (p[3:8], ctx.last_kernel_form())
        ctx.close()


def test_fused_generic_kernel_batch_bit_exact(gpu_required):
    """fs_k_fused_step<0> (adjacency streamed from the ELL arrays instead of the register-resident codes)."""
    from flingbot_amd import sim as fsim

    _bench_batch(9, fsim.FS_SOLVER_FUSED_GENERIC, 30, [0, 4, 8], fsim.FS_FORM_FUSED_GENERIC)


def _deg16_mesh(nx, nz, sp=0.00625):
    """Grid sheet through the MESH path of set_scene (softgym_cloth.h:69-132) with 3-hop springs added to the bend list:
    4 stretch + 4 shear + 4 two-hop + 4 three-hop = 16 springs per interior particle."""
    verts = np.array([[x * sp, 0.0, z * sp] for z in range(nz) for x in range(nx)], np.float32)
    idx = lambda x, z: z * nx + x
    faces, stretch, bend, shear = [], [], [], []
    for z in range(nz):
        for x in range(nx):
            if x + 1 < nx: stretch.append((idx(x, z), idx(x + 1, z)))
            if z + 1 < nz: stretch.append((idx(x, z), idx(x, z + 1)))
            for hop in (2, 3):
                if x + hop < nx: bend.append((idx(x, z), idx(x + hop, z)))
                if z + hop < nz: bend.append((idx(x, z), idx(x, z + hop)))
            if x + 1 < nx and z + 1 < nz:
                shear.append((idx(x, z), idx(x + 1, z + 1)))
                shear.append((idx(x + 1, z), idx(x, z + 1)))
                faces.append((idx(x, z), idx(x + 1, z), idx(x + 1, z + 1)))
                faces.append((idx(x, z), idx(x + 1, z + 1), idx(x, z + 1)))
    return [verts.ravel()] + [np.array(a, np.int32).ravel() for a in (stretch, bend, shear, faces)]


def test_fused_16_slot_kernel_bit_exact(gpu_required):
    """fs_k_fused_step<16>: a cloth with 13..16 springs per particle keeps the coded adjacency in 16 register slots."""
    from flingbot_amd import sim as fsim

    mesh = _deg16_mesh(40, 36)
    p = cloth_params(0, 0, pos=(0.0, -0.06, 0.0), mass=0.3)
    n_envs = 3
    ctx = fsim.FlingSim(n_envs=n_envs, solver=fsim.FS_SOLVER_FUSED)

    def setup(sim, seed):
        sim.set_scene(p, *mesh)
        rng = np.random.RandomState(seed)
        pos = sim.get_positions().reshape(-1, 4).copy()
        pos[:, :3] += (rng.rand(pos.shape[0], 3).astype(np.float32) - 0.5) * 0.004
        pos[: 300, 1] += 0.02  # a flap above the sheet: falls onto it
        pos[: 300, 0] += 0.05
        sim.set_positions(pos.ravel())

    for e in range(n_envs):
        setup(ctx.env(e), e)
    ctx.step(30)
    assert ctx.last_kernel_form() == fsim.FS_FORM_FUSED_16
    orcs = _oracle_runs([lambda o, s=s: setup(o, s) for s in range(n_envs)], 30)
    for e in range(n_envs):
        _assert_bits(ctx, e, orcs[e], f"16-slot fused kernel, episode {e}")
    ctx.close()


@pytest.mark.parametrize("solver,form", [(0, "FS_FORM_STREAM_GRIDL"), (6, "FS_FORM_STREAM_CODED"), (4, "FS_FORM_STREAM_ELL"),
                                         (7, "FS_FORM_STREAM_GRIDL")])
def test_streaming_64_episode_launch_bit_exact(gpu_required, solver, form):
    """BASELINE.json configs[2] (and configs[3]'s per-GPU share): 64 distinct 64x64 episodes in one launch sequence.
    AUTO routes that size to the streaming back-end, whose iterate kernel for canonical grid cloths is fs_k_iterate_gridl
    (neighbours from the grid coordinates, rest lengths from the per-particle table); FS_SOLVER_STREAM_CODED runs the
    dictionary-coded throughput form fs_k_iterate<true> every other cloth gets at this size, FS_SOLVER_STREAM_ELL
    fs_k_iterate<false>; FS_SOLVER_STREAM_SPLIT the grid-L form with separate finalize / predict / scan / scatter launches
    instead of fs_k_boundary (all other cases run the merged boundary kernel; 40 frames in one call chain the frames through
    its finalize + predict form)."""
    from flingbot_amd import sim as fsim

    # 64 x 4096 particles: the launch list is split into two concurrent chains (slots 0..31 and 32..63)
    contacts = _bench_batch(64, solver, 40, [0, 1, 31, 32, 63], getattr(fsim, form), expect_groups=2)
    assert contacts > 50


@pytest.mark.parametrize("n_envs,groups", [(20, 3), (64, 1), (64, 4), (130, 0)])
def test_streaming_concurrent_chains_bit_exact(gpu_required, n_envs, groups):
    """The streaming launch list split into 1..4 slot ranges whose launch chains run concurrently on their own streams
    (fs_set_stream_groups; 0 = the default, two chains from 24 x 4096 particles on): slot ranges of 8 / 8 / 4 episodes,
    the unsplit launch, four ranges, and ranges that end inside the last block of eight -- first / boundary / last episodes
    equal the oracle."""
    from flingbot_amd import sim as fsim

    sample = sorted({0, 7, 8, 15, 16, n_envs // 2, n_envs - 1} & set(range(n_envs)))
    # (130 x 4096 particles is above the 262 144 from which tether-free grid cloths take the throughput form of the iteration)
    form = fsim.FS_FORM_STREAM_GRIDL_TP if n_envs * 4096 > 262144 else fsim.FS_FORM_STREAM_GRIDL
    _bench_batch(n_envs, fsim.FS_SOLVER_STREAM, 30, sample, form, groups=groups, expect_groups=groups if groups else 2)


def test_streaming_small_and_large_launch_forms_bit_exact(gpu_required):
    """The other two forms the launch size selects, on distinct episodes: fs_k_iterate_eager (8 episodes) and
    fs_k_iterate_grid (112 episodes >= 96 x 4096 particles)."""
    from flingbot_amd import sim as fsim

    _bench_batch(8, fsim.FS_SOLVER_STREAM_CODED, 40, [0, 7], fsim.FS_FORM_STREAM_EAGER)
    _bench_batch(112, fsim.FS_SOLVER_STREAM_CODED, 30, [0, 55, 111], fsim.FS_FORM_STREAM_GRID)
    _bench_batch(8, fsim.FS_SOLVER_STREAM, 40, [0, 7], fsim.FS_FORM_STREAM_GRIDL)
    _bench_batch(3, fsim.FS_SOLVER_STREAM_MERGED, 40, [0, 2], fsim.FS_FORM_STREAM_GRIDL)  # fs_k_boundary below its launch-size threshold


def test_large_cloth_104_batch_bit_exact(gpu_required):
    """16 episodes of a 104x104 cloth (10 816 particles each: the upper end of the reference's task sizes,
    environment/tasks.py:108-121), loose heaps so the neighbour lists are long; AUTO -> streaming, throughput form."""
    from flingbot_amd import sim as fsim

    n_envs, dim, steps = 16, 104, 10
    p = cloth_params(dim, dim, pos=(0.0, -0.15, 0.0))
    ctx = fsim.FlingSim(n_envs=n_envs, solver=fsim.FS_SOLVER_AUTO)

    def setup(sim, seed):
        sim.set_scene(p)
        rng = np.random.RandomState(100 + seed)
        pos = sim.get_positions().reshape(-1, 4).copy()
        pos[:, :3] += (rng.randn(pos.shape[0], 3) * 0.003).astype(np.float32)
        k = 2500
        pos[:k, :3] = (rng.rand(k, 3) * [0.12, 0.05, 0.12] + [0.0, 0.02, 0.0]).astype(np.float32)  # a heap on the ground
        sim.set_positions(pos.ravel())

    for e in range(n_envs):
        setup(ctx.env(e), e)
    ctx.step(steps)
    assert ctx.last_kernel_form() == fsim.FS_FORM_STREAM_GRIDL
    assert ctx.last_boundary_form() == 1      # fs_k_boundary: 11 particles per thread
    sample = [0, 9, 15]
    orcs = _oracle_runs([lambda o, s=s: setup(o, s) for s in sample], steps)
    for s, o in zip(sample, orcs):
        _assert_bits(ctx, s, o, f"104x104 episode {s}")
    co, _ = orcs[0].get_last_neighbors()
    assert co.max() > 16, "the heap must produce long neighbour lists"
    ctx.close()


def test_boundary_on_mixed_cloth_sizes_bit_exact(gpu_required):
    """The one-launch substep boundary on ONE launch list that mixes cloth sizes -- 64 x 64, 72 x 72, 96 x 96 -- in two fs_step
    calls (the second continues from the state the first left, histogram included): the oracle's bits."""
    from flingbot_amd import sim as fsim

    dims = [64, 72, 96, 72, 64, 96, 72, 64, 96, 72, 64, 96, 72, 64, 96, 72, 64, 96]
    steps = 8

    def setup(sim, k):
        sim.set_scene(cloth_params(dims[k], dims[k], pos=(0.0, -0.12, 0.0)))
        rng = np.random.RandomState(7 + k)
        pos = sim.get_positions().reshape(-1, 4).copy()
        pos[:, :3] += (rng.randn(pos.shape[0], 3) * 0.004).astype(np.float32)
        pos[:1500, :3] = (rng.rand(1500, 3) * [0.1, 0.04, 0.1] + [0.0, 0.02, 0.0]).astype(np.float32)
        sim.set_positions(pos.ravel())

    sample = [0, 1, 2, 17]
    orcs = _oracle_runs([lambda o, s=s: setup(o, s) for s in sample], steps)
    ctx = fsim.FlingSim(n_envs=len(dims), solver=fsim.FS_SOLVER_AUTO)
    for e in range(len(dims)):
        setup(ctx.env(e), e)
    ctx.step(steps // 2)
    ctx.step(steps - steps // 2)
    assert ctx.last_boundary_form() == 1
    for s, o in zip(sample, orcs):
        _assert_bits(ctx, s, o, f"episode {s} ({dims[s]} x {dims[s]})")
    ctx.close()


def test_cotenant_solver_mode_chooses_per_launch(gpu_required):
    """FS_SOLVER_COTENANT (what the `pyflex` module runs on when the co-tenant table shows a shared device): every launch whose
    episodes all fit the fused kernel takes it, whatever the launch size -- one launch per frame on one compute unit each, so that
    co-tenants' frames run side by side -- and any other launch is AUTO's; no launch can fail because of the mode.  Same bits."""
    from flingbot_amd import sim as fsim

    dims = [32, 104, 48]

    def setup(sim, k):
        sim.set_scene(cloth_params(dims[k], dims[k], pos=(0.0, -0.1, 0.0)))

    orcs = _oracle_runs([lambda o, k=k: setup(o, k) for k in range(3)], 6)
    ctx = fsim.FlingSim(n_envs=3, solver=fsim.FS_SOLVER_COTENANT)
    for k in range(3):
        setup(ctx.env(k), k)
    fused = (fsim.FS_FORM_FUSED_12, fsim.FS_FORM_FUSED_16, fsim.FS_FORM_FUSED_GENERIC, fsim.FS_FORM_FUSED_GRID64)
    ctx.step_list([0, 2], 3)                       # both fit the fused kernel: taken although two episodes are a "small launch"
    assert ctx.last_kernel_form() in fused
    ctx.step_list([1], 3)                          # 10 816 particles: AUTO's choice, the streaming kernels
    assert ctx.last_kernel_form() not in fused
    ctx.step_list([0, 2], 3)
    ctx.step(3, env=1)
    assert ctx.last_kernel_form() not in fused
    for k in range(3):
        _assert_bits(ctx, k, orcs[k], f"cotenant mode, episode {k} ({dims[k]} x {dims[k]})")
    ctx.step(1)                                    # a mixed list: falls back to AUTO as a whole, never an error
    assert ctx.last_kernel_form() not in fused
    ctx.close()


def test_gridl_throughput_form_bit_exact_with_pins_spheres_and_mixed_sizes(gpu_required):
    """fs_k_iterate_gridl_tp (round 6): the iteration kernel of streaming launches above 262 144 particles of tether-free grid
    cloths -- the evaluation loop at the reference's task sizes.  Springs are gathered and evaluated in two halves of six; a half
    whose neighbours all have the particle's own mass takes the equal-mass form (no out-of-grid test: a slot that leaves the grid
    gathers the particle itself and has length 0), a wavefront that sees a pinned particle takes the general form for that half.
    60 cloths of three sizes (342 k particles) crumpled into heaps; some with a PINNED particle (invMass 0, also one on the cloth's
    border and one in its first row), some with two MOVING kinematic spheres; two fs_step calls and a launch of a subset that
    is itself above the threshold.  Sampled episodes: the oracle's bits; and FLINGSIM_GRIDL_TP=0 (the general kernel) gives the
    same bits for every episode."""
    from flingbot_amd import sim as fsim

    dims = [(64, 64), (72, 80), (96, 70)]
    n_envs, steps = 60, 6

    def setup(sim, k):
        dx, dz = dims[k % 3]
        sim.set_scene(cloth_params(dx, dz, pos=(0.0, -0.12, 0.0)))
        rng = np.random.RandomState(50 + k)
        pos = sim.get_positions().reshape(-1, 4).copy()
        pos[:, :3] += (rng.randn(pos.shape[0], 3) * 0.004).astype(np.float32)
        pos[:2000, :3] = (rng.rand(2000, 3) * [0.1, 0.04, 0.1] + [0.0, 0.02, 0.0]).astype(np.float32)
        if k % 4 == 1:                       # pinned particles: interior, border column, first row
            for pin in (dx * (dz // 2) + dx // 2, dx * 3, 5):
                pos[pin, 3] = 0.0
                pos[pin, 1] += 0.05
        sim.set_positions(pos.ravel())
        if k % 4 == 2:
            for c in ((0.03, 0.04, 0.03), (0.08, 0.05, 0.06)):
                sim.add_sphere(0.02, c, [1, 0, 0, 0])

    def move_spheres(get, put):
        st = np.array(get(), np.float32).reshape(-1, 14)
        st[:, 3:6] = st[:, 0:3]
        st[:, 0] += np.float32(0.002)
        put(st.ravel())

    sample = [0, 1, 2, 5, 9, 14, 37, 58, 59]
    from oracle import OracleSim
    orcs = {k: OracleSim() for k in sample}

    def run_oracle(k):
        setup(orcs[k], k)
        for f in range(steps):
            if k % 4 == 2:
                move_spheres(orcs[k].get_shape_states, orcs[k].set_shape_states)
            orcs[k].step()
    threads = [threading.Thread(target=run_oracle, args=(k,)) for k in sample]
    [t.start() for t in threads]
    subset = [k for k in range(n_envs) if k % 7 != 3]
    states = {}
    for tp in ("1", "0"):
        os.environ["FLINGSIM_GRIDL_TP"] = tp          # read by the library at every launch sequence
        try:
            ctx = fsim.FlingSim(n_envs=n_envs, solver=fsim.FS_SOLVER_AUTO)
            for k in range(n_envs):
                setup(ctx.env(k), k)
            for f in range(steps):
                for k in range(2, n_envs, 4):
                    move_spheres(lambda k=k: ctx.get_shape_states(k), lambda s, k=k: ctx.set_shape_states(k, s))
                if f != 3:
                    ctx.step(1)
                    want = fsim.FS_FORM_STREAM_GRIDL_TP if tp == "1" else fsim.FS_FORM_STREAM_GRIDL
                    assert ctx.last_kernel_form() == want, (tp, ctx.last_kernel_form())
                else:
                    ctx.step_list(subset, 1)
                    ctx.step_list([k for k in range(n_envs) if k % 7 == 3], 1)
            if tp == "1":
                [t.join() for t in threads]
                for k in sample:
                    _assert_bits(ctx, k, orcs[k], f"throughput form, episode {k} ({dims[k % 3]})")
            states[tp] = [np.array(ctx.get_positions(k)).view(np.uint32).copy() for k in range(n_envs)]
            ctx.close()
        finally:
            os.environ.pop("FLINGSIM_GRIDL_TP", None)
    for k in range(n_envs):
        assert np.array_equal(states["1"][k], states["0"][k]), k
