"""Per-episode programs on shared launch sequences (flingbot_amd/schedule.py + fs_advance) on the device: the goldens
recorded from the REFERENCE's SimEnv methods (fling / drag / place / stretchdrag / SimEnv.step) bit for bit, and equality
with the lock-step primitives on a batch of cloths of different sizes in different poses."""
import numpy as np
import pytest

from fling_helpers import load_fling_golden, load_primitives_golden, load_step_golden, run_primitives_golden, run_step_golden
from test_fling_gpu import _make

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("caps", [(8, 64), (1, 3), (40, 40)])
def test_scheduled_pick_and_fling_matches_reference_golden(gpu_required, caps):
    """Every golden case as its own program, followed -- inside the same scheduling run -- by postaction's
    reset_end_effectors + wait_until_stable; chunk bounds from "one step at a time" to "longer than most moves"."""
    from flingbot_amd.primitives import FlingPrimitives

    g = load_fling_golden()
    n = len(g["terminate"])
    ctx = _make(g, n)
    prim = FlingPrimitives(ctx, range(n))
    out, _ = prim.act_scheduled({e: ("fling", g["p1"][e], g["p2"][e], g["g1"][e], g["g2"][e]) for e in range(n)},
                                settle=False, cap_min=caps[0], cap=caps[1])
    for e in range(n):
        assert out[e]["terminated"] == bool(g["terminate"][e]), e
        if np.isnan(g["stretch_ret"][e]):
            assert out[e]["dist"] is None
        else:
            assert out[e]["dist"] == g["stretch_ret"][e] and out[e]["fling_height"] == g["lift_ret"][e], e
        assert np.array_equal(ctx.get_positions(e).view(np.uint32), g["pos_fling"][e].view(np.uint32)), e
    assert out[3]["skipped"]
    # the lock-step primitives count the same simulation steps
    ctx2 = _make(g, n)
    prim2 = FlingPrimitives(ctx2, range(n))
    prim2.pick_and_fling(g["p1"], g["p2"], g["g1"], g["g2"])
    assert prim.sim_steps == prim2.sim_steps > 0
    for e in range(n):
        assert np.array_equal(ctx.get_shape_states(e).view(np.uint32), ctx2.get_shape_states(e).view(np.uint32)), e
        assert ctx.picked(e).tolist() == ctx2.picked(e).tolist(), e


def test_advance_waiters_match_wait_until_stable_golden(gpu_required):
    """fs_advance's wait_until_stable leg alone, resumed across chunks of 7 steps: the golden's step counts, stability flags
    and final states (episodes that settle after different numbers of steps, one that runs into max_steps)."""
    from flingbot_amd.primitives import FlingPrimitives

    g = load_fling_golden()
    n = len(g["terminate"])
    ctx = _make(g, n)
    prim = FlingPrimitives(ctx, range(n))
    prim.pick_and_fling(g["p1"], g["p2"], g["g1"], g["g2"])
    for e in range(n):
        lifted = ctx.get_positions(e).reshape(-1, 4).copy()
        lifted[:, 1] += np.float32(0.25)
        ctx.set_positions(e, lifted.ravel())
        vel = np.zeros((lifted.shape[0], 3), np.float32)
        vel[:, 1] = -0.5
        ctx.set_velocities(e, vel.ravel())
    start, total, done = np.zeros(n, np.int32), np.zeros(n, np.int32), {}
    z, calls = np.zeros((n, 2, 3)), 0
    live = list(range(n))
    while live:
        k = len(live)
        prog, status, steps = ctx.advance(live, [1] * k, z[:k], np.zeros((k, 2), int), [0.0] * k, [200] * k, [-1] * k, [0] * k,
                                          start[live], cap_min=7, cap=7, tolerance=2e-2)
        calls += 1
        for q, e in enumerate(list(live)):
            start[e] = prog[q]
            total[e] += steps[q]
            if status[q] != 0:
                done[e] = status[q] == 1
                live.remove(e)
    assert calls > 3
    assert total.tolist() == g["steps_drop"].tolist() and [done[e] for e in range(n)] == g["stable_drop"].tolist()
    for e in range(n):
        assert np.array_equal(ctx.get_positions(e).view(np.uint32), g["pos_final"][e].view(np.uint32)), e
        assert np.array_equal(ctx.get_shape_states(e).view(np.uint32), g["shapes_final"][e].view(np.uint32)), e
    # a budget that is already used up: finished at the limit without a step or a test
    prog, status, steps = ctx.advance([0], [1], z[:1], [[0, 0]], [0.0], [5], [-1], [0], [5], tolerance=2e-2)
    assert (int(prog[0]), int(status[0]), int(steps[0])) == (5, 2, 0)


@pytest.mark.parametrize("depth", [2, 3])
def test_advance_begin_end_waiters_queued_ahead_match_golden(gpu_required, depth):
    """The same wait_until_stable golden through fs_advance_begin / fs_advance_end with `depth` chunks open: every chunk after
    an episode's first lists it with start = -1 (continue from the loop state on the device) BEFORE the previous chunk has
    reported, so an episode that settles in chunk k still has entries in the chunks queued behind it -- they must retire
    without stepping.  While chunks are in flight the service lane serves another context-level request (cloth statistics
    of a finished episode).  Step counts, stability flags and final states equal the golden's bit for bit."""
    from collections import deque
    from flingbot_amd.primitives import FlingPrimitives

    g = load_fling_golden()
    n = len(g["terminate"])
    ctx = _make(g, n)
    prim = FlingPrimitives(ctx, range(n))
    prim.pick_and_fling(g["p1"], g["p2"], g["g1"], g["g2"])
    for e in range(n):
        lifted = ctx.get_positions(e).reshape(-1, 4).copy()
        lifted[:, 1] += np.float32(0.25)
        ctx.set_positions(e, lifted.ravel())
        vel = np.zeros((lifted.shape[0], 3), np.float32)
        vel[:, 1] = -0.5
        ctx.set_velocities(e, vel.ravel())
    z = np.zeros((n, 2, 3))
    live, first, total, done = list(range(n)), {e: True for e in range(n)}, {}, {}
    open_chunks, chunks, lane_calls = deque(), 0, 0
    while live or open_chunks:
        if live and len(open_chunks) < depth:
            k = len(live)
            start = [0 if first[e] else -1 for e in live]
            ticket, prog, status, steps = ctx.advance_begin(live, [1] * k, z[:k], np.zeros((k, 2), int), [0.0] * k, [200] * k,
                                                            [-1] * k, [0] * k, start, cap_min=7, cap=7, tolerance=2e-2)
            assert (status == -1).all()  # a wait's outcome is the device's to decide
            for e in live:
                first[e] = False
            open_chunks.append((ticket, list(live), prog, status, steps))
            chunks += 1
            assert ctx.advance_in_flight() == len(open_chunks)
            continue
        ticket, who, prog, status, steps = open_chunks.popleft()
        if done:  # host-side work for an episode that is NOT part of the running chunks any more, on the service lane
            ctx.service_lane(True)
            e0 = next(iter(done))
            assert ctx.cloth_stats([e0]).shape == (1, 3)
            ctx.service_lane(False)
            lane_calls += 1
        ctx.advance_end(ticket, prog, status, steps)
        for q, e in enumerate(who):
            if e in done:
                assert status[q] != 0 and int(prog[q]) == total[e]  # a stale entry repeats the final report
                continue
            total[e] = int(prog[q])
            if status[q] != 0:
                done[e] = status[q] == 1
                live.remove(e)
    assert ctx.advance_in_flight() == 0 and chunks > 3 and lane_calls > 0
    assert [total[e] for e in range(n)] == g["steps_drop"].tolist() and [done[e] for e in range(n)] == g["stable_drop"].tolist()
    for e in range(n):
        assert np.array_equal(ctx.get_positions(e).view(np.uint32), g["pos_final"][e].view(np.uint32)), e
        assert np.array_equal(ctx.get_shape_states(e).view(np.uint32), g["shapes_final"][e].view(np.uint32)), e
    # protocol errors: chunks are queued on the main lane; a ticket is closed once; the blocking call refuses start = -1
    from flingbot_amd.sim import FlingSimError
    ctx.service_lane(True)
    with pytest.raises(FlingSimError):
        ctx.advance_begin([0], [2], z[:1], [[0, 0]], [0.0], [1], [-1], [0], [0])
    ctx.service_lane(False)
    with pytest.raises(FlingSimError):
        ctx.advance_end(0, np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32))
    with pytest.raises(FlingSimError):
        ctx.advance([0], [1], z[:1], [[0, 0]], [0.0], [5], [-1], [0], [-1])


def test_advance_begin_reports_movers_at_once(gpu_required):
    """fs_advance_begin's movep entries are final when it returns (the trajectory is planned on the host): progress, status
    and step counts equal the blocking fs_advance's on a twin context, chunk by chunk, and so do the final states; plain-step
    entries ride along and report at fs_advance_end."""
    g = load_fling_golden()
    a, b = _make(g, 2), _make(g, 2)
    targets = np.array([[[0.12, 0.25, -0.05], [-0.12, 0.25, -0.05]], [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]])
    grasp = [[0, 0], [0, 0]]
    start, wstart = 0, 0
    for _ in range(40):
        args = ([0, 1], [0, 2], targets, grasp, [5e-3, 0.0], [1000, 30], [-1, -1], [0, 0])
        pa, sa, ta = a.advance(*args, [start, wstart], cap_min=4, cap=6)
        ticket, pb, sb, tb = b.advance_begin(*args, [start, wstart], cap_min=4, cap=6)
        assert (int(pb[0]), int(sb[0]), int(tb[0])) == (int(pa[0]), int(sa[0]), int(ta[0]))
        assert int(sb[1]) == -1 or wstart >= 30
        b.advance_end(ticket, pb, sb, tb)
        assert pb.tolist() == pa.tolist() and sb.tolist() == sa.tolist() and tb.tolist() == ta.tolist()
        start, wstart = int(pa[0]), int(pa[1])
        if sa[0] != 0:
            break
    assert sa[0] == 1 and start > 10
    for e in range(2):
        assert np.array_equal(a.get_positions(e).view(np.uint32), b.get_positions(e).view(np.uint32))
        assert np.array_equal(a.get_shape_states(e).view(np.uint32), b.get_shape_states(e).view(np.uint32))
    assert a.last_movep_steps_raw() == b.last_movep_steps_raw() > 0


def test_scheduled_drag_place_stretchdrag_match_reference_golden(gpu_required):
    """All cases of primitives_golden.npz -- three different primitives -- in ONE scheduling run."""
    g = load_primitives_golden()
    run_primitives_golden(lambda n: _make(g, n), lambda sim, k: sim.get_positions(k), lambda sim, k: sim.get_shape_states(k),
                          scheduled=True)


def test_scheduled_step_bookkeeping_matches_reference_golden(gpu_required):
    """SimEnv.step's golden through BatchedFlingEnv's default (scheduled) execution on the device."""
    from flingbot_amd import sim as fsim

    g = load_step_golden()

    def make(n):
        ctx = fsim.FlingSim(n_envs=n, solver=0)
        for e in range(n):
            env = ctx.env(e)
            env.set_scene(g["scene_params"])
            env.step(1)
            env.set_positions(g["init_pos"].ravel())
            env.set_velocities(np.zeros(3 * g["init_pos"].shape[0], np.float32))
        return ctx

    run_step_golden(make, lambda sim, k: sim.get_positions(k), lambda sim, k: sim.get_shape_states(k), scheduled=True)


def test_scheduled_equals_lockstep_on_mixed_cloths(gpu_required):
    """Six generated tasks with cloth sides 40..70 (different particle counts, so different kernels' launch lists mix),
    each with its own fling; scheduled and lock-step execution of action + postaction give bit-identical particle
    states, picker states, terminate flags and simulation-step counts, and the scheduler needs fewer launch sequences."""
    import random

    from flingbot_amd import sim as fsim, tasks as ftasks
    from flingbot_amd.primitives import FlingPrimitives

    def build():
        random.seed(5)
        np.random.seed(5)
        n = 6
        gen = fsim.FlingSim(n_envs=n, solver=0)
        tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters(min_cloth_size=40, strict_min_edge_length=40,
                                                                        max_cloth_size=70) for _ in range(n)])
        gen.close()
        ctx = fsim.FlingSim(n_envs=n, solver=0)
        envs = ftasks.load_tasks(ctx, tasks)
        prim = FlingPrimitives(ctx, envs)
        prim.setup_pickers()
        return ctx, prim, n

    def grasp_points(ctx, e, k):
        pos = ctx.get_positions(e).reshape(-1, 4)[:, :3]
        order = np.argsort(pos[:, 0])
        a, b = pos[order[3 + k]], pos[order[-4 - k]]  # two particles near the opposite ends of the heap
        return np.array([a[0], 0.0, a[2]], np.float64), np.array([b[0], 0.0, b[2]], np.float64)

    results = []
    for scheduled in (False, True):
        ctx, prim, n = build()
        pts = [grasp_points(ctx, e, e) for e in range(n)]
        g1 = [True] * n
        g2 = [True, True, False, True, True, True]  # one one-handed fling
        prim.preaction()
        if scheduled:
            prim.act_scheduled({e: ("fling", pts[e][0], pts[e][1], g1[e], g2[e]) for e in range(n)})
        else:
            prim.pick_and_fling([p[0] for p in pts], [p[1] for p in pts], g1, g2)
            prim.postaction()
        results.append(dict(pos=[ctx.get_positions(e).copy() for e in range(n)],
                            vel=[ctx.get_velocities(e).copy() for e in range(n)],
                            shapes=[np.array(ctx.get_shape_states(e)).copy() for e in range(n)],
                            terminate=dict(prim.terminate), steps=prim.sim_steps, cov=np.array(ctx.coverage())))
        ctx.close()
    a, b = results
    assert a["steps"] == b["steps"] > 1000
    assert a["terminate"] == b["terminate"]
    assert np.array_equal(a["cov"], b["cov"])
    for e in range(len(a["pos"])):
        assert np.array_equal(a["pos"][e].view(np.uint32), b["pos"][e].view(np.uint32)), e
        assert np.array_equal(a["vel"][e].view(np.uint32), b["vel"][e].view(np.uint32)), e
        assert np.array_equal(a["shapes"][e].view(np.uint32), b["shapes"][e].view(np.uint32)), e
    assert len({p.size for p in a["pos"]}) > 1  # the cloths really differ in size


def test_advance_argument_errors(gpu_required):
    """fs_advance refuses what it cannot run: an episode listed twice, an unknown request kind, chunk bounds out of order, a
    movep before fs_picker_reset -- FlingSimError, and the episodes are left untouched."""
    from flingbot_amd import sim as fsim

    g = load_fling_golden()
    ctx = _make(g, 2)
    before = [ctx.get_positions(e).copy() for e in range(2)]
    z = np.zeros((2, 2, 3))
    gr = np.zeros((2, 2), int)
    with pytest.raises(fsim.FlingSimError):
        ctx.advance([0, 0], [1, 1], z, gr, [0.0, 0.0], [5, 5], [-1, -1], [0, 0], [0, 0])
    with pytest.raises(fsim.FlingSimError):
        ctx.advance([0, 1], [1, 7], z, gr, [0.0, 0.0], [5, 5], [-1, -1], [0, 0], [0, 0])
    with pytest.raises(fsim.FlingSimError):
        ctx.advance([0, 1], [1, 1], z, gr, [0.0, 0.0], [5, 5], [-1, -1], [0, 0], [0, 0], cap_min=9, cap=4)
    fresh = fsim.FlingSim(n_envs=1, solver=0)
    fresh.env(0).set_scene(g["scene_params"])
    with pytest.raises(fsim.FlingSimError):
        fresh.advance([0], [0], z[:1], gr[:1], [0.1], [5], [-1], [0], [0])  # no pickers set up
    fresh.close()
    for e in range(2):
        assert np.array_equal(ctx.get_positions(e), before[e])
    # plain steps: kind 2 takes exactly `limit` frames, resumable
    prog, status, steps = ctx.advance([0, 1], [2, 2], z, gr, [0.0, 0.0], [3, 9], [-1, -1], [0, 0], [0, 0], cap_min=4, cap=4)
    assert prog.tolist() == [3, 4] and status.tolist() == [1, 0] and steps.tolist() == [3, 4]
    prog, status, steps = ctx.advance([1], [2], z[:1], gr[:1], [0.0], [9], [-1], [0], [4], cap_min=8, cap=8)
    assert prog.tolist() == [9] and status.tolist() == [1] and steps.tolist() == [5]
    ref = _make(g, 2)
    ref.step_list([0], 3)
    ref.step_list([1], 9)
    for e in range(2):
        assert np.array_equal(ctx.get_positions(e).view(np.uint32), ref.get_positions(e).view(np.uint32)), e


def test_service_lane_refuses_to_rewrite_an_episode_in_flight(gpu_required):
    """The service lane's contract -- only episodes that are NOT part of a chunk in flight -- is checked where it would
    corrupt state silently: a setter / scene load / picker reset for a listed episode raises there, the same call for another
    episode works, and after fs_advance_end (or on the main lane, which waits for the chunk) the episode can be written."""
    from flingbot_amd import sim as fsim

    g = load_fling_golden()
    ctx = _make(g, 2)
    z, gr = np.zeros((1, 2, 3)), np.zeros((1, 2), int)
    ticket, prog, status, steps = ctx.advance_begin([0], [2], z, gr, [0.0], [6], [-1], [0], [0], cap_min=6, cap=6)
    pos1 = ctx.get_positions(1).copy()
    ctx.service_lane(True)
    try:
        for call in (lambda: ctx.set_positions(0, pos1), lambda: ctx.set_velocities(0, np.zeros(pos1.size // 4 * 3, np.float32)),
                     lambda: ctx.env(0).set_scene(g["scene_params"]), lambda: ctx.picker_reset(0, 0.005, 0.00625)):
            with pytest.raises(fsim.FlingSimError, match="in flight"):
                call()
        ctx.set_positions(1, pos1)  # the episode next to it is free
        # ... but nothing may be STEPPED on the lane while a chunk is in flight, not even the free episode: a launch sequence
        # rebuilds per-context tables the running chunk reads (include/flingsim.h, fs_service_lane)
        for call in (lambda: ctx.step_list([1], 1), lambda: ctx.step(1), lambda: ctx.wait_until_stable([1], 3, 1e-2),
                     lambda: ctx.movep([1], z, gr, speed=0.01, limit=10)):
            with pytest.raises(fsim.FlingSimError, match="service lane"):
                call()
    finally:
        ctx.service_lane(False)
    prog, status, steps = ctx.advance_end(ticket, prog, status, steps)
    assert prog.tolist() == [6] and status.tolist() == [1]
    ref = _make(g, 2)
    ref.step_list([0], 6)
    assert np.array_equal(ctx.get_positions(0).view(np.uint32), ref.get_positions(0).view(np.uint32))  # the refused calls changed nothing
    ctx.service_lane(True)
    ctx.set_positions(0, pos1)  # nothing in flight any more
    ctx.service_lane(False)
    assert np.array_equal(ctx.get_positions(0), pos1)
