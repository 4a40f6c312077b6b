"""Per-episode programs of the manipulation primitives and the scheduler that runs many of them on one simulator.

`primitives.FlingPrimitives` advances a batch of episodes phase by phase in lock step: every `movep` of a phase is one
`fs_movep_batch` call that lasts as long as its SLOWEST episode, and the stretch / lift loops, the five moves of the fling
and the settling all wait for each other.  In the evaluation loop (32 cloths of different sizes in different poses) that
leaves half of the episodes idle in the average launch sequence, and a launch sequence of the streaming back-end costs the
same ~1 ms whether 16 or 32 episodes take part in it (EXPERIMENTS.md, section 4.7 of the round-2 notes).

Here every episode runs the reference's straight-line code as its own coroutine -- the same statements in the same order
as environment/simEnv.py, one episode at a time --

    SimEnv.pick_and_fling_primitive     simEnv.py:283-318        pick_and_fling
    SimEnv.pick_and_drag_primitive      simEnv.py:320-345        pick_and_drag
    SimEnv.pick_and_place_primitive     simEnv.py:347-374        pick_and_place
    SimEnv.pick_stretch_drag_primitive  simEnv.py:376-428        pick_stretch_drag
    SimEnv.stretch_cloth / lift_cloth / fling_primitive          simEnv.py:140-200, 262-281
    SimEnv.postaction (reset_end_effectors + wait_until_stable)  simEnv.py:467-475, flex_utils.py:430-441

and yields a REQUEST wherever the reference touches the simulator: ("movep", targets, speed, min_steps, limit),
("wait", max_steps, tolerance), ("step", n), ("stats",) (cloth heights / speed), ("probe", midpoint_xz, height)
(stretch_cloth's test).
`run_programs` serves the requests of all episodes together: reductions in one batched call per kind, motion through
`FlingSim.advance` (fs_advance), which steps every episode through the next chunk of ITS OWN loop in shared launch sequences.
Episodes are independent and both loops are resumable without changing a bit (a movep iteration reads only the pickers'
current positions and its loop index, wait_until_stable only counts), so each episode gets exactly the trajectory the
lock-step primitives -- and the reference's sequential code -- give it; tests/ pin that against the same goldens.
"""
import numpy as np

from .sim import MoveLimitError


# ---- requests ------------------------------------------------------------------------------------------------------
def _movep(ep, targets, speed=None, min_steps=None, limit=1000):
    # dump_visualizations is off in batch mode: speed None -> 0.1 (simEnv.py:740-744)
    yield ("movep", np.array(targets), 0.1 if speed is None else speed, min_steps, limit)


class Episode:
    """What a program may touch of its episode between two requests: the grasp flags / terminate flag of the owning
    FlingPrimitives object and the host mirror of the picker states."""

    def __init__(self, prim, e):
        self.prim, self.e = prim, int(e)

    @property
    def grasp(self):
        return self.prim.grasp_states[self.e]

    def set_grasp(self, grasp):
        self.prim.set_grasp([self.e], grasp)

    def picker_positions(self):
        return self.prim.picker_positions(self.e)


# ---- simEnv.py:140-184
def stretch_cloth(ep, grasp_dist, fling_height=0.7, max_grasp_dist=0.7, increment_step=0.02):
    grasp_dist = np.float64(grasp_dist)
    left, right = ep.picker_positions()
    left[1] = fling_height
    right[1] = fling_height
    midpoint = (left + right) / 2
    direction = left - right
    direction = direction / np.linalg.norm(direction)
    yield from _movep(ep, [left, right], speed=5e-4, min_steps=20)  # float32 targets: movep's arithmetic stays float32
    stable_steps, cloth_midpoint = 0, 1e2
    while True:
        single, nearest = yield ("probe", midpoint[[0, 2]], np.float32(fling_height - 0.1))
        if single:  # single grasp
            return grasp_dist
        stable = np.linalg.norm(nearest - cloth_midpoint) < 1.5e-2
        stable_steps = stable_steps + 1 if stable else 0
        if stable_steps > 2:
            return grasp_dist
        cloth_midpoint = nearest
        grasp_dist += increment_step
        # float64 targets, spelled out (NumPy >= 2 promotion, see primitives.FlingPrimitives.stretch_cloth)
        mid64, dir64 = midpoint.astype(np.float64), direction.astype(np.float64)
        left = mid64 + dir64 * np.float64(grasp_dist) / 2
        right = mid64 - dir64 * np.float64(grasp_dist) / 2
        yield from _movep(ep, [left, right], speed=5e-4)
        if grasp_dist > max_grasp_dist:
            return max_grasp_dist


# ---- simEnv.py:186-200
def lift_cloth(ep, grasp_dist, fling_height=0.7, increment_step=0.05, max_height=0.7):
    while True:
        stats = yield ("stats",)
        if stats[0] > 0.02:  # heights.min() > 0.02
            return fling_height
        fling_height += increment_step
        yield from _movep(ep, [[grasp_dist / 2, fling_height, -0.3], [-grasp_dist / 2, fling_height, -0.3]], speed=1e-3)
        if fling_height >= max_height:
            return fling_height


def reset_end_effectors(ep):
    yield from _movep(ep, [[0.5, 0.5, -0.5], [-0.5, 0.5, -0.5]], speed=5e-3)


# ---- simEnv.py:262-281
def fling_primitive(ep, dist, fling_height, fling_speed, grasp_height):
    x, gh2 = dist / 2, grasp_height * 2
    yield from _movep(ep, [[x, fling_height, -0.2], [-x, fling_height, -0.2]], speed=fling_speed)
    yield from _movep(ep, [[x, fling_height, 0.2], [-x, fling_height, 0.2]], speed=fling_speed)
    yield from _movep(ep, [[x, fling_height, 0.2], [-x, fling_height, 0.2]], speed=1e-2, min_steps=4)
    yield from _movep(ep, [[x, gh2, -0.2], [-x, gh2, -0.2]], speed=1e-2)      # lower
    yield from _movep(ep, [[x, gh2, -0.25], [-x, gh2, -0.25]], speed=5e-3)
    ep.set_grasp(False)                                                         # release
    yield from reset_end_effectors(ep)


# ---- simEnv.py:283-318
def pick_and_fling(ep, p1, p2, p1_grasp_cloth, p2_grasp_cloth):
    prim = ep.prim
    out = dict(dist=None, fling_height=None, terminated=False, skipped=False)
    if not (p1_grasp_cloth or p2_grasp_cloth):
        out["skipped"] = True  # both points not on cloth
        return out
    p1, p2 = np.array(p1, np.float64), np.array(p2, np.float64)
    p1[1] = prim.grasp_height
    p2[1] = prim.grasp_height
    dist = np.linalg.norm(np.array(p1) - np.array(p2))
    yield from _movep(ep, [p1, p2])
    prim.grasp_states[ep.e] = [bool(p1_grasp_cloth), bool(p2_grasp_cloth)]  # only grasp points on cloth
    yield from _movep(ep, [[dist / 2, 0.3, -0.3], [-dist / 2, 0.3, -0.3]], speed=5e-3)  # lift to prefling
    stats = yield ("stats",)
    if not stats[1] > 0.2:  # is_cloth_grasped: heights.max() > 0.2
        prim.terminate[ep.e] = True
        out["terminated"] = True
        return out
    d = yield from stretch_cloth(ep, dist, fling_height=0.3)
    if prim.fixed_fling_height == -1:
        h = yield from lift_cloth(ep, d, fling_height=0.3)
    else:
        h = prim.fixed_fling_height
    yield from fling_primitive(ep, d, h, prim.fling_speed, prim.grasp_height)
    out["dist"], out["fling_height"] = d, h
    return out


_PARK = [-0.2, 0.3, -0.2]


def _at(pos, h=None):
    q = pos.copy()
    if h is not None:
        q[1] = h
    return [q, _PARK]


# ---- simEnv.py:320-345
def pick_and_drag(ep, p1, p2, p1_grasp_cloth, p2_grasp_cloth=None):
    out = dict(skipped=not bool(p1_grasp_cloth))
    if out["skipped"]:  # first grasp point not on cloth -> nothing happens
        return out
    p1, p2 = np.array(p1, np.float64), np.array(p2, np.float64)
    p1[1] = ep.prim.grasp_height
    p2[1] = ep.prim.grasp_height
    yield from _movep(ep, _at(p1, 0.3), speed=5e-3)   # prestart
    yield from _movep(ep, _at(p1), speed=5e-3)
    ep.set_grasp(True)
    yield from _movep(ep, _at(p2), speed=5e-3)
    ep.set_grasp(False)
    yield from _movep(ep, _at(p2, 0.3), speed=5e-3)   # postend
    yield from reset_end_effectors(ep)
    return out


# ---- simEnv.py:347-374
def pick_and_place(ep, p1, p2, p1_grasp_cloth, p2_grasp_cloth=None, lift_height=0.2):
    out = dict(skipped=not bool(p1_grasp_cloth))
    if out["skipped"]:
        return out
    p1, p2 = np.array(p1, np.float64), np.array(p2, np.float64)
    p1[1] = ep.prim.grasp_height
    p2[1] = ep.prim.grasp_height
    yield from _movep(ep, _at(p1, lift_height), speed=5e-3)   # prepick
    yield from _movep(ep, _at(p1), speed=5e-3)
    ep.set_grasp(True)
    yield from _movep(ep, _at(p1, lift_height), speed=5e-3)
    yield from _movep(ep, _at(p2, lift_height), speed=5e-3)   # preplace
    yield from _movep(ep, _at(p2), speed=5e-3)
    ep.set_grasp(False)
    yield from _movep(ep, _at(p2, lift_height), speed=5e-3)
    yield from reset_end_effectors(ep)
    return out


# ---- simEnv.py:376-428
def pick_stretch_drag(ep, p1, p2, p1_grasp_cloth, p2_grasp_cloth):
    prim = ep.prim
    out = dict(skipped=not (bool(p1_grasp_cloth) or bool(p2_grasp_cloth)), dist=None)
    if out["skipped"]:
        return out
    p1, p2 = np.array(p1, np.float64), np.array(p2, np.float64)
    p1[1] = prim.grasp_height
    p2[1] = prim.grasp_height

    def raised(pos, h):
        q = pos.copy()
        q[1] = h
        return q

    yield from _movep(ep, [raised(p1, 0.3), raised(p2, 0.3)])
    yield from _movep(ep, [p1, p2], speed=2e-3)
    prim.grasp_states[ep.e] = [bool(p1_grasp_cloth), bool(p2_grasp_cloth)]  # only grasp points on cloth
    dist = np.linalg.norm(np.array(p1) - np.array(p2))
    if all(prim.grasp_states[ep.e]):  # stretch if cloth is grasped by both
        dist = yield from stretch_cloth(ep, dist, fling_height=prim.grasp_height)
    drag_direction = np.cross(p1 - p2, np.array([0, 1, 0]))
    drag_direction = prim.stretchdrag_dist * drag_direction / np.linalg.norm(drag_direction)
    left_start, right_start = ep.picker_positions()  # float32 rows; + float64 direction -> float64
    left_end = left_start + drag_direction
    right_end = right_start + drag_direction
    left_end[1] += 0.1  # prevent ee go under cloth
    right_end[1] += 0.1
    left_post, right_post = left_end.copy(), right_end.copy()
    left_post[1] = 0.3
    right_post[1] = 0.3
    out["dist"] = dist
    yield from _movep(ep, [left_end, right_end], speed=2e-3)
    ep.set_grasp(False)
    yield from _movep(ep, [left_post, right_post])
    yield from reset_end_effectors(ep)
    return out


# ---- SimEnv.postaction's simulation part (simEnv.py:467-469)
def settle(ep, max_steps=300, tolerance=1e-2):
    yield from reset_end_effectors(ep)
    result = yield ("wait", max_steps, tolerance)
    return result


def action_then_settle(ep, program, max_steps=300, tolerance=1e-2):
    """One episode's share of SimEnv.step between preaction and the coverage reward: the action handler (a program above,
    or None when no valid action was found) followed by postaction's reset_end_effectors + wait_until_stable."""
    out = None
    if program is not None:
        out = yield from program
    yield from settle(ep, max_steps, tolerance)
    return out


PROGRAMS = {"fling": pick_and_fling, "drag": pick_and_drag, "place": pick_and_place, "stretchdrag": pick_stretch_drag}


# ---- the scheduler ---------------------------------------------------------------------------------------------------
def _serve_stats(sim, reqs):
    rows = sim.cloth_stats([e for e, _ in reqs])
    return [rows[k] for k in range(len(reqs))]


def _serve_probe(sim, reqs):
    single, nearest = sim.stretch_probe([e for e, _ in reqs], [a[0] for _, a in reqs], [a[1] for _, a in reqs])
    return [(bool(single[k]), nearest[k]) for k in range(len(reqs))]


def _request_record(prim, e, req, table):
    """The scheduler's record of a program's request (see run_programs for the request forms)."""
    if req[0] == "movep":
        _, targets, speed, min_steps, limit = req
        return dict(kind=0, targets=np.asarray(targets, np.float64).reshape(-1, 3), f32=int(targets.dtype == np.float32),
                    grasp=[int(bool(g)) for g in prim.grasp_states[e]], speed=float(speed),
                    min_steps=-1 if min_steps is None else int(min_steps), limit=int(limit), start=0, steps=0)
    if req[0] == "wait":
        return dict(kind=1, limit=int(req[1]), tolerance=float(req[2]), start=0, steps=0, known=0, open=0)
    if req[0] == "step":
        return dict(kind=2, limit=int(req[1]), start=0, steps=0, known=0, open=0)
    if req[0] in table:
        return dict(kind=req[0], args=req[1:], after=None)
    raise ValueError(f"run_programs: unknown request {req[0]!r} from episode {e}")


def run_programs(prim, programs, cap_min=8, cap=64, eps=1e-4, services=None, pipeline=False, depth=2, run_ahead=False):
    """Run {episode: generator} to completion on prim.sim; returns {episode: the program's return value}.  Simulation steps
    are added to prim.sim_steps.  cap_min / cap: bounds of one fs_advance chunk (see include/flingsim.h).
    Requests a program may yield: ("movep", targets, speed, min_steps, limit), ("wait", max_steps, tolerance),
    ("step", n) -- served together by FlingSim.advance -- and host-side ones, served in ONE batched call per kind for all the
    episodes that stand at it: "stats", "probe", plus whatever `services` adds ({kind: fn([(episode, args), ...]) -> results};
    flingbot_amd/evaluate.py registers observation, policy and reward services there).
    pipeline: queue the chunks through fs_advance_begin / fs_advance_end with up to `depth` of them open, and serve the
    host-side requests while they run (run_programs_pipelined below); same results, the device no longer waits for the host."""
    if pipeline:
        return run_programs_pipelined(prim, programs, cap_min=cap_min, cap=cap, eps=eps, services=services, depth=depth,
                                      run_ahead=run_ahead)
    sim = prim.sim
    table = {"stats": lambda reqs: _serve_stats(sim, reqs), "probe": lambda reqs: _serve_probe(sim, reqs)}
    table.update(services or {})
    gens = {int(e): g for e, g in programs.items()}
    results, pending = {}, {}

    def resume(e, value):
        try:
            req = gens[e].send(value)
        except StopIteration as stop:
            results[e] = stop.value
            pending.pop(e, None)
            return
        pending[e] = _request_record(prim, e, req, table)

    for e in sorted(gens):
        resume(e, None)
    while pending:
        # host-side requests first, one batched call per kind, until every episode waits for simulation steps
        while True:
            kinds = sorted({r["kind"] for r in pending.values() if not isinstance(r["kind"], int)})
            if not kinds:
                break
            for kind in kinds:
                who = sorted(e for e, r in pending.items() if r["kind"] == kind)
                if not who:
                    continue
                out = table[kind]([(e, pending[e]["args"]) for e in who])
                for k, e in enumerate(who):
                    resume(e, out[k])
        if not pending:
            break
        order = sorted(pending)
        reqs = [pending[e] for e in order]
        n_shapes = max([r["targets"].shape[0] for r in reqs if r["kind"] == 0], default=2)
        zeros, nog = np.zeros((n_shapes, 3)), [0] * n_shapes
        prog, status, steps = sim.advance(
            order, [r["kind"] for r in reqs], [r["targets"] if r["kind"] == 0 else zeros for r in reqs],
            [r["grasp"] if r["kind"] == 0 else nog for r in reqs], [r.get("speed", 0.0) for r in reqs],
            [r["limit"] for r in reqs], [r.get("min_steps", -1) for r in reqs], [r.get("f32", 0) for r in reqs],
            [r["start"] for r in reqs], cap_min=cap_min, cap=cap, eps=eps, tolerance=[r.get("tolerance", -1.0) for r in reqs])
        prim.sim_steps += int(np.sum(steps))
        st = prim.__dict__.setdefault("sched_stats", dict(calls=0, sequences=0, episode_steps=0, slots=0))
        st["calls"] += 1
        st["sequences"] += int(np.max(steps))
        st["episode_steps"] += int(np.sum(steps))
        st["slots"] += int(np.max(steps)) * len(order)
        for k, e in enumerate(order):
            r = reqs[k]
            r["start"], r["steps"] = int(prog[k]), r["steps"] + int(steps[k])
            if status[k] == 0:
                continue
            if r["kind"] == 0:
                if status[k] == 2:
                    raise MoveLimitError(f"movep: step limit reached in episode {e} (MoveJointsException)")
                resume(e, None)
            elif r["kind"] == 1:
                resume(e, (status[k] == 1, r["steps"]))
            else:
                resume(e, None)
    return results


def run_programs_pipelined(prim, programs, cap_min=1, cap=4, eps=1e-4, services=None, depth=2, run_ahead=False):
    """run_programs with the device kept busy.  The blocking loop alternates "host serves requests / resumes programs" and
    "device runs a chunk": measured on the evaluation loop (192 tasks through 96 slots) the device idles 15 % of the wall
    time.  Here a chunk is QUEUED (fs_advance_begin) and the next one is queued behind it before the first has finished:

      * a movep's outcome is planned on the host, so when a chunk is queued the scheduler already knows which moveps end in
        it, where the others stand, and -- run_ahead=True -- what the finishing programs ask for next (their code between two
        requests must then be host-only: true for the programs of this module and BatchedFlingEnv.episode_program);
      * a wait_until_stable / plain-step loop is decided on the device: its state lives there across chunks, so the next
        chunk simply lists the episode again ("continue", start = -1); if the loop ended in the previous chunk, the entries
        retire at once.  The host learns the outcome when it closes the chunk (fs_advance_end);
      * host-side requests (reductions, observations, the policy, resets) are served while chunks run, on the context's
        service lane (fs_service_lane), for episodes whose last chunk has been closed.

    An episode that asks for a host-side service sits out the chunk(s) already queued -- which is why chunks are short here
    (cap_min / cap default 1 / 4 instead of 8 / 64: the per-call host round trip that long chunks amortise is hidden).
    Every episode still receives exactly its own sequence of simulation steps: results equal run_programs' bit for bit."""
    from collections import deque

    sim = prim.sim
    table = {"stats": lambda reqs: _serve_stats(sim, reqs), "probe": lambda reqs: _serve_probe(sim, reqs)}
    table.update(services or {})
    gens = {int(e): g for e, g in programs.items()}
    results, pending = {}, {}
    open_tickets = deque()
    st = prim.__dict__.setdefault("sched_stats", dict(calls=0, sequences=0, episode_steps=0, slots=0))
    seq0 = sim.advance_timing()["sequences"]
    lane = [False]

    def set_lane(on):
        if lane[0] != on:
            sim.service_lane(on)
            lane[0] = on

    def resume(e, value, after=None):
        """Send `value` into episode e's program and file its next request.  after: an open chunk the program's previous
        simulation request belongs to -- a host-side request then waits until that chunk is closed."""
        set_lane(bool(open_tickets))  # whatever the program touches between two requests must not queue behind a chunk
        try:
            req = gens[e].send(value)
        except StopIteration as stop:
            results[e] = stop.value
            pending.pop(e, None)
            return
        pending[e] = _request_record(prim, e, req, table)
        if isinstance(pending[e]["kind"], str):
            pending[e]["after"] = after

    def serve_host():
        """One batched call per kind for every episode that stands at a host-side request and is not waiting for an open
        chunk; repeated until none is left (a served program may ask for the next service right away)."""
        while True:
            ready = [e for e, r in pending.items() if isinstance(r["kind"], str) and (r["after"] is None or r["after"]["closed"])]
            if not ready:
                return
            set_lane(bool(open_tickets))
            for kind in sorted({pending[e]["kind"] for e in ready}):
                # (a program served under an earlier kind of this pass may have ended or moved on to another request)
                who = sorted(e for e in ready if e in pending and pending[e]["kind"] == kind and
                             (pending[e]["after"] is None or pending[e]["after"]["closed"]))
                if not who:
                    continue
                out = table[kind]([(e, pending[e]["args"]) for e in who])
                for k, e in enumerate(who):
                    resume(e, out[k])

    def begin_chunk():
        order = sorted(e for e, r in pending.items() if isinstance(r["kind"], int))
        if not order:
            return False
        reqs = [pending[e] for e in order]
        n_shapes = max([r["targets"].shape[0] for r in reqs if r["kind"] == 0], default=2)
        zeros, nog = np.zeros((n_shapes, 3)), [0] * n_shapes
        set_lane(False)  # chunks go to the main stream, ordered behind whatever the service lane queued
        ticket, prog, status, steps = sim.advance_begin(
            order, [r["kind"] for r in reqs], [r["targets"] if r["kind"] == 0 else zeros for r in reqs],
            [r["grasp"] if r["kind"] == 0 else nog for r in reqs], [r.get("speed", 0.0) for r in reqs],
            [r["limit"] for r in reqs], [r.get("min_steps", -1) for r in reqs], [r.get("f32", 0) for r in reqs],
            [r["start"] if (r["kind"] == 0 or r["open"] == 0) else -1 for r in reqs], cap_min=cap_min, cap=cap, eps=eps,
            tolerance=[r.get("tolerance", -1.0) for r in reqs])
        T = dict(ticket=ticket, prog=prog, status=status, steps=steps, waiters=[], finishers=[], closed=False)
        open_tickets.append(T)
        st["calls"] += 1
        for k, e in enumerate(order):
            r = reqs[k]
            if r["kind"] == 0:
                r["start"], r["steps"] = int(prog[k]), r["steps"] + int(steps[k])
                prim.sim_steps += int(steps[k])
                st["episode_steps"] += int(steps[k])
                if status[k] == 0:
                    continue
                if status[k] == 2:
                    raise MoveLimitError(f"movep: step limit reached in episode {e} (MoveJointsException)")
                if run_ahead:
                    resume(e, None, after=T)
                else:
                    T["finishers"].append(e)
                    pending[e] = dict(kind="__finished__", after=T, args=())
            elif status[k] != -1:  # the loop's steps were used up before the call: answered without the device
                resume(e, (False, r["known"]) if r["kind"] == 1 else None)
            else:
                r["open"] += 1
                T["waiters"].append((k, e, r))
        return True

    def end_chunk():
        T = open_tickets.popleft()
        sim.advance_end(T["ticket"], T["prog"], T["status"], T["steps"])
        T["closed"] = True
        for k, e, r in T["waiters"]:
            r["open"] -= 1
            if pending.get(e) is not r:
                continue  # the loop ended in an earlier chunk; this one only carried its retired entries
            total = int(T["prog"][k])
            prim.sim_steps += total - r["known"]
            st["episode_steps"] += total - r["known"]
            r["known"] = r["start"] = total
            if T["status"][k] == 0:
                continue
            resume(e, (T["status"][k] == 1, total) if r["kind"] == 1 else None)
        for e in T["finishers"]:
            resume(e, None)

    try:
        for e in sorted(gens):
            resume(e, None)
        while pending or open_tickets:
            serve_host()
            if len(open_tickets) < depth and begin_chunk():
                continue
            if open_tickets:
                end_chunk()
            elif pending and not any(isinstance(r["kind"], int) for r in pending.values()):
                serve_host()
    finally:
        while open_tickets:  # (after an exception: nothing may stay in flight)
            T = open_tickets.popleft()
            try:
                sim.advance_end(T["ticket"], T["prog"], T["status"], T["steps"])
            except Exception:
                pass
            T["closed"] = True
        set_lane(False)
        st["sequences"] += int(sim.advance_timing()["sequences"] - seq0)
    return results
