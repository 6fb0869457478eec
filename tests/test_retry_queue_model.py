"""A model of the frame kernel's hand-on protocol (mdec_kernels.hip: draw_ticket / hand_on / end_of_frame / the pop phase at the
top of the frame loop), run under random interleavings.

The GPU tests check bytes; they cannot choose the schedule.  Here every workgroup is a little state machine that performs ONE
shared-memory operation per step -- the same operations on the same words as the kernel (one 64-bit state word: fresh tickets |
slots reserved | pop tickets; the slot array; the exit counter with its abandonment notes; since mdec-k3.7 a ticket is a RUN of 1, 2 or 4
consecutive frames, drawn when the group enters the last frame of the run in hand, and a launch may have groups without a ticket) -- and a seeded scheduler picks who
moves next, including workgroups that start late (not resident: the reason the wait is bounded) and waiting workgroups that
run out of patience.  Properties: every frame is encoded to the end exactly once, every workgroup leaves, nobody waits for a
slot that is never filled, and the last one out leaves counters and slots as the next launch expects them."""
import random

import pytest

EMPTY, ABANDONED = 0xFFFFFFFF, 0xFFFFFFFE


def ticket_plan(n_frames, groups, max_run=4):
    """psxhip_mdec_ticket_plan (mdec_kernels.hip): whole rounds of the grid in runs of 4, then of 2; the remainder as one round of
    runs of 2 when it is more than a frame per group, else single frames.  Returns (t4, t2, n_tickets)."""
    r, a4, a2 = n_frames, 0, 0
    if max_run >= 4:
        a4 = groups * (r // (4 * groups)); r -= 4 * a4
    if max_run >= 2:
        whole = groups * (r // (2 * groups)); a2 = whole; r -= 2 * whole
        if r > groups:
            a2 += r // 2; r &= 1
    return a4, a2, a4 + a2 + r


def ticket_run(plan, t):
    t4, t2, _ = plan
    if t < t4:
        return 4 * t, 4
    if t < t4 + t2:
        return 4 * t4 + 2 * (t - t4), 2
    return 4 * t4 + 2 * t2 + (t - t4 - t2), 1


class Shared:
    def __init__(self, n_frames, grid, cap, max_run=4, groups_max=None):
        self.tickets = self.reserved = self.head = 0          # the state word's three fields (one atomic word in the kernel)
        self.slots = [EMPTY] * cap
        self.left = 0                                         # groups gone | abandonments << 16
        self.started = 0                                      # groups that have started
        self.n = n_frames
        self.plan = ticket_plan(n_frames, groups_max or grid, max_run)
        self.n_tickets = self.plan[2]
        self.grid = grid                                      # groups of the launch (may exceed the tickets: spare groups only take handed-on frames)
        self.encoded = [0] * n_frames                         # times a frame was encoded to the end
        self.passes = 0


def group(S, b, rng, p_wrong, patience, start_delay):
    """generator: yields once per shared-memory operation (and once per unit of local work)"""
    for _ in range(start_delay):
        yield "not started"
    fresh_draws = max(0, S.n_tickets - S.grid)
    ticket, retried, next_draw, queue = b, False, 0, None
    if ticket < S.n_tickets:
        frame, run = ticket_run(S.plan, ticket)
        run_left = run - 1
        if run == 1:
            next_draw = S.tickets; S.tickets += 1; yield "draw"               # atomicAdd(state, 1): entering the last frame of the run
    else:
        frame, run_left = None, 0
        h, reserved, tickets = S.head, S.reserved, S.tickets; S.head += 1; yield "pop ticket"
        queue = h if reserved > h else (-1 if tickets >= S.n_tickets else -2 - h)
    S.started += 1; yield "started"
    while True:
        if ticket < S.n_tickets or retried:
            # ---- one frame: passes until the search is done; a wrong first guess on a fresh frame may be handed on
            deferred = False
            wrong = (not retried) and rng.random() < p_wrong
            S.passes += 1; yield "pass"
            if wrong:
                if not retried and (run_left > 0 or next_draw < fresh_draws):     # hand_on(): this group holds a further fresh frame
                    slot = S.reserved; S.reserved += 1; yield "reserve"           # atomicAdd(state, 1 << 32)
                    old = S.slots[slot]; S.slots[slot] = frame | (5 << 24); yield "fill"      # atomicExch
                    if old == ABANDONED:
                        S.slots[slot] = EMPTY; yield "unfill"                     # the frame stays here
                    else:
                        assert old == EMPTY
                        deferred = True
                if not deferred:
                    S.passes += 1; yield "pass again"
            if not deferred:
                S.encoded[frame] += 1
            # ---- end_of_frame
            retried = False
            queue = None
            entered_last = False
            if run_left > 0 and ticket < S.n_tickets:
                run_left -= 1; frame += 1
                entered_last = run_left == 0
            else:
                ticket = next_draw + S.grid if ticket < S.n_tickets else 1 << 30
                if ticket < S.n_tickets:
                    frame, run = ticket_run(S.plan, ticket)
                    run_left = run - 1
                    entered_last = run == 1
                else:
                    run_left = 0
                    h, reserved, tickets = S.head, S.reserved, S.tickets; S.head += 1; yield "pop ticket"      # atomicAdd(state, 1 << 48)
                    queue = h if reserved > h else (-1 if tickets >= S.n_tickets else -2 - h)
            if entered_last:
                next_draw = S.tickets; S.tickets += 1; yield "draw"
            if ticket < S.n_tickets:
                continue
        # ---- no fresh frame: the pop phase
        if queue == -1:
            break
        h = queue if queue >= 0 else -2 - queue
        there = queue >= 0
        looks = 0
        while not there:
            reserved, tickets, started = S.reserved, S.tickets, S.started; yield "look"
            if reserved > h:
                there = True; break
            if tickets >= S.n_tickets:
                break
            if started < S.grid or looks >= patience:            # nobody waits while a group has yet to start
                old = S.slots[h]                                              # atomicCAS(slot, EMPTY, ABANDONED)
                if old == EMPTY:
                    S.slots[h] = ABANDONED; yield "abandon"
                    S.left += 0x10000; yield "note"
                else:
                    there = True; yield "abandon failed"
                break
            looks += 1
        if not there:
            break
        while S.slots[h] == EMPTY:
            yield "spin"
        v = S.slots[h]; yield "take"
        assert v not in (EMPTY, ABANDONED)
        S.slots[h] = EMPTY; yield "vacate"
        frame, retried = v & 0xFFFFFF, True
    # ---- exit: the last group re-arms
    left = S.left; S.left += 1; yield "leave"
    if (left & 0xFFFF) == S.grid - 1:
        S.left = 0
        if left >> 16:
            for i in range(S.reserved, min(S.head, len(S.slots))):
                S.slots[i] = EMPTY
        S.tickets = S.reserved = S.head = S.started = 0


@pytest.mark.parametrize("seed", range(60))
def test_every_frame_once_everybody_leaves_counters_rearmed(seed):
    rng = random.Random(seed)
    groups_max = rng.choice([2, 3, 8, 16])
    n = rng.randint(groups_max + 1, 8 * groups_max)
    max_run = rng.choice([1, 2, 4, 4])
    n_tickets = ticket_plan(n, groups_max, max_run)[2]
    # the host's grid: one group per ticket, at most groups_max -- or all of groups_max when runs left slots empty (spare groups)
    grid = min(n_tickets, groups_max)
    if n_tickets < groups_max and n > n_tickets and rng.random() < 0.7:
        grid = groups_max
    patience = rng.choice([0, 1, 3, 50])
    p_wrong = rng.choice([0.0, 0.2, 0.6, 1.0])
    late = rng.random() < 0.5                                   # some groups start long after the others (not resident)
    S = Shared(n, grid, cap=n + grid + 4, max_run=max_run, groups_max=groups_max)
    gens = [group(S, b, random.Random(seed * 1000 + b), p_wrong, patience, rng.randint(0, 400) if late and b % 2 else 0) for b in range(grid)]
    live = list(range(grid))
    steps = 0
    while live:
        i = rng.choice(live) if rng.random() < 0.9 else live[0]
        try:
            next(gens[i])
        except StopIteration:
            live.remove(i)
        steps += 1
        assert steps < 2_000_000, "somebody never leaves"
    assert S.encoded == [1] * n, (seed, [k for k, c in enumerate(S.encoded) if c != 1][:8])
    assert (S.tickets, S.reserved, S.head, S.left, S.started) == (0, 0, 0, 0, 0)
    assert all(s == EMPTY for s in S.slots), (seed, [(k, hex(s)) for k, s in enumerate(S.slots) if s != EMPTY][:4])


def test_handing_on_levels_the_groups():
    """what it is for: with every first guess wrong, groups without a fresh ticket take the restarts"""
    S = Shared(64, 8, cap=80)
    gens = [group(S, b, random.Random(b), 1.0, 10 ** 6, 0) for b in range(8)]
    live = list(range(8))
    while live:                                                  # round-robin: everybody moves at the same speed
        for i in list(live):
            try:
                next(gens[i])
            except StopIteration:
                live.remove(i)
    assert S.encoded == [1] * 64 and S.passes == 128


def test_ticket_plan_covers_every_frame_once_and_keeps_the_rounds_whole():
    """the runs partition [0, n), long runs first, and no round of the grid is split between run lengths (what keeps runs from
    costing balance); the C function the library uses agrees with this model"""
    import ctypes as C
    import os
    lib = None
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "psxavenc_amd", "libpsxav_hip.so")
    if os.path.exists(so):
        try:
            lib = C.CDLL(so)
        except OSError:
            lib = None
    for groups in (1, 2, 7, 256, 512):
        for n in list(range(0, 40)) + [groups - 1, groups, groups + 1, 2 * groups - 1, 2 * groups, 2 * groups + 1, 1000, 1250, 4000, 10000, 65535]:
            if n < 0:
                continue
            for max_run in (1, 2, 4):
                plan = ticket_plan(n, groups, max_run)
                covered = []
                for t in range(plan[2]):
                    first, ln = ticket_run(plan, t)
                    covered.extend(range(first, first + ln))
                    assert ln <= max_run
                assert covered == list(range(n)), (n, groups, max_run, plan)
                t4, t2, nt = plan
                assert t4 % groups == 0 and (max_run < 2 or nt - t4 - t2 <= groups)
                if lib is not None:
                    a, b, c = C.c_int(), C.c_int(), C.c_int()
                    lib.psxhip_mdec_ticket_plan(n, groups, max_run, C.byref(a), C.byref(b), C.byref(c))
                    assert (a.value, b.value, c.value) == plan, (n, groups, max_run)
    assert ticket_plan(1000, 512) == (0, 500, 500) and ticket_plan(1250, 512) == (0, 512, 738) and ticket_plan(4000, 512) == (512, 976, 1488)
