"""A model of the frame kernel's hand-on protocol (mdec_kernels.hip: draw_ticket / hand_on / end_of_frame / the pop phase at the
top of the frame loop), run under random interleavings.

The GPU tests check bytes; they cannot choose the schedule.  Here every workgroup is a little state machine that performs ONE
shared-memory operation per step -- the same operations on the same words as the kernel (one 64-bit state word: fresh tickets |
slots reserved | pop tickets; the slot array; the exit counter with its abandonment notes) -- and a seeded scheduler picks who
moves next, including workgroups that start late (not resident: the reason the wait is bounded) and waiting workgroups that
run out of patience.  Properties: every frame is encoded to the end exactly once, every workgroup leaves, nobody waits for a
slot that is never filled, and the last one out leaves counters and slots as the next launch expects them."""
import random

import pytest

EMPTY, ABANDONED = 0xFFFFFFFF, 0xFFFFFFFE


class Shared:
    def __init__(self, n_frames, grid, cap):
        self.tickets = self.reserved = self.head = 0          # the state word's three fields (one atomic word in the kernel)
        self.slots = [EMPTY] * cap
        self.left = 0                                         # groups gone | abandonments << 16
        self.started = 0                                      # groups that have started
        self.n, self.grid = n_frames, grid
        self.encoded = [0] * n_frames                         # times a frame was encoded to the end
        self.passes = 0


def group(S, b, rng, p_wrong, patience, start_delay):
    """generator: yields once per shared-memory operation (and once per unit of local work)"""
    for _ in range(start_delay):
        yield "not started"
    frame, retried = b, False
    next_draw = S.tickets; S.tickets += 1; yield "draw"                       # atomicAdd(state, 1)
    S.started += 1; yield "started"
    fresh_draws = S.n - S.grid
    while True:
        # ---- one frame: passes until the search is done; a wrong first guess on a fresh frame may be handed on
        deferred = False
        wrong = (not retried) and rng.random() < p_wrong
        S.passes += 1; yield "pass"
        if wrong:
            if not retried and next_draw < fresh_draws:                       # hand_on()
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
        frame, retried = next_draw + S.grid, False
        queue = None
        if next_draw < fresh_draws:
            next_draw = S.tickets; S.tickets += 1; yield "draw"
        else:
            h, reserved, tickets = S.head, S.reserved, S.tickets; S.head += 1; yield "pop ticket"      # atomicAdd(state, 1 << 48)
            queue = h if reserved > h else (-1 if tickets >= S.n else -2 - h)
        if frame < S.n:
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
            if tickets >= S.n:
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


@pytest.mark.parametrize("seed", range(40))
def test_every_frame_once_everybody_leaves_counters_rearmed(seed):
    rng = random.Random(seed)
    grid = rng.choice([2, 3, 8, 16])
    n = rng.randint(grid + 1, 8 * grid)
    patience = rng.choice([0, 1, 3, 50])
    p_wrong = rng.choice([0.0, 0.2, 0.6, 1.0])
    late = rng.random() < 0.5                                   # some groups start long after the others (not resident)
    S = Shared(n, grid, cap=n + grid + 4)
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
