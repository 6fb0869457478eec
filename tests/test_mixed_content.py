"""The scene-structured workload of the bench line (psxavenc_amd/mixed.py): a pure function of (seed, frame index) whose
every frame fits the headline's 8192-byte budget (the reference asserts otherwise, mdec.c:723) and whose answers span many quant scales --
checked with the CPU oracle; no GPU needed."""
import collections

import numpy as np

import oracle_lib as O
from psxavenc_amd import mixed


def test_plan_is_a_pure_function_of_seed_and_covers_the_frames():
    a, runs = mixed.plan(7, 3000)
    b, _ = mixed.plan(7, 3000)
    c, _ = mixed.plan(8, 3000)
    assert a == b and a != c and len(a) == 3000
    # a prefix of a longer plan is the shorter plan: any rank can build any frame range
    d, _ = mixed.plan(7, 1200)
    assert d == a[:1200]
    assert sum(cnt for _, cnt, _, _ in runs) == 3000 and all(5 <= cnt <= 30 for _, cnt, _, _ in runs[:-1])
    amps = {amp for _, _, amp, _ in runs}
    assert min(amps) >= 2 and max(amps) <= 40 and len(amps) >= 20
    kinds = collections.Counter(k for k, _, _ in a)
    assert 0.03 < kinds["special"] / 3000.0 < 0.08


def test_every_frame_fits_the_headline_budget_and_the_answers_span_many_scales():
    w, h = 320, 240
    frames = mixed.frames_host(O, w, h, 1, 0, 300)
    out, res, rc = O.mdec_encode(0, w, h, frames, 8192)
    assert rc == 0 and (res[:, 0] < 64).all()
    assert len(set(res[:, 0].tolist())) >= 8, sorted(set(res[:, 0].tolist()))
    # frames of one scene sit next to each other on the scale axis, cuts do not: most neighbours agree, some differ by a lot
    d = np.abs(np.diff(res[:, 0]))
    assert (d == 0).mean() > 0.6 and (d >= 3).sum() >= 10
    # every hand-made frame on its own, all three codecs
    for k in range(mixed.N_SPECIAL):
        f = mixed.special_frame(k, w, h)[None]
        for codec in (0, 1, 2):
            _, r, rc = O.mdec_encode(codec, w, h, f, 8192)
            assert rc == 0 and 1 <= r[0, 0] <= 63, (k, codec, r[0])
    # a slice built on its own equals the same slice of the whole
    assert np.array_equal(mixed.frames_host(O, w, h, 1, 120, 40), frames[120:160])
