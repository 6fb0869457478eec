"""CPU test of the MDEC rate-control search policy (psxavenc_amd/csrc/mdec_search.h): whatever the bits(scale)
curve looks like -- monotone, bumpy, with or without the lower-bound proof being available -- the search must
return the FIRST scale that fits (the reference's ascending loop, psxavenc/mdec.c:663-723) and terminate."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("search") / "libsearch_sim.so")
    subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests/cpu/search_sim.cpp")],
                   check=True)
    L = C.CDLL(so)
    ip = C.POINTER(C.c_int)
    L.search_sim.argtypes = [ip, ip, C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, ip]
    L.pilot_sim.argtypes = [ip, C.c_int, C.c_int, C.c_int, ip, ip]
    return L


def run(L, tb, fb, limit, fixed, guess, overflow=None):
    tb = np.ascontiguousarray(tb, dtype=np.int32)
    fb = np.ascontiguousarray(fb, dtype=np.int32)
    n, lo, hi = C.c_int(), C.c_int(), C.c_int()
    ip = C.POINTER(C.c_int)
    r = L.search_sim(tb.ctypes.data_as(ip), fb.ctypes.data_as(ip), limit, fixed, guess,
                     limit + 2000 if overflow is None else overflow, C.byref(n), C.byref(lo), C.byref(hi))
    return r, n.value


def first_fit(tb, limit):
    for s in range(1, 64):
        if tb[s] <= limit:
            return s
    return 64


def curve(rng, fixed, kind):
    """total bits for scales 1..63 (index 0 unused) and a VALID lower bound: fb[s'] <= min over s <= s' of tb[s]."""
    base = rng.integers(20000, 400000)
    gamma = rng.uniform(0.6, 1.4)
    s = np.arange(64, dtype=np.float64)
    s[0] = 1
    ac = base / s ** gamma
    if kind == "bumpy":
        ac *= 1 + rng.uniform(-0.08, 0.08, 64)        # non-monotone
    elif kind == "flat":
        ac = np.maximum(ac, rng.integers(1000, 30000))
    tb = (ac + fixed).astype(np.int64)
    runmin = np.minimum.accumulate(tb[1:])              # min over s <= s'
    fb = np.empty(64, np.int64)
    fb[0] = 0
    slack = rng.choice([0, 0, 0, rng.integers(0, 3000)], 63)    # mostly tight (no escapes), sometimes loose
    fb[1:] = runmin - slack
    if kind == "noproof":
        fb[1:] = fixed                                       # the bound never proves anything
    tb[0] = 0
    return tb, fb


@pytest.mark.parametrize("kind", ["smooth", "bumpy", "flat", "noproof"])
def test_search_returns_first_fit(sim, kind):
    rng = np.random.default_rng({"smooth": 1, "bumpy": 2, "flat": 3, "noproof": 4}[kind])
    passes = []
    for _ in range(3000):
        fixed = int(rng.integers(3000, 30000))
        tb, fb = curve(rng, fixed, kind)
        limit = int(rng.integers(fixed - 2000, 140000))
        want = first_fit(tb, limit)
        for guess in (want if want < 64 else 63, max(1, want - 1), min(63, want + 1), int(rng.integers(1, 64))):
            got, n = run(sim, tb, fb, limit, fixed, guess)
            assert got == want, (kind, limit, fixed, guess, got, want, n)
            passes.append(n)
    passes = np.array(passes)
    # a good guess on a well-behaved curve needs one pass; nothing needs more than a scan of all scales
    assert passes.max() <= 64
    if kind == "smooth":
        assert np.mean(passes[0::4]) < 1.2


@pytest.mark.parametrize("kind", ["smooth", "bumpy", "flat"])
def test_pilot_policy_finds_the_first_fit_of_its_estimate_in_few_rounds(sim, kind):
    """mdec_pilot_next (round 5: the pilot steered by the search's two-point model instead of bracketing by halves): on a monotone
    estimate it returns exactly the first scale that fits, whatever the hint; on a bumpy one a scale next to it; two to three rounds
    and about five evaluations on average, never more than six rounds"""
    rng = np.random.default_rng({"smooth": 11, "bumpy": 12, "flat": 13}[kind])
    rounds, evals, exact, near, total = [], [], 0, 0, 0
    for _ in range(3000):
        fixed = int(rng.integers(3000, 30000))
        tb, _ = curve(rng, fixed, kind)
        limit = int(rng.integers(fixed + 500, 140000))
        want = min(first_fit(tb, limit), 63)
        est = np.ascontiguousarray(tb, dtype=np.int32)
        for h0 in (0, want, max(1, want - 1), min(63, want + 3), int(rng.integers(1, 64))):
            r, e = C.c_int(), C.c_int()
            got = sim.pilot_sim(est.ctypes.data_as(C.POINTER(C.c_int)), limit, fixed, h0, C.byref(r), C.byref(e))
            assert 1 <= got <= 63, (kind, limit, fixed, h0, got)
            assert r.value <= 6
            total += 1
            exact += got == want
            near += abs(got - want) <= 1
            rounds.append(r.value)
            evals.append(e.value)
            if kind == "smooth":
                assert got == want or r.value == 6, (limit, fixed, h0, got, want, r.value)
    assert np.mean(rounds) < 3.2 and np.mean(evals) < 8.0, (np.mean(rounds), np.mean(evals))
    assert exact >= 0.93 * total and near >= (0.99 if kind == "smooth" else 0.95) * total, (kind, exact, near, total)
    print(kind, "rounds", np.mean(rounds), "evaluations", np.mean(evals), "exact", exact / total)


def test_overflowing_emit_is_not_trusted(sim):
    rng = np.random.default_rng(7)
    for _ in range(500):
        fixed = 10000
        tb, fb = curve(rng, fixed, "smooth")
        limit = int(rng.integers(20000, 100000))
        want = first_fit(tb, limit)
        got, n = run(sim, tb, fb, limit, fixed, int(rng.integers(1, 64)), overflow=limit)   # every non-fitting emit overflows
        assert got == want


def test_nothing_fits(sim):
    tb = np.full(64, 90000, np.int32)
    fb = np.full(64, 80000, np.int32)
    got, n = run(sim, tb, fb, 50000, 10000, 5)
    assert got == 64 and n <= 8
    got, n = run(sim, tb, np.full(64, 10000, np.int32), 50000, 10000, 5)     # no proof available: has to look at every scale
    assert got == 64
    got, n = run(sim, tb, fb, 5000, 10000, 5)                                   # budget below the fixed cost
    assert got == 64 and n == 0


@pytest.mark.parametrize("w,h,large", [(320, 240, 0), (320, 240, 1), (640, 480, 1), (160, 112, 0), (16, 16, 0), (1024, 1024, 1), (336, 240, 0)])
def test_pass_order_visits_every_macroblock_once_and_spreads_the_first_quarter(w, h, large):
    """The order in which a pass's tickets visit the macroblocks (psxhip_mdec_pass_order, host code of the library: no GPU
    needed): a permutation of the frame's macroblocks plus padding, whose first quarter -- the sample the checkpoint
    projects the frame's bits from -- touches every horizontal band of the frame about equally."""
    from psxavenc_amd import _lib
    L = _lib.lib()
    L.psxhip_mdec_pass_order.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_int]
    L.psxhip_mdec_pass_order.restype = C.c_int
    n = L.psxhip_mdec_pass_order(w, h, large, None, 0)
    waves = 16 if large else 12
    nx, ny = w // 16, h // 16
    nmb = nx * ny
    assert n == -(-nmb // waves) * waves
    buf = (C.c_uint32 * n)()
    assert L.psxhip_mdec_pass_order(w, h, large, buf, n) == n
    o = np.frombuffer(buf, dtype=np.uint32)
    valid = o != 0xFFFF
    assert valid.sum() == nmb
    fx, fy = o[valid] & 0xFF, o[valid] >> 8
    assert fx.max() < nx and fy.max() < ny
    assert len(set((fy * nx + fx).tolist())) == nmb
    if n // waves >= 8:
        q = o[:(n // waves // 4) * waves]
        q = q[q != 0xFFFF]
        bands = np.bincount(((q >> 8) * 4 // ny).astype(np.int64), minlength=4)     # quarters of the frame's height
        rounds = n // waves // 4
        if rounds >= 8:
            assert bands.min() * 3 >= bands.max(), bands
        else:
            assert (bands > 0).sum() >= min(rounds, 3), bands      # a handful of rounds cannot be even, only scattered
        # ... and the rounds behind the quarter mark follow in raster order (neighbouring rounds share the 128-byte fetch
        # granules at their ends: scattered, those were fetched twice)
        rest = o[(n // waves // 4) * waves:]
        first_of_round = rest[::waves]
        first_of_round = first_of_round[first_of_round != 0xFFFF].astype(np.int64)
        raster = (first_of_round >> 8) * nx + (first_of_round & 0xFF)
        assert (np.diff(raster) > 0).all()
