"""CPU-side checks of the MDEC restatement (oracle/mdec_oracle.c): FDCT sanity pins, the closed
forms the HIP kernel relies on (integer rounding division, fp32 reciprocal quantiser, first-fit rate control),
self-golden regression vectors and encode -> decode round trips."""
import hashlib
import os

import numpy as np
import pytest
import scipy.fft

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mdec_selfgolden.npz")


def fdct(block):
    b = np.ascontiguousarray(block, dtype=np.int16).copy()
    O.lib().orc_fdct_islow8(O.ptr(b, O.i16p))
    return b


def test_fdct_constant_blocks():
    for c in (-128, -1, 0, 1, 5, 127):
        out = fdct(np.full(64, c))
        assert out[0] == 64 * c and not out[1:].any()


def test_fdct_close_to_float_dct():
    rng = np.random.default_rng(5)
    blocks = [rng.integers(-128, 128, 64) for _ in range(500)]
    yy, xx = np.mgrid[0:8, 0:8]
    blocks += [((xx + yy) * 16 - 112).ravel(), (((xx + yy) & 1) * 255 - 128).ravel(), (((xx // 4) & 1) * 255 - 128).ravel()]
    worst = 0.0
    for b in blocks:
        want = 8.0 * scipy.fft.dctn(np.asarray(b, float).reshape(8, 8), norm="ortho")
        worst = max(worst, np.abs(fdct(b).reshape(8, 8) - want).max())
    assert worst < 1.0


def test_dc_range_matches_clamp_range():
    # DC = round(64*c/16): c=127 -> 508, c=-128 -> -512 (the clamp range at mdec.c:262-265)
    assert fdct(np.full(64, 127))[0] == 8128 and round(8128 / 16) == 508
    assert fdct(np.full(64, -128))[0] == -8192 and round(-8192 / 16) == -512


def test_integer_form_of_divide_rounded():
    """mdec.c:438 (round-half-away of n/d in double) == sgn(n) * ((2|n| + d) // (2d)) for every divisor in use."""
    import sys
    sys.path.insert(0, os.path.join(O.ROOT, "tools"))
    import gen_tables as G
    divisors = sorted({q * s for q in G.QUANT[1:] for s in range(1, 64)} | {16, 4})
    n = np.arange(-40000, 40001, dtype=np.int64)
    for d in divisors[::7] + [16, 4]:
        x = n / float(d)
        ref = np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)   # round half away from zero
        mine = np.sign(n) * ((2 * np.abs(n) + d) // (2 * d))
        assert np.array_equal(ref, mine), d


def test_fp32_reciprocal_quantiser_is_exact():
    """The kernel computes floor(N / D) as trunc(fmaf(float(2|n|), r, 0.5f + 0.5f * r)), r = fp32(1 / D), N = 2|n| + d, D = 2d
    (ac_eval() in mdec_kernels.hip; quant_level() documents the equivalent two-rounding form; all three forms are checked).  Exhaustive over every
    divisor and every N the path can produce, with the reciprocal perturbed by +-2 ulp (v_rcp_f32 is 1 ulp)."""
    import sys
    sys.path.insert(0, os.path.join(O.ROOT, "tools"))
    import gen_tables as G
    divisors = sorted({q * s for q in G.QUANT[1:] for s in range(1, 65)})
    a2 = 2 * np.arange(0, 32769, dtype=np.int64)
    for d in divisors:
        N = a2 + d
        D = 2 * d
        want = N // D
        r = np.float32(1.0) / np.float32(D)
        for ulps in (-2, 0, 2):
            rr = r
            for _ in range(abs(ulps)):
                rr = np.nextafter(rr, np.float32(np.inf if ulps > 0 else -np.inf), dtype=np.float32)
            got = ((N.astype(np.float32) + np.float32(0.5)) * rr).astype(np.int64)
            assert np.array_equal(got, want), (d, ulps)
            # fused form: exact product and sum in float64 (N < 2^18, r has 24 significant bits), one rounding to fp32
            hr = np.float32(0.5) * rr
            fused = (N.astype(np.float64) * np.float64(rr) + np.float64(hr)).astype(np.float32)
            assert np.array_equal(fused.astype(np.int64), want), (d, ulps, "fma")
            # the kernel's final form folds d / 2d = 0.5 into the addend: trunc(fmaf(float(2|n|), r, 0.5f + 0.5f * r))
            bias = np.float32(0.5) + np.float32(0.5) * rr
            folded = (a2.astype(np.float64) * np.float64(rr) + np.float64(bias)).astype(np.float32)
            assert np.array_equal(folded.astype(np.int64), want), (d, ulps, "folded")


def _levels_numpy(coefs, scale):
    """Quantised levels [6, nmb, 64] (raster order) by the integer closed form."""
    import sys
    sys.path.insert(0, os.path.join(O.ROOT, "tools"))
    import gen_tables as G
    q = np.array(G.QUANT, np.int64) * scale
    q[0] = 16
    n = coefs.astype(np.int64)
    lv = np.sign(n) * ((2 * np.abs(n) + q) // (2 * q))
    return np.clip(lv, -512, 510)


@pytest.mark.parametrize("codec,w,h,budget,amp", [(0, 320, 240, 8192, 4), (1, 320, 240, 8192, 8), (2, 48, 32, 4096, 8),
                                                   (1, 640, 480, 32768, 8), (0, 16, 16, 4096, 0)])
def test_roundtrip_decode_recovers_levels(codec, w, h, budget, amp):
    import sys
    sys.path.insert(0, os.path.join(O.ROOT, "tools"))
    import gen_tables as G
    fr = O.synth_frames(w, h, 2, seed=9, amp=amp)
    out, res, rc = O.mdec_encode(codec, w, h, fr, budget)
    assert rc == 0
    nx, ny = w // 16, h // 16
    zz = np.array(G.zagzig())
    for k in range(2):
        rc2, levels, scale, version, nbits = O.mdec_decode(w, h, out[k], v3dc_wrap=int(codec == 2))
        assert rc2 == 0 and scale == res[k, 0] and version == (2 if codec == 0 else 3)
        assert res[k, 1] == ((8 + 2 * ((nbits + 15) // 16) + 3) & ~3)
        want = _levels_numpy(O.mdec_coefs(w, h, fr[k]), scale)           # [6, nmb, 64] raster, mb index fy*nx+fx
        # decoder order: macroblocks fx-major, 6 blocks each, zig-zag inside
        got = levels.reshape(nx, ny, 6, 64)
        for fx in range(nx):
            for fy in range(ny):
                w_ = want[:, fy * nx + fx, :][:, zz]
                assert np.array_equal(got[fx, fy][:, 1:], w_[:, 1:]), (fx, fy)
                if codec == 0:
                    assert np.array_equal(got[fx, fy][:, 0], w_[:, 0])
                else:   # v3 carries DC as a DPCM of multiples of 4 (mdec.c:460-461): decoder sees dc rounded to 4
                    assert not (got[fx, fy][:, 0] & 3).any()
                    assert np.abs(got[fx, fy][:, 0].astype(int) - w_[:, 0]).max() <= 2
        # everything after the end-of-frame code is zero (the reference's memset, mdec.c:676)
        assert not out[k, res[k, 1]:].any()
        rec = O.mdec_reconstruct(w, h, levels, scale)
        mse = ((rec.astype(float) - fr[k]) ** 2).mean()
        assert 10 * np.log10(255.0 ** 2 / max(mse, 1e-9)) > 30.0


def test_rate_control_is_first_fit():
    """chosen scale = min s with 8 + 2*ceil(bits(s)/16) <= budget; one step tighter budget moves to a later scale."""
    w, h = 320, 240
    fr = O.synth_frames(w, h, 1, seed=3, amp=8)
    out, res, rc = O.mdec_encode(0, w, h, fr, 60000)
    assert rc == 0 and res[0, 0] == 1
    _, _, _, _, nbits = O.mdec_decode(w, h, out[0])
    need = 8 + 2 * ((nbits + 15) // 16)
    for budget, expect_scale1 in ((need, True), (need + 1, True), (need - 1, False), (need - 2, False)):
        _, r2, rc2 = O.mdec_encode(0, w, h, fr, budget)
        assert rc2 == 0 and (r2[0, 0] == 1) == expect_scale1, (budget, r2)


def test_no_fit_is_reported():
    fr = O.synth_frames(320, 240, 1, seed=3, amp=8)
    _, _, rc = O.mdec_encode(0, 320, 240, fr, 2700)       # floor: 1800*12+10 bits = 2710 bytes of payload
    assert rc == -2


def test_selfgolden_vectors():
    g = np.load(GOLD)
    t = g["table"]
    for row in t[::3]:                                    # every third case keeps the CPU suite quick; the GPU suite runs all
        codec, w, h, budget, amp, n, rc = (int(v) for v in row[:7])
        fr = O.synth_frames(w, h, n, seed=100 + amp, amp=amp, first=3)
        out, res, rc2 = O.mdec_encode(codec, w, h, fr, budget)
        assert rc2 == rc
        if rc == 0:
            assert res.ravel().tolist() == row[7:7 + 4 * n].tolist()
            sha = hashlib.sha256(out.tobytes()).digest()
            assert sha == g["sha_c%d_%dx%d_b%d_a%d" % (codec, w, h, budget, amp)].tobytes()


def test_dct_linear_forms():
    """the HIP kernel evaluates the islow butterfly as integer LINEAR FORMS (two packed int16 dot products per output,
    csrc/mdec_kernels.hip fdct8_pk) and feeds the row pass RAW pixels, correcting only the DC term for the level shift.
    Same arithmetic in numpy (exact integers), against the oracle's butterfly, incl. extreme blocks; also checks that
    every packed operand fits int16 and every coefficient of the forms fits int16."""
    K = dict(k298=2446, k390=3196, k541=4433, k765=6270, k899=7373, k1175=9633, k1501=12299, k1847=15137, k1961=16069,
             k2053=16819, k2562=20995, k3072=25172)
    A, B, Cc = K["k541"] + K["k765"], K["k541"], K["k541"] - K["k1847"]
    odd = {
        7: (K["k298"] - K["k899"] - K["k1961"] + K["k1175"], K["k1175"], K["k1175"] - K["k1961"], K["k1175"] - K["k899"]),
        5: (K["k1175"], K["k2053"] - K["k2562"] - K["k390"] + K["k1175"], K["k1175"] - K["k2562"], K["k1175"] - K["k390"]),
        3: (K["k1175"] - K["k1961"], K["k1175"] - K["k2562"], K["k3072"] - K["k2562"] - K["k1961"] + K["k1175"], K["k1175"]),
        1: (K["k1175"] - K["k899"], K["k1175"] - K["k390"], K["k1175"], K["k1501"] - K["k899"] - K["k390"] + K["k1175"]),
    }
    assert all(-32768 <= c <= 32767 for v in odd.values() for c in v) and A <= 32767 and Cc >= -32768

    def pass1d(d, column):
        d = d.astype(np.int64)
        s07, s16, s25, s34 = d[..., 0] + d[..., 7], d[..., 1] + d[..., 6], d[..., 2] + d[..., 5], d[..., 3] + d[..., 4]
        o0, o1, o2, o3 = d[..., 3] - d[..., 4], d[..., 2] - d[..., 5], d[..., 1] - d[..., 6], d[..., 0] - d[..., 7]
        for v in (s07, s16, s25, s34, o0, o1, o2, o3):
            assert v.min() >= -32768 and v.max() <= 32767          # packed int16 operands cannot wrap
        sh = 17 if column else 9
        rnd = 1 << (sh - 1)
        out = np.zeros(d.shape, np.int64)
        if column:
            out[..., 0] = (s07 + s16 + s25 + s34 + 8) >> 4
            out[..., 4] = (s07 - s16 - s25 + s34 + 8) >> 4
        else:
            out[..., 0] = (s07 + s16 + s25 + s34 - 8 * 128) * 16      # raw pixels: the level shift only moves this term
            out[..., 4] = (s07 - s16 - s25 + s34) * 16
        out[..., 2] = (s07 * A + s16 * B - s25 * B - s34 * A + rnd) >> sh
        out[..., 6] = (s07 * B + s16 * Cc - s25 * Cc - s34 * B + rnd) >> sh
        for k, (c0, c1, c2, c3) in odd.items():
            out[..., k] = (o0 * c0 + o1 * c1 + o2 * c2 + o3 * c3 + rnd) >> sh
        return out.astype(np.int16).astype(np.int64)                 # stored as int16 between and after the passes

    rng = np.random.default_rng(5)
    blocks = rng.integers(0, 256, (20000, 8, 8))
    blocks[0], blocks[1] = 0, 255
    blocks[2] = np.where((np.indices((8, 8)).sum(0) & 1) > 0, 255, 0)
    blocks[3:1000] = np.where(rng.integers(0, 2, (997, 8, 8)) > 0, 255, 0)
    rows = pass1d(blocks, False)                                    # row pass on raw pixels
    cols = pass1d(rows.transpose(0, 2, 1), True).transpose(0, 2, 1)  # column pass
    want = (blocks - 128).astype(np.int16).reshape(-1, 64).copy()
    for i in range(want.shape[0]):
        O.lib().orc_fdct_islow8(O.ptr(want[i], O.i16p))
    assert np.array_equal(cols.reshape(-1, 64), want.astype(np.int64))
    assert np.array_equal(cols[:, 0, 0], (blocks - 128).sum(axis=(1, 2)))   # DC == sum of the level-shifted samples (v3 pre-pass)


def test_integer_identities_of_the_kernel():
    """Arithmetic rewrites the frame kernel relies on (psxavenc_amd/csrc/mdec_kernels.hip), checked over their whole ranges:
    * quant_dc: sgn(c) * ((|c| + 8) >> 4) == (c + 8 + (c >> 31)) >> 4  (the DC quantiser DIVIDE_ROUNDED(c, 16), mdec.c:438,447);
    * column pass: out = (acc + 2^16) >> 17 taken as ((acc + 2^16) >> 16) >> 1 from the accumulator's HIGH half (pack_sh17);
    * column pass outputs 0 / 4: (sum + 8) >> 4 == (8192 * sum + 2^16) >> 17, and 8192 * sum stays inside int32."""
    c = np.arange(-140000, 140000, dtype=np.int64)
    a = np.abs(c)
    q = (a + 8) >> 4
    assert np.array_equal(np.where(c < 0, -q, q), (c + 8 + (c >> 63)) >> 4)
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.integers(-2**31, 2**31 - 2**16, 2_000_000, dtype=np.int64), np.arange(-70000, 70000, dtype=np.int64),
                        np.array([-2**31, 2**31 - 2**16 - 1], np.int64)])
    acc = x + 65536
    hi = (acc.astype(np.int32).view(np.int32) >> 16).astype(np.int16)          # the high half, as the byte permute takes it
    assert np.array_equal((acc >> 17).astype(np.int64), (hi.astype(np.int64) >> 1))
    s = np.arange(-4 * 32640, 4 * 32640 + 1, dtype=np.int64)                    # |sum| <= 4 * (32768 - 128)
    assert np.abs(8192 * s + 65536).max() < 2**31
    assert np.array_equal((s + 8) >> 4, (8192 * s + 65536) >> 17)
