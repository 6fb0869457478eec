"""ADPCM restatement (oracle/adpcm_oracle.c) pinned two ways: against the reference's own libpsxav compiled
unchanged (oracle/_ref, when present) and against tests/golden/adpcm_ref.npz, which was generated from that
library by tests/golden/make_adpcm_golden.py.  Also the semantics the GPU search relies on."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adpcm_ref.npz")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()


def stereo_pad(kind, n, seed, pad=4032):
    pcm = np.zeros((n + pad) * 2, np.int16)
    pcm[0:2 * n:2] = O.synth_pcm(seed, 0, 0, n, kind)
    pcm[1:2 * n:2] = O.synth_pcm(seed, 1, 0, n, kind)
    return pcm


def mono_pad(kind, n, seed, pad=4032):
    pcm = np.zeros(n + pad, np.int16)
    pcm[:n] = O.synth_pcm(seed, 0, 0, n, kind)
    return pcm


def test_known_answers_from_survey():
    i = np.arange(22050)
    sine = np.rint(16384 * np.sin(2 * np.pi * 440 * i / 22050)).astype(np.int16)
    buf = np.zeros(20000, np.uint8)
    ln = O.lib().orc_spu_encode_simple(O.ptr(sine, O.i16p), sine.size, O.ptr(buf, O.u8p), -1)
    assert ln == 12624 and buf[:4].tobytes() == bytes([0x24, 0x00, 0x70, 0x13])
    assert buf[ln - 16:ln].tobytes() == bytes([0, 5] + [0] * 14)           # trailing LOOP_TRAP block
    out, _ = O.spu_encode(np.zeros(56, np.int16))
    assert out[0] == 0x0B and out[16] == 0x0B                               # silent block: filter 0, shift 11


def test_spu_golden_from_reference():
    g = np.load(GOLD)
    for kind in range(6):
        for n in (28, 29, 280, 28 * 100 + 13):
            pcm = O.synth_pcm(11, kind, 0, n, kind)
            data, st = O.spu_encode(pcm)
            key = "spu_k%d_n%d" % (kind, n)
            assert np.array_equal(data, g[key]), key
            assert [st.prev1, st.prev2] == g[key + "_state"].tolist()
    pcm = O.synth_pcm(11, 0, 0, 28 * 20000, 0)
    data, st = O.spu_encode(pcm)
    assert sha(data) == g["spu_long_sha"].tobytes()
    assert [st.prev1, st.prev2] == g["spu_long_state"].tolist()
    pcm2 = stereo_pad(0, 28 * 50, 5, pad=0)
    data, _ = O.spu_encode(pcm2, pitch=2, n=28 * 50)
    assert np.array_equal(data, g["spu_pitch2"])


def test_spu_simple_golden_from_reference():
    g = np.load(GOLD)
    i = np.arange(22050)
    sine = np.rint(16384 * np.sin(2 * np.pi * 440 * i / 22050)).astype(np.int16)
    for loop in (-1, 280):
        buf = np.zeros(20000, np.uint8)
        ln = O.lib().orc_spu_encode_simple(O.ptr(sine, O.i16p), sine.size, O.ptr(buf, O.u8p), loop)
        assert ln == int(g["spu_simple_sine_loop%d_len" % loop][0])
        assert sha(buf[:ln]) == g["spu_simple_sine_loop%d_sha" % loop].tobytes()


def test_xa_golden_from_reference():
    g = np.load(GOLD)
    for fmt in (0, 1):
        for stereo in (0, 1):
            for bits in (4, 8):
                for freq in (37800, 18900):
                    for kind, n in ((0, 5000), (5, 300), (2, 2016), (3, 100), (1, 4033)):
                        s = O.XaSettings(fmt, stereo, freq, bits, 3, 7)
                        pcm = stereo_pad(kind, n, 21) if stereo else mono_pad(kind, n, 21)
                        data, st = O.xa_encode(s, pcm, n, lba=1234)
                        key = "xa_f%d_s%d_b%d_q%d_k%d_n%d" % (fmt, stereo, bits, freq, kind, n)
                        assert data.size == int(g[key + "_len"][0]), key
                        assert sha(data) == g[key + "_sha"].tobytes(), key
                        assert [st.left.prev1, st.left.prev2, st.right.prev1, st.right.prev2] == g[key + "_state"].tolist()
    s = O.XaSettings(1, 1, 37800, 4, 1, 0)
    data, _ = O.xa_encode(s, stereo_pad(0, 2016, 9), 2016, lba=0)
    assert np.array_equal(data, g["xa_full_sector"])


def test_against_live_reference_build():
    if O.ref() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(1)
    for trial in range(40):
        kind = int(rng.integers(0, 6))
        n = int(rng.integers(1, 3000))
        pcm = O.synth_pcm(int(rng.integers(1, 1 << 30)), trial, int(rng.integers(0, 1 << 20)), n, kind)
        a, sa = O.spu_encode(pcm)
        b, sb = O.ref_spu_encode(pcm)
        assert np.array_equal(a, b) and (sa.prev1, sa.prev2) == (sb.prev1, sb.prev2)
    for trial in range(24):
        s = O.XaSettings(trial & 1, (trial >> 1) & 1, 37800 if trial & 4 else 18900, 8 if trial & 8 else 4, trial, trial * 3)
        n = int(rng.integers(1, 9000))
        kind = int(rng.integers(0, 6))
        pcm = stereo_pad(kind, n, trial) if s.stereo else mono_pad(kind, n, trial)
        a, sa = O.xa_encode(s, pcm, n, lba=trial * 1000)
        b, sb = O.ref_xa_encode(s, pcm, n, lba=trial * 1000)
        assert np.array_equal(a, b), trial
    # size helpers
    R = O.ref()
    for trial in range(16):
        s = O.XaSettings(trial & 1, (trial >> 1) & 1, 37800 if trial & 4 else 18900, 8 if trial & 8 else 4, 0, 0)
        rs = O.ref_settings(s)
        assert O.lib().orc_xa_samples_per_sector(s) == R.psx_audio_xa_get_samples_per_sector(rs)
        assert O.lib().orc_xa_sector_size(s) == R.psx_audio_xa_get_buffer_size_per_sector(rs)
        assert O.lib().orc_xa_sector_interleave(s) == R.psx_audio_xa_get_sector_interleave(rs)


def test_state_carry_across_calls():
    """28-sample calls with carried state == one big call (SURVEY 8(b) call pattern, filefmt.c:243)."""
    pcm = O.synth_pcm(5, 2, 0, 28 * 40, 0)
    whole, st = O.spu_encode(pcm)
    st2 = O.Chan(0, 0)
    parts = []
    for k in range(40):
        o, st2 = O.spu_encode(pcm[28 * k:28 * k + 28], state=st2)
        parts.append(o)
    assert np.array_equal(np.concatenate(parts), whole)
    assert (st.prev1, st.prev2) == (st2.prev1, st2.prev2)


def test_search_window_is_not_the_full_grid():
    """the argmin runs over <= 3 shifts around a per-filter minimum, not all 5 x 13.
    A silent block must come out as filter 0, shift 11 (header 0x0B); an unrestricted search would say 0x00 or 0x0C."""
    out, _ = O.spu_encode(np.zeros(28, np.int16))
    assert out[0] == 0x0B


def test_edc_crc_properties():
    L = O.lib()
    z = np.zeros(2048, np.uint8)
    assert L.orc_edc_crc32(O.ptr(z, O.u8p), 2048) == 0
    one = np.zeros(1, np.uint8)
    one[0] = 1
    # table-less definition: reflected polynomial 0xD8018001 applied 8 times to 0x01
    v = 1
    for _ in range(8):
        v = (v >> 1) ^ (0xD8018001 if v & 1 else 0)
    assert L.orc_edc_crc32(O.ptr(one, O.u8p), 1) == v


def test_min_shift_closed_form_equals_the_reference_loops():
    """adpcm.c:72-73 finds the right shift with two while loops; the HIP kernel (adpcm_kernels.hip, encode_unit) uses
    clamp(bit_length(max(s_max, ~s_min)) - (15 - range), 0, range).  Same function for every reachable operand."""
    def loops(hi, lo, rng):
        rs = 0
        while rs < rng and (hi >> rs) > (0x7FFF >> rng):
            rs += 1
        while rs < rng and (lo >> rs) < (-0x8000 >> rng):
            rs += 1
        return rs

    def closed(hi, lo, rng):
        m = max(hi, ~lo)
        return min(max((m.bit_length() if m > 0 else 0) - (15 - rng), 0), rng)

    rnd = np.random.default_rng(3)
    edges = sorted({0, 1} | {(1 << k) + d for k in range(1, 19) for d in (-1, 0, 1)})
    for rng in (8, 12):
        for hi in edges:
            for lo in edges:
                assert loops(hi, -lo, rng) == closed(hi, -lo, rng), (hi, -lo, rng)
        for hi, lo in zip(rnd.integers(0, 1 << 18, 20000).tolist(), rnd.integers(0, 1 << 18, 20000).tolist()):
            assert loops(hi, -lo, rng) == closed(hi, -lo, rng), (hi, -lo, rng)
