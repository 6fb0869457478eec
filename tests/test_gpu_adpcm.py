"""GPU parity tests of the SPU / XA ADPCM path: libpsxav_hip.so vs the reference-pinned CPU oracle (and, when
oracle/_ref travelled to the GPU box, the reference's own libpsxav directly).  Bar: bit-exact."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adpcm_ref.npz")


def stereo_pad(kind, n, seed, pad=4032):
    pcm = np.zeros((n + pad) * 2, np.int16)
    pcm[0:2 * n:2] = O.synth_pcm(seed, 0, 0, n, kind)
    pcm[1:2 * n:2] = O.synth_pcm(seed, 1, 0, n, kind)
    return pcm


def mono_pad(kind, n, seed, pad=4032):
    pcm = np.zeros(n + pad, np.int16)
    pcm[:n] = O.synth_pcm(seed, 0, 0, n, kind)
    return pcm


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()


def test_spu_golden_from_reference():
    from psxavenc_amd import adpcm
    g = np.load(GOLD)
    for kind in range(6):
        for n in (28, 29, 280, 28 * 100 + 13):
            pcm = O.synth_pcm(11, kind, 0, n, kind)
            st = np.zeros((1, 2), np.int32)
            out = adpcm.spu_encode_streams(pcm.reshape(1, -1), states=st)[0]
            key = "spu_k%d_n%d" % (kind, n)
            assert np.array_equal(out, g[key]), key
            assert st[0].tolist() == g[key + "_state"].tolist(), key
    pcm = O.synth_pcm(11, 0, 0, 28 * 20000, 0)
    st = np.zeros((1, 2), np.int32)
    out = adpcm.spu_encode_streams(pcm.reshape(1, -1), states=st)[0]
    assert sha(out) == g["spu_long_sha"].tobytes() and st[0].tolist() == g["spu_long_state"].tolist()
    pcm2 = stereo_pad(0, 28 * 50, 5, pad=0)
    out = adpcm.spu_encode_streams(pcm2.reshape(1, -1), pitch=2, sample_count=28 * 50)[0]
    assert np.array_equal(out, g["spu_pitch2"])


def test_spu_many_streams_ragged_vs_oracle():
    """independent streams of one batch (different content classes) == independent oracle runs; 4 chains share
    a wavefront, so stream counts that are not multiples of 4 and mixed classes exercise the masking"""
    from psxavenc_amd import adpcm
    for n_streams, n in ((1, 28), (3, 57), (7, 28 * 31 + 5), (64, 28 * 16)):
        pcm = np.stack([O.synth_pcm(99, c, 1000 * c, n, c % 6) for c in range(n_streams)])
        st = np.zeros((n_streams, 2), np.int32)
        out = adpcm.spu_encode_streams(pcm, states=st)
        for c in range(n_streams):
            want, wst = O.spu_encode(pcm[c])
            assert np.array_equal(out[c], want), (n_streams, n, c)
            assert st[c].tolist() == [wst.prev1, wst.prev2]


def test_spu_state_carry_28_sample_calls():
    """encode_file_spu's pattern (filefmt.c:243): 28 samples per call, state carried by the caller"""
    from psxavenc_amd import adpcm
    pcm = O.synth_pcm(5, 2, 0, 28 * 12, 0)
    want, wst = O.spu_encode(pcm)
    st = adpcm.ChannelState()
    parts = [adpcm.psx_audio_spu_encode(st, pcm[28 * k:28 * k + 28], 28) for k in range(12)]
    assert np.array_equal(np.concatenate(parts), want)
    assert (st.prev1, st.prev2) == (wst.prev1, wst.prev2)


def test_spu_simple_known_answer():
    from psxavenc_amd import adpcm
    g = np.load(GOLD)
    i = np.arange(22050)
    sine = np.rint(16384 * np.sin(2 * np.pi * 440 * i / 22050)).astype(np.int16)
    for loop in (-1, 280):
        out = adpcm.psx_audio_spu_encode_simple(sine, sine.size, loop)
        assert out.size == int(g["spu_simple_sine_loop%d_len" % loop][0])
        assert sha(out) == g["spu_simple_sine_loop%d_sha" % loop].tobytes()
    out = adpcm.psx_audio_spu_encode_simple(sine, sine.size, -1)
    assert out.size == 12624 and out[:4].tobytes() == bytes([0x24, 0x00, 0x70, 0x13])


def test_xa_golden_from_reference():
    from psxavenc_amd import adpcm
    g = np.load(GOLD)
    for fmt in (0, 1):
        for stereo in (0, 1):
            for bits in (4, 8):
                for freq in (37800, 18900):
                    for kind, n in ((0, 5000), (5, 300), (2, 2016), (3, 100), (1, 4033)):
                        s = adpcm.XaSettings(fmt, stereo, freq, bits, 3, 7)
                        pcm = stereo_pad(kind, n, 21) if stereo else mono_pad(kind, n, 21)
                        st = np.zeros((1, 2, 2), np.int32)
                        out = adpcm.xa_encode_streams(s, pcm.reshape(1, -1), n, lbas=[1234], states=st)[0]
                        key = "xa_f%d_s%d_b%d_q%d_k%d_n%d" % (fmt, stereo, bits, freq, kind, n)
                        assert out.size == int(g[key + "_len"][0]), key
                        assert sha(out) == g[key + "_sha"].tobytes(), key
                        assert st[0].ravel().tolist() == g[key + "_state"].tolist(), key
    s = adpcm.XaSettings(1, 1, 37800, 4, 1, 0)
    out = adpcm.xa_encode_streams(s, stereo_pad(0, 2016, 9).reshape(1, -1), 2016, lbas=[0])[0]
    assert np.array_equal(out, g["xa_full_sector"])


def test_xa_unpadded_input_equals_reference_with_padding():
    """the library never reads past sample_count (the reference needs >= 4032 zero samples of padding)"""
    from psxavenc_amd import adpcm
    s = adpcm.XaSettings(1, 1, 37800, 4, 0, 0)
    n = 777
    padded = stereo_pad(0, n, 3)
    want, _ = O.xa_encode(O.XaSettings(1, 1, 37800, 4, 0, 0), padded, n, lba=9)
    out = adpcm.xa_encode_streams(s, padded[:2 * n].reshape(1, -1), n, lbas=[9])[0]
    assert np.array_equal(out, want)


def test_xa_sector_by_sector_with_state_and_finalize():
    """encode_file_xa's pattern (filefmt.c:167-210): one sector per call, lba = sector index, EOF on the last"""
    from psxavenc_amd import adpcm
    s = adpcm.XaSettings(adpcm.PSX_AUDIO_XA_FORMAT_XACD, True, 37800, 4, 1, 2)
    os_ = O.XaSettings(1, 1, 37800, 4, 1, 2)
    sps = adpcm.xa_get_samples_per_sector(s)
    assert sps == 2016
    n = sps * 3 + 500
    pcm = stereo_pad(0, n, 17)
    st, ost = adpcm.EncoderState(), O.State()
    for k in range(4):
        cnt = min(sps, n - k * sps)
        chunk = pcm[2 * k * sps:]
        got = adpcm.psx_audio_xa_encode(s, st, chunk, cnt, k)
        want, ost = O.xa_encode(os_, chunk, cnt, lba=k, state=ost)
        if k == 3:
            got = adpcm.psx_audio_xa_encode_finalize(s, got)
            O.lib().orc_xa_encode_finalize(os_, O.ptr(want, O.u8p), want.size)
        assert np.array_equal(got, want), k
    assert (st.left.prev1, st.left.prev2, st.right.prev1, st.right.prev2) == (ost.left.prev1, ost.left.prev2, ost.right.prev1, ost.right.prev2)


def test_xa_many_channels_like_xacd_config():
    """config 'xacd' shape at reduced length: 8 XA channels x stereo, 37800 Hz 4-bit, several sectors each"""
    from psxavenc_amd import adpcm
    s = adpcm.XaSettings(1, True, 37800, 4, 1, 0)
    n = 2016 * 6
    pcm = np.stack([stereo_pad(c % 3, n, 50 + c, pad=0) for c in range(8)])
    out = adpcm.xa_encode_streams(s, pcm, n, lbas=np.arange(8) * 100, finalize=True)
    for c in range(8):
        want, _ = O.xa_encode(O.XaSettings(1, 1, 37800, 4, 1, 0), np.concatenate([pcm[c], np.zeros(8064, np.int16)]), n, lba=c * 100)
        O.lib().orc_xa_encode_finalize(O.XaSettings(1, 1, 37800, 4, 1, 0), O.ptr(want, O.u8p), want.size)
        assert np.array_equal(out[c], want), c


def test_against_reference_build_when_shipped():
    """oracle/_ref/libpsxav_ref.so is the reference's own code; it travels with gpurun"""
    if O.ref() is None:
        pytest.skip("oracle/_ref not present")
    from psxavenc_amd import adpcm
    rng = np.random.default_rng(7)
    for trial in range(10):
        n = int(rng.integers(30, 4000))
        kind = int(rng.integers(0, 6))
        pcm = O.synth_pcm(int(rng.integers(1, 1 << 30)), trial, 0, n, kind)
        want, _ = O.ref_spu_encode(pcm)
        got = adpcm.spu_encode_streams(pcm.reshape(1, -1))[0]
        assert np.array_equal(got, want), trial
