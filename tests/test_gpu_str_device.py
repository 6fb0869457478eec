"""psxhip_str_encode_device (frames and PCM resident in HBM -> muxed sectors in HBM; one or several independent streams per call):
every sector against encode_file_str restated over the oracle (tests/str_reference_loop.py: filefmt.c:391-520, decoding.c:510-586) and
against the host-buffer path of the same library.  Bar: bit-exact."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _pcm(channels, n, seed, kind=0):
    pcm = np.zeros(n * max(1, channels), np.int16)
    for c in range(channels):
        pcm[c::channels] = O.synth_pcm(seed, c, 0, n, kind) if n else 0
    return pcm


@pytest.mark.parametrize("fmt,codec,w,h,n_frames,channels", [(7, 0, 320, 240, 160, 2), (6, 1, 160, 112, 40, 1), (9, 2, 96, 64, 30, 0)])
def test_one_stream_whole_stream_vs_reference_loop(fmt, codec, w, h, n_frames, channels):
    import torch
    import str_reference_loop as R
    from psxavenc_amd import strmux
    s = strmux.settings(fmt=fmt, codec=codec, width=w, height=h, channels=channels, frequency=37800, bits=4)
    frames = O.synth_frames(w, h, n_frames, seed=21, amp=6)
    n = 0
    if channels:
        pl = strmux.plan(s, n_frames)
        n = (pl.n_audio_sectors + 2) * pl.audio_samples_per_sector + 100
    pcm = _pcm(channels, n, 9)
    mux = strmux.StrMuxer((0,))
    d_frames = torch.from_numpy(frames).to("cuda:0")
    d_pcm = torch.from_numpy(pcm).to("cuda:0") if channels else None
    for rep in range(2):          # (the second call runs on the tables and buffers the first one left in the handle)
        d_out, p = mux.encode_device(s, d_frames, d_pcm)
        got = d_out.cpu().numpy()[0]
        want, qsum, frames_encoded = R.encode_file_str(fmt, codec, w, h, 15, 1, 2, frames, pcm, channels=channels)
        assert got.shape == want.shape, (got.shape, want.shape)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, "sectors differ: %s" % bad[:8].tolist()
        assert p.quant_scale_sum == qsum and p.n_frames_encoded == frames_encoded
    host, _ = mux.encode(s, frames, pcm)
    assert np.array_equal(host, got)
    mux.close()


def test_several_streams_in_one_call_equal_one_call_per_stream():
    """S independent streams (different pictures, different audio -- tonal, noise, gated) of config 3's shape in one call: the S x 2
    XA chains share the verify passes; every stream's sectors equal the host-buffer call on that stream alone"""
    import torch
    from psxavenc_amd import strmux
    w, h, n_frames, S = 320, 240, 120, 5
    s = strmux.settings()
    pl = strmux.plan(s, n_frames)
    n = (pl.n_audio_sectors + 2) * 2016 + 100
    frames = np.stack([O.synth_frames(w, h, n_frames, seed=30 + i, amp=3 + 2 * i) for i in range(S)])
    pcm = np.stack([_pcm(2, n, 40 + i, kind=[0, 2, 5, 4, 1][i]) for i in range(S)])
    mux = strmux.StrMuxer((0,))
    d_out, p = mux.encode_device(s, torch.from_numpy(frames).to("cuda:0"), torch.from_numpy(pcm).to("cuda:0"))
    got = d_out.cpu().numpy()
    qsum = 0
    for i in range(S):
        want, pi = mux.encode(s, frames[i], pcm[i])
        assert np.array_equal(got[i], want), (i, np.nonzero((got[i] != want).any(axis=1))[0][:6].tolist())
        qsum += pi.quant_scale_sum
    assert p.quant_scale_sum == qsum
    # a strided layout (streams further apart than their frames: the batch-list route), into a caller's buffer
    pad = np.zeros((S, n_frames + 3, frames.shape[2]), np.uint8)
    pad[:, :n_frames] = frames
    d_pad = torch.from_numpy(pad).to("cuda:0")
    d_out2 = torch.zeros_like(d_out)
    mux2 = strmux.StrMuxer((0,))
    import ctypes as C
    from psxavenc_amd import _lib
    p2 = strmux.StrPlan()
    d_p = torch.from_numpy(pcm).to("cuda:0")
    rc = strmux._bind().psxhip_str_encode_device(mux2._h, C.byref(s), S, d_pad.data_ptr(), d_pad.stride(0), n_frames, d_p.data_ptr(), d_p.stride(0),
                                                 n, d_out2.data_ptr(), d_out2.stride(0), C.byref(p2), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc)
    assert torch.equal(d_out2, d_out)
    mux.close()
    mux2.close()


@pytest.mark.parametrize("n_audio_sectors_x10", [0, 5, 30])
def test_audio_shorter_than_video_and_shape_changes_on_one_handle(n_audio_sectors_x10):
    """the reference's stream ends with whichever input ends first: short last sector completed from zeros, empty audio slots
    (zero sectors), EOF flags -- and a handle that is reused for another shape right after"""
    import torch
    import str_reference_loop as R
    from psxavenc_amd import strmux
    w, h, n_frames = 160, 112, 24
    s = strmux.settings(fmt=7, codec=0, width=w, height=h)
    frames = O.synth_frames(w, h, n_frames, seed=5, amp=6)
    n = 2016 * n_audio_sectors_x10 // 10
    pcm = _pcm(2, n, 11)
    mux = strmux.StrMuxer((0,))
    d_frames = torch.from_numpy(frames).to("cuda:0")
    d_pcm = torch.from_numpy(pcm).to("cuda:0") if n else torch.zeros((1, 0), dtype=torch.int16, device="cuda:0")
    d_out, p = mux.encode_device(s, d_frames, d_pcm)
    want, qsum, frames_encoded = R.encode_file_str(7, 0, w, h, 15, 1, 2, frames, pcm)
    got = d_out.cpu().numpy()[0]
    assert got.shape == want.shape and np.array_equal(got, want), (got.shape, want.shape)
    assert (p.quant_scale_sum, p.n_frames_encoded) == (qsum, frames_encoded)
    # another shape on the same handle (fewer frames, plenty of audio)
    pcm2 = _pcm(2, 2016 * 8, 12)
    d_out, p = mux.encode_device(s, d_frames[:10].contiguous(), torch.from_numpy(pcm2).to("cuda:0"))
    want, qsum, frames_encoded = R.encode_file_str(7, 0, w, h, 15, 1, 2, frames[:10], pcm2)
    assert np.array_equal(d_out.cpu().numpy()[0], want)
    mux.close()


def test_config3_at_1000_frames_device_resident():
    """BASELINE config 3 at its full size, device-resident: 1000 frames 320x240 @15 fps + 37800 Hz 4-bit stereo XA (the tonal test
    signal: its XA track needs the verify passes) -> 9981 STRCD sectors, every sector against the reference's sector loop
    (tests/str_reference_loop.py: filefmt.c:391-520) with the XA sectors from the reference's own psx_audio_xa_encode (libpsxav/adpcm.c
    compiled unchanged into oracle/_ref) where that build travelled, else from the restatement"""
    import torch
    import str_reference_loop as R
    from psxavenc_amd import strmux
    w, h, n_frames = 320, 240, 1000
    s = strmux.settings()
    frames = O.synth_frames(w, h, n_frames, seed=1, amp=4)
    n = 2016 * 1260
    pcm = _pcm(2, n, 1)
    mux = strmux.StrMuxer((0,))
    d_out, p = mux.encode_device(s, torch.from_numpy(frames).to("cuda:0"), torch.from_numpy(pcm).to("cuda:0"))
    assert p.n_frames_encoded == 998 and p.n_sectors == d_out.shape[1]
    want, qsum, frames_encoded = R.encode_file_str(7, 0, w, h, 15, 1, 2, frames, pcm, xa_encode=O.ref_xa_encode if O.ref() is not None else None)
    got = d_out.cpu().numpy()[0]
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "sectors differ: %s" % bad[:8].tolist()
    assert (p.quant_scale_sum, p.n_frames_encoded) == (qsum, frames_encoded)
    mux.close()
