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


# ---------------------------------------------------------------- speculate-and-verify along time
@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4, 5])
def test_chunked_equals_serial_every_signal_class(kind):
    """every chunk is encoded from a guessed state, verify passes repair the wrong guesses; the fixpoint must be the
    serial encode bit for bit -- including class 4 (pure tone), where guesses essentially never coincide"""
    import torch
    from psxavenc_amd import adpcm
    n_units = 700
    n = n_units * 28
    streams = 3
    pcm = np.stack([O.synth_pcm(123, c, 77 * c, n, kind) for c in range(streams)])
    d = torch.from_numpy(pcm).to("cuda:0")
    chains = adpcm.make_chains(np.arange(streams) * n, 1, n, n_units)
    base = np.arange(streams, dtype=np.int32) * n_units
    d_units, d_states, passes = adpcm.encode_chains_device(d.reshape(-1), chains, base, 5, 4, chunk_units=32, warmup_units=8)
    assert passes >= 1
    got = adpcm.spu_pack_device(d_units, streams * n_units).cpu().numpy().reshape(streams, -1)
    st = d_states.cpu().numpy()
    for c in range(streams):
        want, wst = O.spu_encode(pcm[c])
        assert np.array_equal(got[c], want), (kind, c, passes)
        assert st[c].tolist() == [wst.prev1, wst.prev2]


def test_chunked_respects_initial_state_ragged_lengths_and_xa_layout():
    """chains of different lengths, non-zero start states, stereo XA interleave (unit_stride 2, pitch 2), 8-bit"""
    import torch
    from psxavenc_amd import adpcm
    # SPU: ragged lengths, carried-in state
    lens = [1, 31, 32, 33, 257]
    pcm = [O.synth_pcm(9, c, 0, 28 * u, 0) for c, u in enumerate(lens)]
    offs = np.cumsum([0] + [p.size for p in pcm[:-1]])
    d = torch.from_numpy(np.concatenate(pcm)).to("cuda:0")
    chains = adpcm.make_chains(offs, 1, [p.size for p in pcm], lens)
    base = np.cumsum([0] + lens[:-1]).astype(np.int32)
    st0 = np.array([[100 * c, -50 * c] for c in range(len(lens))], np.int32)
    d_states = torch.from_numpy(st0.copy()).to("cuda:0")
    d_units, d_states, _ = adpcm.encode_chains_device(d, chains, base, 5, 4, d_states=d_states, chunk_units=16, warmup_units=4)
    got = adpcm.spu_pack_device(d_units, sum(lens)).cpu().numpy().reshape(-1)
    for c, u in enumerate(lens):
        want, wst = O.spu_encode(pcm[c], state=O.Chan(int(st0[c, 0]), int(st0[c, 1])))
        assert np.array_equal(got[base[c] * 16:(base[c] + u) * 16], want), c
        assert d_states.cpu().numpy()[c].tolist() == [wst.prev1, wst.prev2]
    # XA 8-bit stereo, 3 sectors
    s = adpcm.XaSettings(1, True, 37800, 8, 2, 5)
    n = 1008 * 3
    inter = stereo_pad(0, n, 31, pad=0)
    d = torch.from_numpy(inter).to("cuda:0")
    units_per_chain = 3 * 18 * 4 // 2
    chains = adpcm.make_chains([0, 1], 2, n, units_per_chain, unit_stride=2)
    d_units, _, _ = adpcm.encode_chains_device(d, chains, np.array([0, 1], np.int32), 4, 8, chunk_units=10, warmup_units=6)
    got = adpcm.xa_assemble_device(d_units, 3, s, first_lba=40).cpu().numpy().reshape(-1)
    want, _ = O.xa_encode(O.XaSettings(1, 1, 37800, 8, 2, 5), np.concatenate([inter, np.zeros(8064, np.int16)]), n, lba=40)
    assert np.array_equal(got, want)


def test_chunked_long_chain_property():
    """config 'xacd' chain shape at reduced length (200k units per chain x 4 chains, XA filter set)"""
    import torch
    from psxavenc_amd import adpcm, synth
    n_units = 200000
    n = n_units * 28
    d = torch.empty((4, n), dtype=torch.int16, device="cuda:0")
    for c in range(4):
        synth.pcm_device(5, c, 0, n, 0, out=d[c])
    chains = adpcm.make_chains(np.arange(4) * n, 1, n, n_units)
    base = np.arange(4, dtype=np.int32) * n_units
    d_units, d_states, passes = adpcm.encode_chains_device(d.reshape(-1), chains, base, 4, 4, chunk_units=64, warmup_units=16)
    assert 1 <= passes < 200, passes      # tonal material converges slowly; the count only affects speed
    got = adpcm.spu_pack_device(d_units, 4 * n_units).cpu().numpy().reshape(4, -1)
    pcm = d.cpu().numpy()
    for c in (0, 3):
        # XA uses 4 filters: compare unit by unit through the oracle's unit encoder packed like SPU blocks
        import ctypes as C
        st = O.Chan(0, 0)
        out = np.zeros(n_units * 16, np.uint8)
        codes = (C.c_uint8 * 28)()
        L = O.lib()
        L.orc_adpcm_encode_unit.argtypes = [C.POINTER(O.Chan), O.i16p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        L.orc_adpcm_encode_unit.restype = C.c_uint8
        x = np.ascontiguousarray(pcm[c])
        # serial oracle over the chain, one unit per call (XA filter set: 4 filters)
        hdrs = np.zeros(n_units, np.uint8)
        blocks = np.zeros((n_units, 14), np.uint8)
        for u in range(n_units):
            h = L.orc_adpcm_encode_unit(C.byref(st), O.ptr(x[u * 28:], O.i16p), n - u * 28, 1, 4, 12, codes)
            hdrs[u] = h
            cc = np.frombuffer(codes, np.uint8)
            blocks[u] = (cc[0::2] & 0x0F) | (cc[1::2] << 4)
            if u >= 3000 and c == 3:
                break
        upto = 3001 if c == 3 else n_units
        g = got[c].reshape(n_units, 16)
        assert np.array_equal(g[:upto, 0], hdrs[:upto]) and np.array_equal(g[:upto, 2:], blocks[:upto])
        if c == 0:
            assert d_states.cpu().numpy()[0].tolist() == [st.prev1, st.prev2]


def test_time_sharded_sessions_simulated_ranks():
    """8(e) for ADPCM: chains sharded ALONG TIME over 4 (simulated) ranks on one GPU -- each rank's session guesses its
    start state from the units before its range, ranks exchange final states until a round changes nothing"""
    import torch
    from psxavenc_amd import adpcm
    from psxavenc_amd.parallel import shard_range, simulate_time_sharded
    world, n_units, n_chains = 4, 1501, 3
    n = n_units * 28
    pcm = np.stack([O.synth_pcm(77, c, 0, n, k) for c, k in enumerate((0, 4, 5))])
    d = torch.from_numpy(pcm).to("cuda:0").reshape(-1)
    d_units = torch.zeros((n_chains * n_units, adpcm.record_bytes(4)), dtype=torch.uint8, device="cuda:0")
    sessions = []
    for r in range(world):
        first, count = shard_range(n_units, r, world)
        chains = adpcm.make_chains(np.arange(n_chains) * n + first * 28, 1, n - first * 28, count)
        base = (np.arange(n_chains) * n_units + first).astype(np.int32)
        sessions.append(adpcm.AdpcmSession(d, chains, base, 5, 4, d_units=d_units, lead_units=np.full(n_chains, first, np.int32),
                                           chunk_units=32, warmup_units=8))
    init = np.array([[0, 0], [300, -300], [0, 0]], np.int32)
    final = simulate_time_sharded(sessions, init)
    got = adpcm.spu_pack_device(d_units, n_chains * n_units).cpu().numpy().reshape(n_chains, -1)
    for c in range(n_chains):
        want, wst = O.spu_encode(pcm[c], state=O.Chan(int(init[c, 0]), int(init[c, 1])))
        assert np.array_equal(got[c], want), c
        assert final[c].tolist() == [wst.prev1, wst.prev2]
    for s in sessions:
        s.close()


def test_long_host_stream_takes_the_chunked_path_and_matches():
    """psxhip_spu_encode_streams_host switches to speculate-and-verify for streams >= 4096 units"""
    from psxavenc_amd import adpcm
    n = 28 * 6000 + 11
    pcm = np.stack([O.synth_pcm(31, c, 0, n, c) for c in (0, 2)])
    st = np.array([[10, 20], [-30, 40]], np.int32)
    st0 = st.copy()
    out = adpcm.spu_encode_streams(pcm, states=st)
    for c in range(2):
        want, wst = O.spu_encode(pcm[c], state=O.Chan(int(st0[c, 0]), int(st0[c, 1])))
        assert np.array_equal(out[c], want), c
        assert st[c].tolist() == [wst.prev1, wst.prev2]
    # XA, long
    s = adpcm.XaSettings(0, True, 18900, 4, 0, 0)
    ns = 2016 * 300
    x = stereo_pad(0, ns, 3, pad=0)
    got = adpcm.xa_encode_streams(s, x.reshape(1, -1), ns, lbas=[7])[0]
    want, _ = O.xa_encode(O.XaSettings(0, 1, 18900, 4, 0, 0), np.concatenate([x, np.zeros(8064, np.int16)]), ns, lba=7)
    assert np.array_equal(got, want)


def test_empty_and_tiny_inputs():
    """edge cases: zero streams / zero samples return empty results; a single sample still yields one block"""
    from psxavenc_amd import adpcm
    assert adpcm.spu_encode_streams(np.zeros((1, 0), np.int16)).shape == (1, 0)
    one = adpcm.spu_encode_streams(np.array([[1234]], np.int16))
    want, _ = O.spu_encode(np.array([1234], np.int16))
    assert np.array_equal(one[0], want)
    s = adpcm.XaSettings(1, False, 37800, 8, 0, 0)
    got = adpcm.xa_encode_streams(s, np.array([[5, -5, 7]], np.int16), 3)[0]
    want, _ = O.xa_encode(O.XaSettings(1, 0, 37800, 8, 0, 0), np.concatenate([np.array([5, -5, 7], np.int16), np.zeros(5000, np.int16)]), 3)
    assert np.array_equal(got, want)


def test_spu_file_framing_vs_reference_golden():
    """SPU / VAG / SPUI / VAGI files (psxhip_spu_file_encode_host: one batched GPU encode + host framing) against the
    files the REFERENCE's own psx_audio_spu_encode produces when driven through filefmt.c's framing
    (tests/golden/spufile_ref.npz, generated by tests/golden/make_spufile_golden.py from oracle/_ref): 88 cases --
    leading dummy on/off, loop point, loop flag, alignments, ragged tails, 1/2/4 channels, short first chunks."""
    import hashlib
    import importlib.util
    from psxavenc_amd import spufile
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_spufile_golden", os.path.join(here, "golden", "make_spufile_golden.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    gold = np.load(os.path.join(here, "golden", "spufile_ref.npz"))
    assert len(G.CASES) == len(gold["keys"]) == 88
    for key, fmt, opts, rec in G.CASES:
        pcm = G.interleaved_pcm(rec["seed"], rec["kind"], rec["n"], rec["channels"])
        s = spufile.settings(fmt, channels=rec["channels"], interleave=opts.get("interleave", 2048),
                             alignment=opts.get("alignment"), loop_point=opts.get("loop_point", -1),
                             enable_loop=opts.get("enable_loop", False), no_dummy=opts.get("no_dummy", False))
        got = spufile.encode(s, pcm)
        assert got.size == int(gold[key + "_size"][0]), key
        if key + "_bytes" in gold:
            assert np.array_equal(got, gold[key + "_bytes"]), key
        assert hashlib.sha256(got.tobytes()).digest() == gold[key + "_sha"].tobytes(), key
    # SURVEY 8(d) config 1 `spu`: the whole 12 672-byte file, produced by the product's framing
    sine = G.sine_spu_config()
    got = spufile.encode(spufile.settings(spufile.FORMAT_SPU), sine)
    assert got.size == 12672 and got[16:20].tolist() == [0x24, 0x00, 0x70, 0x13]
    assert np.array_equal(got, gold["config_spu_sine"])


def test_spu_strided_input_is_not_read_past_its_last_sample():
    """the reference reads samples[i * pitch] for i < n (adpcm.c:65,110): with `samples + channel` and pitch = channels the
    caller's buffer ends (n - 1) * pitch + 1 elements after the pointer.  Put that end on a page boundary with an
    inaccessible page behind it: the host path must not touch it, and the output must equal the contiguous encode."""
    import ctypes as C
    import mmap
    from psxavenc_amd import _lib
    L = _lib.lib()
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    page = mmap.PAGESIZE
    n, pitch = 28 * 40, 2
    need = ((n - 1) * pitch + 1) * 2                      # bytes the reference may read from `samples + 1`
    m = mmap.mmap(-1, 4 * page)
    base = C.addressof(C.c_char.from_buffer(m))
    assert libc.mprotect(base + 3 * page, page, 0) == 0  # PROT_NONE behind the data
    start = base + 3 * page - need                        # the strided view ends exactly at the page boundary
    mono = O.synth_pcm(17, 0, 0, n, 0)
    view = (C.c_int16 * ((n - 1) * pitch + 1)).from_address(start)
    for i in range(n):
        view[i * pitch] = int(mono[i])

    class Chan(C.Structure):
        _fields_ = [("qerr", C.c_int), ("mse", C.c_uint64), ("prev1", C.c_int), ("prev2", C.c_int)]
    st = Chan()
    out = np.zeros(n // 28 * 16, np.uint8)
    L.psx_audio_spu_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    ln = L.psx_audio_spu_encode(C.byref(st), start, n, pitch, out.ctypes.data)
    want, _ = O.spu_encode(mono)
    assert ln == want.size and np.array_equal(out, want)
    libc.mprotect(base + 3 * page, page, 3)
    del view
    m.close()


def test_extreme_signals_where_most_candidates_saturate():
    """Full-scale alternations, rails and spikes: most (filter, shift) candidates clip and their squared error passes 2^32.
    The kernel sums the error in 32 bits with saturation -- exact for the arg-min because filter 0 at its minimum shift
    never clips -- and computes the minimum shift in closed form; every byte must still equal the oracle (SPU: 5 filters,
    XA 4-bit and 8-bit)."""
    from psxavenc_amd import adpcm
    rng = np.random.default_rng(77)
    n = 28 * 600
    sigs = [
        np.where(np.arange(n) % 2 == 0, 32767, -32768),
        np.where((np.arange(n) // 3) % 2 == 0, 32767, -32768),
        np.where((np.arange(n) // 28) % 2 == 0, -32768, 32767),
        rng.choice(np.array([-32768, -1, 0, 1, 32767]), n),
        np.where(rng.random(n) < 0.05, rng.choice(np.array([-32768, 32767]), n), rng.integers(-40, 40, n)),
        np.full(n, -32768), np.full(n, 32767),
    ]
    for k, sig in enumerate(sigs):
        pcm = sig.astype(np.int16)
        want, _ = O.spu_encode(pcm)
        got = adpcm.spu_encode_streams(pcm[None, :], sample_count=n)[0][:want.size]
        assert np.array_equal(got, want), ("spu", k)
        for bits in (4, 8):
            s, so = adpcm.XaSettings(1, False, 37800, bits, 0, 0), O.XaSettings(1, 0, 37800, bits, 0, 0)
            padded = np.concatenate([pcm, np.zeros(8064, np.int16)])
            want, _ = O.xa_encode(so, padded, n, lba=3)
            got = adpcm.xa_encode_streams(s, padded[None, :], n, lbas=np.array([3], np.int32))[0]
            assert got.size == want.size and np.array_equal(got, want), ("xa", bits, k)


# ---------------------------------------------------------------- several devices behind one call
def test_xa_streams_over_a_device_list_equal_single_device():
    """psxhip_xa_encode_streams_host_multi over {0, 0, 0}: the 8 XA channels of config 'xacd' (reduced length) sharded
    3 / 3 / 2 over three host threads -- bytes and carried states of the single-device call"""
    from psxavenc_amd import adpcm, multi
    s = adpcm.XaSettings(1, True, 37800, 4, 1, 0)
    n = 2016 * 40 + 700
    pcm = np.stack([stereo_pad(c % 3, n, 90 + c, pad=0) for c in range(8)])
    lbas = np.arange(8, dtype=np.int32) * 1000
    st1 = np.zeros((8, 2, 2), np.int32)
    want = adpcm.xa_encode_streams(s, pcm, n, lbas=lbas, states=st1, finalize=True)
    got, st2, rep = multi.xa_encode_streams_multi((0, 0, 0), s, pcm, lbas=lbas, finalize=True)
    assert np.array_equal(got, want)
    assert np.array_equal(st2.reshape(8, 2, 2), st1)
    assert [r["units"] for r in rep] == [3, 3, 2]
    one, _, _ = multi.xa_encode_streams_multi((0, 0, 0), s, pcm[:1], lbas=lbas[:1], finalize=True)      # fewer streams than devices
    assert np.array_equal(one[0], want[0])


def test_xacd_config5_full_size_every_sector_against_the_reference():
    """BASELINE config 5 at its full size: 8 XA channels x stereo x 37800 Hz x 60 min = 540 000 sectors (77.76 M sound
    units) in one speculate-and-verify session on one GPU.  EVERY sector of every channel is compared with the reference's
    own psx_audio_xa_encode (oracle/_ref, libpsxav/adpcm.c compiled unchanged; the restatement when that is absent) run
    serially on the host, one channel per thread; plus the size-independent properties: time codes, subheaders, EDC."""
    import threading
    import torch
    from psxavenc_amd import adpcm, synth
    from psxavenc_amd.parallel import run_time_sharded
    settings = adpcm.XaSettings(adpcm.PSX_AUDIO_XA_FORMAT_XACD, True, 37800, 4, 1, 0)
    sps = adpcm.xa_get_samples_per_sector(settings)
    n_ch, n_sectors = 8, 3600 * 37800 // sps
    assert n_sectors == 67500 and n_ch * n_sectors == 540000
    n = n_sectors * sps
    dev = torch.device("cuda", 0)
    pcm = torch.empty((n_ch, n * 2), dtype=torch.int16, device=dev)
    for c in range(n_ch):
        for side in range(2):
            # loud two-tone + noise, quiet tone, full-scale noise, half-silent -- a different class per chain
            synth.pcm_device(1, 2 * c + side, 0, n, (0, 1, 2, 5)[(2 * c + side) % 4], device=0, out=pcm[c][side:], pitch=2)
    chains = adpcm.make_chains([c * n * 2 + side for c in range(n_ch) for side in range(2)], 2, n, n_sectors * 72, unit_stride=2)
    base = np.array([c * n_sectors * 144 + side for c in range(n_ch) for side in range(2)], np.int32)
    chunk_units, warmup_units = adpcm.pick_chunking(int(chains["n_units"].sum()))
    sess = adpcm.AdpcmSession(pcm.reshape(-1), chains, base, 4, 4, chunk_units=chunk_units, warmup_units=warmup_units)
    run_time_sharded(sess, 0, 1, None, np.zeros((2 * n_ch, 2), np.int32))
    outs = [adpcm.xa_assemble_device(sess.d_units[c * n_sectors * 144:], n_sectors, settings, first_lba=150 + c * n_sectors) for c in range(n_ch)]
    torch.cuda.synchronize()
    got = [o.cpu().numpy() for o in outs]
    host_pcm = pcm.cpu().numpy()
    sess.close()
    del pcm, outs

    use_ref = O.ref() is not None
    os_ = O.XaSettings(1, 1, 37800, 4, 1, 0)
    bad = {}

    def check(c):
        x = np.concatenate([host_pcm[c], np.zeros(8064, np.int16)])
        enc = O.ref_xa_encode if use_ref else O.xa_encode
        want, _ = enc(os_, x, n, lba=150 + c * n_sectors)
        w = want.reshape(n_sectors, 2352)
        if not np.array_equal(got[c], w):
            bad[c] = np.nonzero((got[c] != w).any(axis=1))[0][:5].tolist()

    ths = [threading.Thread(target=check, args=(c,)) for c in range(n_ch)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not bad, "sectors differ from the %s: %s" % ("reference build" if use_ref else "oracle", bad)
    # properties on everything: sync pattern, BCD time code of lba 150 + k, mode 2, subheader twice, EDC over 0x10..0x92B
    for c in (0, 7):
        g = got[c]
        assert (g[:, 0] == 0).all() and (g[:, 1:11] == 0xFF).all() and (g[:, 11] == 0).all() and (g[:, 15] == 2).all()
        lba = 150 + (150 + c * n_sectors + np.arange(n_sectors))      # psx_cdrom_init_sector adds the 2-second lead-in (cdrom.c:62)
        m, s_, f = lba // 4500, (lba // 75) % 60, lba % 75
        bcd = lambda v: ((v // 10) << 4) | (v % 10)
        assert np.array_equal(g[:, 12], bcd(m)) and np.array_equal(g[:, 13], bcd(s_)) and np.array_equal(g[:, 14], bcd(f))
        assert np.array_equal(g[:, 16:20], g[:, 20:24]) and (g[:, 18] == 0x64).all()
        for k in range(0, n_sectors, 337):
            edc = O.lib().orc_edc_crc32(O.ptr(np.ascontiguousarray(g[k, 16:0x92C]), O.u8p), 0x92C - 16)
            assert int.from_bytes(g[k, 0x92C:0x930].tobytes(), "little") == edc


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_per_call_path_speculates_inside_one_launch_and_equals_the_serial_encode(kind):
    """The reference's call pattern (filefmt.c:184,243,335: one sector / 28 samples / one SPUI chunk per call) is ONE launch
    (adpcm_call_kernel), whose spare rows speculate on later segments of the chain and whose result must be the serial encode's
    whatever the guesses: every signal class (tones never fall into the guessed state, silence ties, noise saturates), chain
    lengths around the speculation limits, 1-4 streams, carried states, XA 4/8-bit mono/stereo one and two sectors per call."""
    from psxavenc_amd import adpcm
    # SPU: one chain -> four segments; 2 chains -> two each; 3, 4 chains -> none
    for n_streams, units in [(1, 1), (1, 23), (1, 24), (1, 25), (1, 97), (1, 128), (1, 290), (2, 40), (2, 73), (2, 130), (3, 60), (4, 70)]:
        n = units * 28 - (5 if units % 3 == 1 and units > 1 else 0)            # ragged tails too
        pcm = np.stack([O.synth_pcm(500 + kind, s, 7 * units, n, kind) for s in range(n_streams)])
        st = np.array([[(-1) ** s * 300 * s, 17 * s] for s in range(n_streams)], np.int32)
        st0 = st.copy()
        got = adpcm.spu_encode_streams(pcm, states=st)
        for s in range(n_streams):
            want, ost = O.spu_encode(pcm[s], state=O.Chan(int(st0[s, 0]), int(st0[s, 1])))
            assert np.array_equal(got[s], want), (kind, n_streams, units, s)
            assert (int(st[s, 0]), int(st[s, 1])) == (ost.prev1, ost.prev2), (kind, n_streams, units, s)
    # XA: sector by sector and two sectors at a time, state carried
    for stereo, bits, per_call in [(True, 4, 1), (True, 4, 2), (False, 4, 1), (True, 8, 1), (True, 8, 3), (False, 8, 2)]:
        s = adpcm.XaSettings(adpcm.PSX_AUDIO_XA_FORMAT_XACD, stereo, 37800, bits, 1, 0)
        os_ = O.XaSettings(1, int(stereo), 37800, bits, 1, 0)
        sps = adpcm.xa_get_samples_per_sector(s)
        calls = 3
        n = sps * per_call * calls - 333                                        # the last call is short
        pcm = stereo_pad(kind, n, 600 + kind) if stereo else mono_pad(kind, n, 600 + kind)
        ch = 2 if stereo else 1
        st, ost = adpcm.EncoderState(), O.State()
        for k in range(calls):
            first = k * sps * per_call
            cnt = min(sps * per_call, n - first)
            chunk = pcm[ch * first:]
            got = adpcm.psx_audio_xa_encode(s, st, chunk, cnt, k * per_call)
            want, ost = O.xa_encode(os_, chunk, cnt, lba=k * per_call, state=ost)
            assert np.array_equal(got, want), (kind, stereo, bits, per_call, k)
        assert (st.left.prev1, st.left.prev2) == (ost.left.prev1, ost.left.prev2)
        if stereo:
            assert (st.right.prev1, st.right.prev2) == (ost.right.prev1, ost.right.prev2)


def test_mid_size_single_streams_take_the_chunked_path_and_match():
    """round 5: a call with few chains cuts them along time from 512 units on (a lone chain encoded serially runs at one wavefront's
    pace); lengths either side of the switch, SPU and XA, every signal class, state carried in and out"""
    from psxavenc_amd import adpcm
    for kind in (0, 2, 4, 5):
        for n_units in (500, 511, 512, 513, 788, 1500):
            n = 28 * n_units - (5 if n_units % 2 else 0)
            pcm = O.synth_pcm(61, kind, 0, n, kind).reshape(1, -1)
            st = np.array([[123, -77]], np.int32)
            out = adpcm.spu_encode_streams(pcm, states=st.copy())
            want, wst = O.spu_encode(pcm[0], state=O.Chan(123, -77))
            assert np.array_equal(out[0], want), (kind, n_units)
    s = adpcm.XaSettings(1, True, 37800, 4, 1, 0)
    for sectors in (7, 8, 20):
        ns = 2016 * sectors - 300
        x = stereo_pad(5, ns, 3, pad=4032)
        got = adpcm.xa_encode_streams(s, x.reshape(1, -1), ns, lbas=[11])[0]
        os_ = O.XaSettings(1, 1, 37800, 4, 1, 0)
        want, _ = O.xa_encode(os_, x, ns, lba=11)
        assert np.array_equal(got, want), sectors
