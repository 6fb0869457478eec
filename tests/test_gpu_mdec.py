"""GPU parity tests of the MDEC BS frame encoder: libpsxav_hip.so (through its C ABI) vs the CPU oracle on the
same seeded inputs, vs the committed vectors, and -- at BASELINE.json's full sizes -- through size-independent
properties.  Bar: bit-exact (integer path)."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mdec_selfgolden.npz")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def encoder(codec, w, h, budget):
    from psxavenc_amd.mdec import MdecEncoder
    return MdecEncoder(codec, w, h, max_frame_size=budget, device=0)


def assert_same(got, got_res, want, want_res, tag=""):
    assert np.array_equal(got_res, want_res), (tag, got_res[:4], want_res[:4])
    if not np.array_equal(got, want):
        bad = np.nonzero((got != want).any(axis=1))[0]
        k = int(bad[0])
        at = int(np.nonzero(got[k] != want[k])[0][0])
        raise AssertionError("%s: %d frames differ; frame %d first differs at byte %d" % (tag, bad.size, k, at))


def test_library_is_the_hip_build():
    from psxavenc_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    assert b"gfx950" in _lib.lib().psxhip_version()


def test_selfgolden_matrix_all_cases():
    """every (codec, size, budget, content) case of the committed matrix, against the committed hashes AND the live oracle"""
    g = np.load(GOLD)
    from psxavenc_amd import _lib
    for row in g["table"]:
        codec, w, h, budget, amp, n, rc = (int(v) for v in row[:7])
        fr = O.synth_frames(w, h, n, seed=100 + amp, amp=amp, first=3)
        enc = encoder(codec, w, h, budget)
        if rc != 0:
            with pytest.raises(_lib.PsxHipError) as e:
                enc.encode_frames_host(fr, budget)
            assert e.value.code == _lib.PSXHIP_ENOFIT
        else:
            out, res = enc.encode_frames_host(fr, budget)
            want, want_res, orc = O.mdec_encode(codec, w, h, fr, budget)
            assert orc == 0
            assert_same(out, res, want, want_res, str(row[:7]))
            assert res.ravel().tolist() == row[7:7 + 4 * n].tolist(), row[:7]
            assert hashlib.sha256(out.tobytes()).digest() == g["sha_c%d_%dx%d_b%d_a%d" % (codec, w, h, budget, amp)].tobytes(), row[:7]
        enc.close()


def test_special_frames_vs_oracle():
    """flat fields (v3 DC ties in both directions), 8x8 checkerboard (DC deltas near +-255, v3dc wrap),
    hard edges (escape codes), DC staircase"""
    import sys
    sys.path.insert(0, os.path.join(O.ROOT, "tests", "golden"))
    from make_mdec_golden import special_frames
    g = np.load(GOLD)
    from psxavenc_amd import _lib
    for codec in (0, 1, 2):
        for (w, h, budget) in ((48, 32, 4096), (320, 240, 30000), (320, 240, 9000)):
            fr = special_frames(w, h)
            enc = encoder(codec, w, h, budget)
            key = "special_c%d_%dx%d_b%d" % (codec, w, h, budget)
            for k in range(fr.shape[0]):
                want, want_res, rc = O.mdec_encode(codec, w, h, fr[k:k + 1], budget)
                assert rc == int(g[key + "_rc"][k])
                if rc == 0:
                    out, res = enc.encode_frames_host(fr[k:k + 1], budget)
                    assert_same(out, res, want, want_res, "%s frame %d" % (key, k))
                    assert hashlib.sha256(out.tobytes()).digest() == g[key + "_sha"][k].tobytes()
                else:
                    with pytest.raises(_lib.PsxHipError):
                        enc.encode_frames_host(fr[k:k + 1], budget)
            enc.close()


def test_escape_codes_are_exercised():
    """the hard-edge frame must actually contain 22-bit escapes at its accepted scale (guards the test above)"""
    import sys
    sys.path.insert(0, os.path.join(O.ROOT, "tests", "golden"))
    from make_mdec_golden import special_frames
    fr = special_frames(320, 240)[8:9]
    out, res, rc = O.mdec_encode(0, 320, 240, fr, 30000)
    assert rc == 0
    _, levels, scale, _, _ = O.mdec_decode(320, 240, out[0])
    assert (np.abs(levels[:, 1:]) > 40).any()


@pytest.mark.parametrize("codec", [0, 1, 2])
def test_per_frame_budgets_str_cycle(codec):
    """strcd config: budgets cycle 16128, 18144, 18144, 18144, odd budgets mixed in"""
    w, h, n = 320, 240, 24
    fr = O.synth_frames(w, h, n, seed=42, amp=8)
    budgets = np.array([16128, 18144, 18144, 18144] * (n // 4), np.int32)
    budgets[5] = 8191
    budgets[6] = 4097
    want, want_res, rc = O.mdec_encode(codec, w, h, fr, budgets, stride=18144)
    assert rc == 0
    enc = encoder(codec, w, h, 18144)
    out, res = enc.encode_frames_host(fr, budgets)
    assert_same(out, res, want, want_res, "str cycle")
    enc.close()


def test_device_path_equals_host_path_and_is_stream_ordered(torch_cuda):
    torch = torch_cuda
    w, h, n, budget = 320, 240, 40, 8192
    fr = O.synth_frames(w, h, n, seed=5, amp=4)
    enc = encoder(0, w, h, budget)
    host_out, host_res = enc.encode_frames_host(fr, budget)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d = torch.from_numpy(fr).to("cuda:0", non_blocking=False)
        d_out, d_res = enc.encode_frames_device(d, budget)
        d_out2, d_res2 = enc.encode_frames_device(d, budget)     # back-to-back launches share the scratch slab
    s.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), host_out) and np.array_equal(d_res.cpu().numpy(), host_res)
    assert np.array_equal(d_out2.cpu().numpy(), host_out)
    enc.close()


def test_drop_in_single_frame_call_pattern():
    """encode_file_sbs's loop (filefmt.c:633-662): frame_max_size poked by the caller, one frame per call,
    quant_scale_sum accumulated across calls"""
    w, h, budget = 320, 240, 8192
    fr = O.synth_frames(w, h, 5, seed=77, amp=8)
    want, want_res, _ = O.mdec_encode(0, w, h, fr, budget)
    enc = encoder(0, w, h, budget)
    enc.frame_max_size = budget
    for k in range(5):
        out = enc.encode_frame_bs(fr[k])
        assert np.array_equal(out, want[k])
        assert [enc.quant_scale, enc.bytes_used, enc.blocks_used, enc.uncomp_hwords_used] == want_res[k].tolist()
    assert enc.quant_scale_sum == int(want_res[:, 0].sum())
    enc.close()


def test_full_size_sbs_v2_1000_frames(torch_cuda):
    """BASELINE config 'sbs v2': 1000 synthetic 320x240 frames, budget 8192 -- every byte against the oracle,
    with more frames than resident workgroups (persistent loop) and both content classes."""
    torch = torch_cuda
    from psxavenc_amd import synth
    w, h, n, budget = 320, 240, 1000, 8192
    enc = encoder(0, w, h, budget)
    for amp in (4, 8):
        d_frames = synth.frames_device(w, h, seed=1, first=0, n=n, amp=amp, device=0)
        d_out, d_res = enc.encode_frames_device(d_frames, budget)
        torch.cuda.synchronize()
        fr = d_frames.cpu().numpy()
        assert np.array_equal(fr[:3], O.synth_frames(w, h, 3, seed=1, amp=amp))
        want, want_res, rc = O.mdec_encode(0, w, h, fr, budget)
        assert rc == 0
        assert_same(d_out.cpu().numpy()[:, :budget], d_res.cpu().numpy(), want, want_res, "sbs v2 amp %d" % amp)
    enc.close()


def test_full_size_sbs_v3_640x480_properties(torch_cuda):
    """BASELINE config 'sbs v3' shape (640x480, one GPU's share): oracle on a sample, and size-independent
    properties on everything: header fields, zero tail, decodability, bits consistent with bytes_used."""
    torch = torch_cuda
    from psxavenc_amd import synth
    w, h, n, budget = 640, 480, 320, 32768
    enc = encoder(1, w, h, budget)
    d_frames = synth.frames_device(w, h, seed=2, first=5000, n=n, amp=8, device=0)
    d_out, d_res = enc.encode_frames_device(d_frames, budget)
    torch.cuda.synchronize()
    out, res = d_out.cpu().numpy(), d_res.cpu().numpy()
    assert ((res[:, 0] >= 1) & (res[:, 0] <= 63)).all()
    assert (out[:, 2] == 0).all() and (out[:, 3] == 0x38).all() and (out[:, 6] == 3).all() and (out[:, 7] == 0).all()
    assert np.array_equal(out[:, 4].astype(np.int32) | (out[:, 5].astype(np.int32) << 8), res[:, 0])
    assert np.array_equal(out[:, 0].astype(np.int32) | (out[:, 1].astype(np.int32) << 8), res[:, 2])
    for k in range(n):
        assert not out[k, res[k, 1]:].any()
    idx = list(range(0, n, 16))
    fr = d_frames[idx].cpu().numpy()
    want, want_res, rc = O.mdec_encode(1, w, h, fr, budget)
    assert rc == 0
    assert_same(out[idx][:, :budget], res[idx], want, want_res, "sbs v3 sample")
    for k in idx[:4]:
        rc2, levels, scale, version, nbits = O.mdec_decode(w, h, out[k])
        assert rc2 == 0 and version == 3 and scale == res[k, 0]
        assert res[k, 1] == ((8 + 2 * ((nbits + 15) // 16) + 3) & ~3)
        assert res[k, 3] == ((int(np.count_nonzero(levels[:, 1:])) + 2 * levels.shape[0] + 2 + 63) & ~63)
    enc.close()


def test_bad_arguments_fail_loudly():
    from psxavenc_amd import _lib
    from psxavenc_amd.mdec import MdecEncoder
    with pytest.raises(_lib.PsxHipError):
        MdecEncoder(0, 321, 240)
    with pytest.raises(_lib.PsxHipError):
        MdecEncoder(3, 320, 240)
    enc = MdecEncoder(0, 320, 240, max_frame_size=8192)
    with pytest.raises(_lib.PsxHipError):
        enc.encode_frames_host(np.zeros((1, 320 * 240 * 3 // 2), np.uint8), 9000)     # above the context's maximum
    enc.close()


def test_two_contexts_with_different_lds_needs_coexist():
    """the kernel's dynamic-LDS attribute is per kernel, not per context: a small context created after a large one must
    not break the large one (640x512 is the largest size the reference's CLI accepts, args.c:410-421)"""
    big = encoder(1, 640, 512, 40000)
    small = encoder(1, 48, 32, 4096)
    fr_b = O.synth_frames(640, 512, 2, seed=4, amp=8)
    fr_s = O.synth_frames(48, 32, 2, seed=4, amp=8)
    want_b, res_b, rc_b = O.mdec_encode(1, 640, 512, fr_b, 40000)
    want_s, res_s, rc_s = O.mdec_encode(1, 48, 32, fr_s, 4096)
    assert rc_b == 0 and rc_s == 0
    out_s, r_s = small.encode_frames_host(fr_s, 4096)
    out_b, r_b = big.encode_frames_host(fr_b, 40000)
    assert_same(out_b, r_b, want_b, res_b, "640x512")
    assert_same(out_s, r_s, want_s, res_s, "48x32")
    big.close()
    small.close()


def test_randomised_geometry_content_budgets_vs_oracle():
    """seeded fuzz: random sizes (multiples of 16 up to 128x96), codecs, per-frame budgets (odd ones too) and content
    mixes -- smooth ramps, noise of random amplitude, random flat 8x8 tiles (DC ties / big DC swings), sparse impulses
    (long zero runs -> run-length escapes).  Frames that fit no scale must be reported, everything else byte-exact."""
    from psxavenc_amd import _lib
    rng = np.random.default_rng(20260928)
    n_cases, n_nofit = 0, 0
    for case in range(60):
        w, h = 16 * int(rng.integers(1, 9)), 16 * int(rng.integers(1, 7))
        codec = int(rng.integers(0, 3))
        n = int(rng.integers(1, 6))
        npx = w * h
        frames = np.zeros((n, npx * 3 // 2), np.uint8)
        for k in range(n):
            kind = int(rng.integers(0, 4))
            yy, xx = np.mgrid[0:h, 0:w]
            if kind == 0:
                y = (xx * int(rng.integers(0, 4)) + yy * int(rng.integers(0, 4))) % 256 + rng.integers(-3, 4, (h, w))
            elif kind == 1:
                y = 128 + rng.integers(-int(rng.integers(1, 60)), int(rng.integers(1, 60)) + 1, (h, w))
            elif kind == 2:
                tiles = rng.integers(0, 256, (h // 8, w // 8))
                y = np.kron(tiles, np.ones((8, 8), np.int64))
            else:
                y = np.full((h, w), int(rng.integers(0, 256)))
                for _ in range(int(rng.integers(1, 12))):
                    y[int(rng.integers(0, h)), int(rng.integers(0, w))] = int(rng.integers(0, 256))
            frames[k, :npx] = np.clip(y, 0, 255).astype(np.uint8).ravel()
            frames[k, npx:] = np.clip(128 + rng.integers(-int(rng.integers(1, 40)), int(rng.integers(1, 40)) + 1, npx // 2), 0, 255).astype(np.uint8)
        nblk = (w // 16) * (h // 16) * 6
        floor_bytes = 8 + 2 * ((nblk * 12 + 10 + 15) // 16)
        budgets = rng.integers(floor_bytes, floor_bytes + int(rng.integers(16, 6000)), n).astype(np.int32)
        if case % 10 == 0:
            budgets[0] = floor_bytes - 2        # below the floor of 12 bits per block: no scale can fit
        enc = encoder(codec, w, h, int(budgets.max()))
        for k in range(n):
            want, want_res, rc = O.mdec_encode(codec, w, h, frames[k:k + 1], int(budgets[k]))
            if rc == 0:
                out, res = enc.encode_frames_host(frames[k:k + 1], int(budgets[k]))
                assert_same(out, res, want, want_res, "fuzz case %d frame %d (%dx%d codec %d budget %d)" % (case, k, w, h, codec, budgets[k]))
                n_cases += 1
            else:
                assert rc == -2
                with pytest.raises(_lib.PsxHipError) as e:
                    enc.encode_frames_host(frames[k:k + 1], int(budgets[k]))
                assert e.value.code == _lib.PSXHIP_ENOFIT
                n_nofit += 1
        enc.close()
    assert n_cases > 80 and n_nofit > 0


def test_out_of_range_device_budgets_are_contained(torch_cuda):
    """per-frame budgets come from device memory: values outside [8, context maximum] must flag the frame
    (quant_scale 64) without touching its output row or anyone else's"""
    torch = torch_cuda
    w, h, n, cap = 48, 32, 6, 4096
    fr = O.synth_frames(w, h, n, seed=2, amp=4)
    enc = encoder(1, w, h, cap)
    d = torch.from_numpy(fr).to("cuda:0")
    budgets = torch.tensor([4096, 1 << 20, 4096, 0, -5, 2048], dtype=torch.int32, device="cuda:0")
    d_out = torch.full((n, cap), 0xAB, dtype=torch.uint8, device="cuda:0")
    d_out, d_res = enc.encode_frames_device(d, budgets, d_out=d_out)
    torch.cuda.synchronize()
    out, res = d_out.cpu().numpy(), d_res.cpu().numpy()
    for k, b in enumerate([4096, None, 4096, None, None, 2048]):
        if b is None:
            assert res[k, 0] == 64 and (out[k] == 0xAB).all()
        else:
            want, want_res, rc = O.mdec_encode(1, w, h, fr[k:k + 1], b)
            assert rc == 0 and np.array_equal(out[k, :b], want[0]) and np.array_equal(res[k], want_res[0])
            assert (out[k, b:] == 0xAB).all()
    enc.close()


def test_long_mixed_batch_exercises_hint_checkpoint_and_retry_paths(torch_cuda):
    """thousands of small frames through one launch: every workgroup encodes many frames back to back (the previous
    frame's answer seeds the next one), content and per-frame budgets change abruptly (wrong seeds -> checkpoint aborts,
    extra passes), some frames are escape-heavy (the lower-bound proof is loose there).  Every byte against the oracle."""
    torch = torch_cuda
    rng = np.random.default_rng(4242)
    w, h, n = 192, 128, 2400                 # 96 macroblocks: 8 iterations per pass -> the checkpoint is active
    npx = w * h
    frames = np.zeros((n, npx * 3 // 2), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for k in range(n):
        kind = (k // 7) % 5                  # runs of similar frames, then a cut
        if kind == 0:
            y = (xx * 255 // w + yy * 255 // h) // 2 + rng.integers(-3, 4, (h, w))
        elif kind == 1:
            a = int(rng.integers(1, 50))
            y = 128 + rng.integers(-a, a + 1, (h, w))
        elif kind == 2:
            y = np.kron(rng.integers(0, 256, (h // 8, w // 8)), np.ones((8, 8), np.int64))
        elif kind == 3:
            y = np.full((h, w), int(rng.integers(0, 256)))
            for _ in range(int(rng.integers(1, 40))):
                y[int(rng.integers(0, h)), int(rng.integers(0, w))] = int(rng.integers(0, 256))
        else:
            y = ((xx // 3 + yy // 5) % 2) * int(rng.integers(20, 255)) + rng.integers(0, 3, (h, w))
        frames[k, :npx] = np.clip(y, 0, 255).astype(np.uint8).ravel()
        frames[k, npx:] = np.clip(128 + rng.integers(-20, 21, npx // 2), 0, 255).astype(np.uint8)
    floor_bytes = 8 + 2 * ((96 * 6 * 12 + 10 + 15) // 16)
    budgets = (floor_bytes + 64 + rng.integers(0, 9000, n)).astype(np.int32)
    budgets[::11] = 6000                     # plus a recurring common budget
    for codec in (0, 1, 2):
        want, want_res, rc = O.mdec_encode(codec, w, h, frames, budgets, stride=int(budgets.max()))
        if rc != 0:                          # drop frames that fit no scale (the oracle stops at the first)
            keep = [k for k in range(n) if O.mdec_encode(codec, w, h, frames[k:k + 1], int(budgets[k]))[2] == 0]
            fr, bd = frames[keep], budgets[keep]
            want, want_res, rc = O.mdec_encode(codec, w, h, fr, bd, stride=int(budgets.max()))
            assert rc == 0
        else:
            fr, bd = frames, budgets
        enc = encoder(codec, w, h, int(budgets.max()))
        d = torch.from_numpy(fr).to("cuda:0")
        d_b = torch.from_numpy(bd).to("cuda:0")
        d_out, d_res = enc.encode_frames_device(d, d_b)
        torch.cuda.synchronize()
        out, res = d_out.cpu().numpy()[:, :int(budgets.max())], d_res.cpu().numpy()
        bad = [k for k in range(len(fr)) if not (np.array_equal(out[k, :bd[k]], want[k, :bd[k]]) and np.array_equal(res[k], want_res[k]))]
        assert not bad, "codec %d: %d frames differ, first %d (budget %d, got scale %d want %d)" % (
            codec, len(bad), bad[0], bd[bad[0]], res[bad[0], 0], want_res[bad[0], 0])
        assert len(set(res[:, 0].tolist())) > 8         # the batch really spans many scales
        enc.close()


@pytest.mark.parametrize("mode", ["default", "impatient", "off"])
def test_frames_handed_on_between_workgroups(torch_cuda, monkeypatch, mode):
    """a frame whose first guess fails is handed to a queue while its workgroup still holds a fresh frame, and taken up by a
    workgroup that has run out of fresh frames (mdec_kernels.hip, top of the frame loop).  Content whose answer flips between
    neighbouring frames makes that the common case.  Three settings -- the default, waiting workgroups that give up at once
    (their slots are marked abandoned; the frame stays with its workgroup), and no queue -- must give the oracle's bytes, launch
    after launch (the queue re-arms itself), also with two contexts' kernels sharing the device."""
    import threading
    torch = torch_cuda
    from psxavenc_amd import synth
    monkeypatch.delenv("PSXHIP_MDEC_NO_RETRY_QUEUE", raising=False)
    monkeypatch.delenv("PSXHIP_MDEC_QUEUE_PATIENCE", raising=False)
    if mode == "impatient":
        monkeypatch.setenv("PSXHIP_MDEC_QUEUE_PATIENCE", "0")
    elif mode == "off":
        monkeypatch.setenv("PSXHIP_MDEC_NO_RETRY_QUEUE", "1")
    w, h, n, budget = 320, 240, 1400, 8192
    rng = np.random.default_rng(77)
    frames = torch.cat([synth.frames_device(w, h, 1, 200 * i, 200, a) for i, a in enumerate((8, 6, 3, 8, 12, 6, 8))]).cpu().numpy()
    frames = np.ascontiguousarray(frames[rng.permutation(n)])  # neighbours disagree: many wrong first guesses
    want, want_res, rc = O.mdec_encode(0, w, h, frames, budget)
    assert rc == 0
    encs = [encoder(0, w, h, budget) for _ in range(2)]
    d = torch.from_numpy(frames).to("cuda:0")
    outs = [[None] * 4, [None] * 4]

    def work(i):
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for r in range(4):
                o, q = encs[i].encode_frames_device(d, budget, stream=s)
                outs[i][r] = (o, q)
        s.synchronize()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    for i in range(2):
        for r in range(4):
            o, q = outs[i][r]
            assert_same(o.cpu().numpy()[:, :budget], q.cpu().numpy(), want, want_res, "%s ctx %d launch %d" % (mode, i, r))
    for e in encs:
        e.close()


@pytest.mark.parametrize("n", [4096, 4097, 70000])
def test_launch_sizes_either_side_of_the_queue_limits(torch_cuda, n):
    """the retry queue serves launches of more than one and at most eight frames per workgroup (512 in flight: 4096 is the last
    size with it, 4097 the first without) and keeps its counters in 20-bit fields -- 70 000 frames go through the plain 32-bit
    ticket counter.  Tiny frames with abruptly changing content (wrong first guesses everywhere); a sample against the oracle,
    and the launch's determinism (same bytes when repeated)."""
    torch = torch_cuda
    from psxavenc_amd import synth
    w, h, budget = 48, 32, 260
    parts = [synth.frames_device(w, h, 7, 1000 * i, min(1000, n - 1000 * i), (3, 30, 9, 60, 1, 18)[i % 6]) for i in range((n + 999) // 1000)]
    d = torch.cat(parts)
    perm = torch.from_numpy(np.random.default_rng(n).permutation(n)).to("cuda:0")
    d = d[perm].contiguous()
    enc = encoder(0, w, h, budget)
    o1, r1 = enc.encode_frames_device(d, budget)
    o2, r2 = enc.encode_frames_device(d, budget)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(r1, r2)
    pick = np.unique(np.concatenate([np.arange(0, n, max(1, n // 600)), np.arange(n - 40, n), np.arange(40)]))
    fr = d[torch.from_numpy(pick).to("cuda:0")].cpu().numpy()
    want, want_res, rc = O.mdec_encode(0, w, h, np.ascontiguousarray(fr), budget)
    assert rc == 0
    assert_same(o1[torch.from_numpy(pick).to("cuda:0")].cpu().numpy()[:, :budget], r1[torch.from_numpy(pick).to("cuda:0")].cpu().numpy(), want, want_res, "n=%d" % n)
    assert len(set(r1[:, 0].cpu().numpy().tolist())) > 3
    enc.close()


def test_device_fdct_matches_oracle_on_200k_blocks():
    """the DCT alone, through the entry point tools/check_fdct_vs_ffmpeg.c uses off-box (psxhip_mdec_fdct_host): same
    fdct8_pk / lane mapping / LDS transposes as the frame kernel, against orc_fdct_islow8 on flat, ramp, checkerboard,
    full-range-noise and extreme blocks"""
    import ctypes as C
    from psxavenc_amd import _lib
    L = _lib.lib()
    L.psxhip_mdec_fdct_host.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(99)
    n = 200000
    blocks = rng.integers(-128, 128, (n, 64)).astype(np.int16)
    blocks[0::7] = np.repeat(rng.integers(-128, 128, (len(blocks[0::7]), 1)), 64, axis=1).astype(np.int16)          # flat
    blocks[1::7] = np.where(rng.integers(0, 2, blocks[1::7].shape) > 0, 127, -128).astype(np.int16)                 # extremes
    yy, xx = np.mgrid[0:8, 0:8]
    blocks[2::7] = np.where(((xx ^ yy) & 1).ravel() > 0, 127, -128).astype(np.int16)                                # checkerboard
    blocks[3::7] = np.clip(rng.integers(-128, 128, (len(blocks[3::7]), 1)) + rng.integers(-4, 5, blocks[3::7].shape), -128, 127).astype(np.int16)
    blocks[4] = -128
    blocks[5] = 127
    got = np.zeros_like(blocks)
    _lib.check(L.psxhip_mdec_fdct_host(0, blocks.ctypes.data, n, got.ctypes.data))
    want = blocks.copy()
    for i in range(n):
        O.lib().orc_fdct_islow8(O.ptr(want[i], O.i16p))
    assert np.array_equal(got, want), "first differing block %d" % int(np.nonzero((got != want).any(axis=1))[0][0])
    assert got[4, 0] == 64 * -128 and got[5, 0] == 64 * 127 and not got[4, 1:].any()     # SURVEY 8(c) sanity pins


def test_large_budgets_assemble_the_frame_image_in_tiles():
    """budgets above 8 KiB: the frame image is merged and written out one 8 KiB tile at a time (so the LDS need does not
    grow with the budget twice); 640x512 (the reference CLI's largest size, args.c:410-421) up to 64 KiB, odd budgets,
    budgets that end inside a tile, streams that end exactly on a tile boundary region"""
    rng = np.random.default_rng(31)
    for (codec, w, h, cap, amps) in ((1, 640, 512, 65536, (2, 30)), (0, 320, 240, 56000, (0, 12, 40)), (2, 640, 480, 70001, (6,))):
        enc = encoder(codec, w, h, cap)
        for amp in amps:
            fr = O.synth_frames(w, h, 3, seed=61 + amp, amp=amp)
            budgets = np.array([cap, int(rng.integers(8193, cap)), 8192 * int(rng.integers(2, cap // 8192 + 1)) - int(rng.integers(0, 3))], np.int32)
            budgets = np.minimum(budgets, cap)
            want, want_res, rc = O.mdec_encode(codec, w, h, fr, budgets, stride=cap)
            if rc != 0:
                continue
            out, res = enc.encode_frames_host(fr, budgets)
            for k in range(3):
                assert np.array_equal(out[k, :budgets[k]], want[k, :budgets[k]]), (codec, w, h, amp, k, int(budgets[k]))
                assert np.array_equal(res[k], want_res[k])
        enc.close()


def test_host_path_chunks_over_two_streams_and_page_locked_buffers():
    """encode_frames_host moves a batch in chunks that alternate between two streams (kernels stay ordered), and serves
    page-locked caller buffers by DMA without the staging copy -- a strided 2-D copy when the caller's rows are wider than
    the budget.  2600 frames = three chunks; pageable and page-locked runs must both equal the oracle."""
    from psxavenc_amd import _lib
    from psxavenc_amd.mdec import MdecEncoder, register_host, unregister_host
    import ctypes as C
    w, h, n = 160, 112, 2600
    fr = O.synth_frames(w, h, n, seed=21, amp=6)
    budgets = (1400 + 8 * (np.arange(n) % 60)).astype(np.int32)
    want, want_res, rc = O.mdec_encode(1, w, h, fr, budgets, stride=int(budgets.max()))
    assert rc == 0
    enc = MdecEncoder(1, w, h, max_frame_size=int(budgets.max()))
    out, res = enc.encode_frames_host(fr, budgets)
    assert np.array_equal(out, want) and np.array_equal(res, want_res)
    # page-locked input and a page-locked output with rows wider than the largest budget
    stride = int(budgets.max()) + 52
    wide = np.full((n, stride), 0xEE, dtype=np.uint8)
    res2 = np.zeros((n, 4), dtype=np.int32)
    frp = fr.copy()
    register_host(frp)
    register_host(wide)
    try:
        rc = _lib.lib().psxhip_mdec_encode_frames_host(enc._h, frp.ctypes.data, n, budgets.ctypes.data, 0, wide.ctypes.data,
                                                       stride, res2.ctypes.data)
        _lib.check(rc)
    finally:
        unregister_host(frp)
        unregister_host(wide)
    assert np.array_equal(wide[:, :int(budgets.max())], want) and np.array_equal(res2, want_res)
    assert (wide[:, int(budgets.max()):] == 0xEE).all()          # bytes past the row's budget width are the caller's
    enc.close()


def test_single_frame_calls_follow_the_content(torch_cuda, monkeypatch):
    """The drop-in call pattern (one frame per launch, filefmt.c:637-661): the first guess of a call is the answer of the
    call before it -- also right after the content changed (the hint a launch leaves behind is its last frame's answer,
    whether or not that frame's own first guess was right).  Read from the diagnostics instantiation's per-frame record;
    every output byte against the oracle."""
    import ctypes as C
    from psxavenc_amd import _lib
    monkeypatch.setenv("PSXHIP_MDEC_STATS", "1")
    w, h, budget = 320, 240, 8192
    calm = O.synth_frames(w, h, 1, seed=5, amp=4)
    busy = O.synth_frames(w, h, 1, seed=6, amp=16)
    enc = encoder(0, w, h, budget)
    NT = 8 + 4 * 1024 + 16 + 2048
    seen = []
    for fr in (calm, calm, busy, busy, busy, calm, calm):
        want, want_res, rc = O.mdec_encode(0, w, h, fr, budget)
        assert rc == 0
        out, res = enc.encode_frames_host(fr, budget)
        assert_same(out[:, :budget], res, want, want_res, "single-frame call")
        t = (C.c_ulonglong * NT)()
        _lib.check(_lib.lib().psxhip_mdec_read_stats(enc._h, t, NT, 0))
        rec = int(t[8 + 4 * 1024 + 16])
        seen.append((rec & 0xFF, (rec >> 16) & 0xFF, (rec >> 24) & 0xFF))       # first guess, answer, passes
    enc.close()
    calm_scale, busy_scale = seen[1][1], seen[3][1]
    assert busy_scale > calm_scale + 2, seen
    for k in (1, 3, 4, 6):                   # the second call on the same content starts from the right scale ...
        assert seen[k][0] == seen[k][1], seen
    assert seen[2][0] == calm_scale and seen[5][0] == busy_scale, seen      # ... the first one after a cut from the old one


def test_scene_cuts_send_frames_back_to_the_pilot(torch_cuda, monkeypatch):
    """Scene-structured content through consecutive launches of one context (psxavenc_amd/mixed.py: cuts between noise amplitudes,
    hand-made flat / hard-edged frames): a frame that starts from a hint and is stopped at the quarter mark with a verdict FAR from
    it takes one more turn of the frame loop, from the pilot (mdec-k3.7) -- frames taken from the retry queue included.  The
    diagnostics instantiation's per-frame records show that such frames exist; every output byte against the oracle, launch
    after launch (the verdict on foreign hints travels from one launch to the next)."""
    import ctypes as C
    from psxavenc_amd import _lib, mixed
    torch = torch_cuda
    monkeypatch.setenv("PSXHIP_MDEC_STATS", "1")
    w, h, budget, n, launches = 320, 240, 8192, 768, 3
    frames = mixed.frames_host(O, w, h, 11, 100, n * launches)
    want, want_res, rc = O.mdec_encode(0, w, h, frames, budget)
    assert rc == 0
    enc = encoder(0, w, h, budget)
    NT = 8 + 4 * 1024 + 16 + 2048
    sent_back = far_first_guess = 0
    for k in range(launches):
        d = torch.from_numpy(frames[k * n:(k + 1) * n]).to("cuda:0")
        d_out, d_res = enc.encode_frames_device(d, budget)
        torch.cuda.synchronize()
        assert_same(d_out.cpu().numpy()[:, :budget], d_res.cpu().numpy(), want[k * n:(k + 1) * n], want_res[k * n:(k + 1) * n], "launch %d" % k)
        t = (C.c_ulonglong * NT)()
        _lib.check(_lib.lib().psxhip_mdec_read_stats(enc._h, t, NT, 1))
        rec = np.array(list(t)[8 + 4 * 1024 + 16:8 + 4 * 1024 + 16 + n], dtype=np.uint64)
        guess, abort, passes = (rec & 0xFF).astype(int), ((rec >> 8) & 0xFF).astype(int), ((rec >> 24) & 0xFF).astype(int)
        first = ((rec >> 32) & 0x3F).astype(int)             # the scale of the first pass of the frame's LAST attempt
        far = np.where(guess > 8, guess >> 2, 2)
        cut = (abort != 0) & (np.abs(abort - guess) >= far)
        far_first_guess += int(cut.sum())
        # a frame that was sent back: its last attempt does not start where its first guess was (the pilot chose), and it took >= 2 passes
        sent_back += int((cut & (passes >= 2) & (np.abs(first - guess) > 1)).sum())
    enc.close()
    assert far_first_guess >= 5, far_first_guess          # the sequence has a cut every ~17 frames
    assert sent_back >= 1, (sent_back, far_first_guess)


# ---------------------------------------------------------------- several devices behind one call (psxhip_multi.cpp)
@pytest.mark.parametrize("schedule,ticket", [(0, 0), (1, 0), (1, 100)])
def test_multi_device_list_equals_single_device(schedule, ticket):
    """psxhip_mdec_multi_encode_frames_host over the device list {0, 0} (two contexts, two host threads, one GPU): bytes and
    results of the single-device call, for contiguous ranges and for the host ticket queue, uniform and per-frame budgets"""
    from psxavenc_amd import multi
    w, h, n = 320, 240, 1500
    fr = np.concatenate([O.synth_frames(w, h, n // 2, seed=1, amp=4), O.synth_frames(w, h, n - n // 2, seed=2, amp=8)])
    single = encoder(0, w, h, 18144)
    m = multi.MdecMulti((0, 0), 0, w, h, 18144)
    want, want_res = single.encode_frames_host(fr, 8192)
    got, got_res = m.encode_frames_host(fr, 8192, schedule=schedule, ticket_frames=ticket)
    assert_same(got, got_res, want, want_res, "multi uniform")
    rep = m.last_report
    assert sum(r["units"] for r in rep) == n and all(r["device"] == 0 for r in rep)
    if schedule == 1 and ticket:
        assert sum(r["tickets"] for r in rep) == -(-n // ticket)
    # per-frame budgets (the STR cycle): rows are as wide as the batch's largest budget in every sub-range
    budgets = np.array([16128, 18144, 18144, 18144] * (n // 4), np.int32)
    budgets[: n // 2] = 16128                              # a sub-range whose own maximum is smaller than the batch's
    want, want_res = single.encode_frames_host(fr, budgets, out=np.full((n, 18144), 0xAA, np.uint8))
    got, got_res = m.encode_frames_host(fr, budgets, schedule=schedule, ticket_frames=ticket, out=np.full((n, 18144), 0xAA, np.uint8))
    assert_same(got, got_res, want, want_res, "multi per-frame budgets")
    ow, owr, rc = O.mdec_encode(0, w, h, fr[::97], np.ascontiguousarray(budgets[::97]), stride=18144)
    assert rc == 0
    assert_same(got[::97], got_res[::97], ow, owr, "multi vs oracle")
    single.close()
    m.close()


def test_multi_reports_no_fit_like_the_single_device_call():
    from psxavenc_amd import _lib, multi
    w, h = 64, 48
    rng = np.random.default_rng(5)
    fr = O.synth_frames(w, h, 40, seed=9, amp=4)
    fr[23] = rng.integers(0, 256, fr.shape[1], dtype=np.uint8)           # white noise: fits no scale at this budget
    m = multi.MdecMulti((0, 0, 0), 0, w, h, 400)
    with pytest.raises(_lib.PsxHipError) as e:
        m.encode_frames_host(fr, 400)
    assert e.value.code == _lib.PSXHIP_ENOFIT and "frame 23" in str(e.value)
    m.close()


# ---------------------------------------------------------------- production shapes at production batch sizes
def _check_frame_properties(out, res, version, budget):
    assert ((res[:, 0] >= 1) & (res[:, 0] <= 63)).all()
    assert (out[:, 2] == 0).all() and (out[:, 3] == 0x38).all() and (out[:, 6] == version).all() and (out[:, 7] == 0).all()
    assert np.array_equal(out[:, 4].astype(np.int32) | (out[:, 5].astype(np.int32) << 8), res[:, 0])
    assert np.array_equal(out[:, 0].astype(np.int32) | (out[:, 1].astype(np.int32) << 8), res[:, 2])
    assert (res[:, 1] <= budget).all() and (res[:, 1] % 4 == 0).all()
    col = np.arange(out.shape[1])[None, :]
    assert not (out * (col >= res[:, 1:2])).any()          # zero tail after bytes_used


@pytest.mark.parametrize("w,h,budget,n,amp,tile,what", [
    (640, 480, 8192, 1250, 4, 4096, "config 'sbs v3': one GPU's share, 8 KiB budgets -> 4 KiB image tile"),
    (640, 512, 8192, 640, 4, 2048, "the CLI's largest frame (args.c:410-421) at 8 KiB budgets -> 2 KiB image tile"),
    (640, 480, 32768, 600, 8, None, "32 KiB budgets: one 16-wavefront group per CU"),
])
def test_production_shapes_in_one_launch(torch_cuda, w, h, budget, n, amp, tile, what):
    """BS v3 at the shapes a full batch really runs in -- asserted through psxhip_mdec_query_geometry: 12 wavefronts, two
    groups per CU and the small image tiles for 8 KiB budgets (a launch of > 256 frames, so not the small-batch shape) --
    every 8th frame byte for byte against the oracle, header / tail / result properties on all, a few decoded back."""
    torch = torch_cuda
    from psxavenc_amd import synth
    from psxavenc_amd.mdec import query_geometry
    g = query_geometry(1, w, h, budget)
    assert g.fits
    if tile is not None:
        assert (g.groups_per_cu, g.wavefronts_per_group, g.image_tile_bytes) == (2, 12, tile), (what, g.groups_per_cu, g.wavefronts_per_group, g.image_tile_bytes)
    else:
        assert (g.groups_per_cu, g.wavefronts_per_group) == (1, 16)
    assert n > 256 and n > g.frames_in_flight
    enc = encoder(1, w, h, budget)
    d_frames = synth.frames_device(w, h, seed=2, first=1250 * 3, n=n, amp=amp, device=0)
    d_out, d_res = enc.encode_frames_device(d_frames, budget)
    torch.cuda.synchronize()
    out, res = d_out.cpu().numpy()[:, :budget], d_res.cpu().numpy()
    _check_frame_properties(out, res, 3, budget)
    idx = list(range(0, n, 8))
    fr = d_frames[idx].cpu().numpy()
    want, want_res, rc = O.mdec_encode(1, w, h, fr, budget)
    assert rc == 0
    assert_same(out[idx], res[idx], want, want_res, what)
    for k in idx[:3]:
        rc2, levels, scale, version, nbits = O.mdec_decode(w, h, out[k])
        assert rc2 == 0 and version == 3 and scale == res[k, 0]
        assert res[k, 1] == ((8 + 2 * ((nbits + 15) // 16) + 3) & ~3)
    enc.close()


@pytest.mark.parametrize("codec", [0, 1])
def test_batches_in_one_launch_equal_one_call_per_batch(torch_cuda, codec):
    """psxhip_mdec_encode_batches_device: the frames of several batches drawn from ONE ticket counter (no launch boundary between
    the batches) -- bytes and results must be those of one call per batch, and of the oracle.  Sizes either side of the shapes'
    limits, an empty batch, a one-frame batch, per-frame budgets in some batches, and more batches than one launch carries."""
    torch = torch_cuda
    w, h, budget = 320, 240, 8192
    sizes = [700, 0, 37, 600, 1, 300, 2, 129, 64, 5, 90]          # 11 batches: two launches (8 + 3)
    rng = np.random.default_rng(77 + codec)
    enc, ref = encoder(codec, w, h, budget), encoder(codec, w, h, budget)
    ostride = (budget + 3) & ~3
    batches, wants = [], []
    for i, n in enumerate(sizes):
        fr = O.synth_frames(w, h, n, seed=900 + i, amp=(4, 8, 16)[i % 3], first=17 * i) if n else np.zeros((0, w * h * 3 // 2), np.uint8)
        d = torch.from_numpy(fr).to("cuda:0")
        d_out = torch.full((n, ostride), 0xAB, dtype=torch.uint8, device="cuda:0")
        d_res = torch.zeros((n, 4), dtype=torch.int32, device="cuda:0")
        d_sz = None
        if i % 4 == 3 and n:
            bd = rng.integers(4096, budget + 1, n).astype(np.int32)
            d_sz = torch.from_numpy(bd).to("cuda:0")
            d_out.zero_()                 # (rows of a per-frame-budget batch are only written up to each frame's own budget)
        batches.append((d, d_out, d_res) + ((d_sz,) if d_sz is not None else ()))
        if n:
            w_out, w_res = ref.encode_frames_device(d, d_sz if d_sz is not None else budget)
            torch.cuda.synchronize()
            wants.append((w_out.cpu().numpy(), w_res.cpu().numpy(), None if d_sz is None else bd))
            want, want_res, rc = O.mdec_encode(codec, w, h, fr[:8], budget if d_sz is None else bd[:8], **({} if d_sz is None else {"stride": budget}))
            assert rc == 0 and np.array_equal(wants[-1][1][:8], want_res)
        else:
            wants.append(None)
    enc.encode_batches_device(batches, budget)
    torch.cuda.synchronize()
    for i, (b, wnt) in enumerate(zip(batches, wants)):
        if wnt is None:
            continue
        got, got_res = b[1].cpu().numpy(), b[2].cpu().numpy()
        assert np.array_equal(got_res, wnt[1]), "batch %d results" % i
        if wnt[2] is None:
            assert np.array_equal(got, wnt[0]), "batch %d bytes" % i
        else:
            for k in range(got.shape[0]):
                assert np.array_equal(got[k, :wnt[2][k]], wnt[0][k, :wnt[2][k]]), "batch %d frame %d" % (i, k)
    assert enc.watchdog() == 0
    # argument vetting: a NULL output in a non-empty batch, and a misaligned pointer
    from psxavenc_amd import _lib
    bad = list(batches[:2]) + [(batches[2][0], batches[2][1][:, 1:], batches[2][2])]
    with pytest.raises((_lib.PsxHipError, AssertionError)):
        enc.encode_batches_device(bad, budget)
    enc.close()
    ref.close()


def test_two_launch_lanes_overlap_launches_of_one_context_same_bytes(torch_cuda):
    """psxhip_mdec_set_lanes(2): the in-order caller of filefmt.c:641-647 with double buffering -- inputs are produced ON the
    caller's stream right before each call (dependent launches), outputs are read on the stream one call later (results lag one
    call), the last one after psxhip_mdec_fence.  Every byte equals the one-lane encode; interleaved with independent launches."""
    torch = torch_cuda
    w, h, budget, n = 320, 240, 8192, 700
    K = 9
    src = [torch.from_numpy(O.synth_frames(w, h, n, seed=300 + k, amp=(4, 8)[k & 1], first=k * n)).to("cuda:0") for k in range(K)]
    one = encoder(0, w, h, budget)
    wants = []
    for k in range(K):
        o, r = one.encode_frames_device(src[k], budget)
        torch.cuda.synchronize()
        wants.append((o.cpu().numpy(), r.cpu().numpy()))
    want0, want0_res, rc = O.mdec_encode(0, w, h, src[0][:16].cpu().numpy(), budget)
    assert rc == 0 and np.array_equal(wants[0][0][:16, :budget], want0)
    one.close()
    enc = encoder(0, w, h, budget)
    enc.set_lanes(2)
    ostride = (budget + 3) & ~3
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d_in = [torch.empty_like(src[0]) for _ in range(2)]
        d_out = [torch.zeros((n, ostride), dtype=torch.uint8, device="cuda:0") for _ in range(3)]
        d_res = [torch.zeros((n, 4), dtype=torch.int32, device="cuda:0") for _ in range(3)]
        snaps = []
        for rep in range(3):
            for k in range(K):
                d_in[k & 1].copy_(src[k], non_blocking=True)                       # produced on the stream: the launch must wait for it
                enc.encode_frames_device(d_in[k & 1], budget, d_out=d_out[k % 3], d_results=d_res[k % 3])
                if k >= 1:                                                         # launch k-1 is ordered into the stream by call k
                    snaps.append((k - 1, d_out[(k - 1) % 3].clone(), d_res[(k - 1) % 3].clone()))
                if k == 4:                                                         # an independent launch of the same context in between
                    enc.encode_frames_device(src[0], budget, d_out=d_out[(k + 1) % 3], d_results=d_res[(k + 1) % 3])
                    enc.fence()
                    snaps.append((0, d_out[(k + 1) % 3].clone(), d_res[(k + 1) % 3].clone()))
            enc.fence()
            snaps.append((K - 1, d_out[(K - 1) % 3].clone(), d_res[(K - 1) % 3].clone()))
    s.synchronize()
    for k, o, r in snaps:
        assert np.array_equal(r.cpu().numpy(), wants[k][1]), "launch %d results" % k
        assert np.array_equal(o.cpu().numpy(), wants[k][0]), "launch %d bytes" % k
    assert enc.watchdog() == 0
    enc.set_lanes(1)                 # back to plain stream order
    o, r = enc.encode_frames_device(src[3], budget)
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy(), wants[3][0])
    enc.close()


@pytest.mark.gpu
def test_host_path_after_unfenced_lane_launches(torch_cuda):
    """two lanes, device launches left outstanding (no fence), then the host-buffer entry points of the same context -- the
    batch and the single-frame call (mdec.c:580): they wait for the lanes themselves; every byte as with one lane"""
    torch = torch_cuda
    w, h, budget, n = 320, 240, 8192, 900
    frames = O.synth_frames(w, h, n, seed=77, amp=4)
    d_frames = torch.from_numpy(frames).to("cuda:0")
    one = encoder(0, w, h, budget)
    want_o, want_r = one.encode_frames_device(d_frames, budget)
    torch.cuda.synchronize()
    want_o, want_r = want_o.cpu().numpy(), want_r.cpu().numpy()
    one.close()
    enc = encoder(0, w, h, budget)
    enc.set_lanes(2)
    for rep in range(4):
        d1 = enc.encode_frames_device(d_frames, budget)
        d2 = enc.encode_frames_device(d_frames, budget)
        if rep & 1:
            ho, hr = enc.encode_frames_host(frames[:300], budget)
            assert np.array_equal(hr, want_r[:300]) and np.array_equal(ho, want_o[:300, :budget])
        else:
            enc.frame_max_size = budget
            bs = enc.encode_frame_bs(frames[5])
            assert np.array_equal(bs, want_o[5, :budget]) and enc.quant_scale == want_r[5, 0]
        enc.fence()
        torch.cuda.synchronize()
        for o, r in (d1, d2):
            assert np.array_equal(r.cpu().numpy(), want_r) and np.array_equal(o.cpu().numpy(), want_o)
    assert enc.watchdog() == 0
    enc.close()


def test_host_path_after_fenced_but_still_running_lane_launches(torch_cuda):
    """psxhip_mdec_fence clears a lane's "pending" flag while the launch may still be running -- the flag means
    "nobody ordered behind it yet", not "in flight".  Sequence: two lanes; device launches of a LARGE batch (long enough to be
    in flight when the host call arrives); fence; then, with no stream synchronise, the host-buffer entry points of the same
    context (they use both lanes' ticket counters on their own streams).  Also: two device calls, then a host call (lane 0's flag
    was cleared by the second call).  Every byte as with one lane."""
    torch = torch_cuda
    w, h, budget, n = 320, 240, 8192, 4000
    frames = O.synth_frames(w, h, 400, seed=78, amp=8)
    big = np.concatenate([frames] * (n // 400), axis=0)
    d_frames = torch.from_numpy(big).to("cuda:0")
    one = encoder(0, w, h, budget)
    want_o, want_r = one.encode_frames_device(d_frames[:400], budget)
    torch.cuda.synchronize()
    want_o, want_r = want_o.cpu().numpy(), want_r.cpu().numpy()
    one.close()
    enc = encoder(0, w, h, budget)
    enc.set_lanes(2)
    d_out = [torch.zeros((n, budget), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    d_res = [torch.zeros((n, 4), dtype=torch.int32, device="cuda:0") for _ in range(2)]
    for rep in range(6):
        for k in range(2):
            d_res[k].zero_()
        enc.encode_frames_device(d_frames, budget, d_out=d_out[0], d_results=d_res[0])
        if rep % 3 != 2:
            enc.encode_frames_device(d_frames, budget, d_out=d_out[1], d_results=d_res[1])
        if rep % 3 == 0:
            enc.fence()                      # flags down, launches still running
        if rep & 1:
            ho, hr = enc.encode_frames_host(frames[:300], budget)
            assert np.array_equal(hr, want_r[:300]) and np.array_equal(ho, want_o[:300, :budget])
        else:
            enc.frame_max_size = budget
            bs = enc.encode_frame_bs(frames[7])
            assert np.array_equal(bs, want_o[7, :budget]) and enc.quant_scale == want_r[7, 0]
        enc.fence()
        torch.cuda.synchronize()
        for k in range(2 if rep % 3 != 2 else 1):
            r = d_res[k].cpu().numpy().reshape(n // 400, 400, 4)
            o = d_out[k].cpu().numpy().reshape(n // 400, 400, budget)
            for j in range(n // 400):
                assert np.array_equal(r[j], want_r) and np.array_equal(o[j], want_o), (rep, k, j)
    assert enc.watchdog() == 0
    enc.close()
