"""One frame across many workgroups (csrc/mdec_split.inc): launches of a few frames -- encode_frame_bs, one frame per call
(filefmt.c:641-647), most of all -- take a kernel of their own.  Every byte against the CPU oracle and against the frame kernel
(the same library with PSXHIP_MDEC_SPLIT_MAX=0): codecs, sizes, even / odd / per-frame budgets, content from flat to escape-heavy,
frames that fit only above scale 16 (a second round of scales), frames that fit nowhere.  Bar: bit-exact."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _encoder(codec, w, h, budget, split=True, m=None):
    from psxavenc_amd.mdec import MdecEncoder
    old = os.environ.get("PSXHIP_MDEC_SPLIT_MAX")
    if split:
        os.environ.pop("PSXHIP_MDEC_SPLIT_MAX", None)                     # the default: launches of up to 12 frames
    else:
        os.environ["PSXHIP_MDEC_SPLIT_MAX"] = "0"                         # (read when the context is created)
    try:
        return MdecEncoder(codec, w, h, max_frame_size=budget, device=0)
    finally:
        if old is None:
            os.environ.pop("PSXHIP_MDEC_SPLIT_MAX", None)
        else:
            os.environ["PSXHIP_MDEC_SPLIT_MAX"] = old


def _check(codec, w, h, frames, budgets, stride, tag):
    want, want_res, rc = O.mdec_encode(codec, w, h, frames, budgets, stride=stride)
    assert rc == 0, tag
    enc = _encoder(codec, w, h, stride)
    ref = _encoder(codec, w, h, stride, split=False)
    for e, name in ((enc, "split"), (ref, "frame kernel")):
        out, res = e.encode_frames_host(frames, budgets)
        assert np.array_equal(res, want_res), (tag, name, res[:4], want_res[:4])
        bad = np.nonzero((out != want).any(axis=1))[0]
        assert bad.size == 0, "%s (%s): frames %s differ, first at byte %d" % (tag, name, bad[:6].tolist(),
                                                                              int(np.nonzero(out[bad[0]] != want[bad[0]])[0][0]))
    enc.close()
    ref.close()


@pytest.mark.parametrize("codec", [0, 1, 2])
@pytest.mark.parametrize("w,h", [(320, 240), (640, 480), (16, 16), (48, 32), (160, 112), (640, 512)])
def test_one_to_twelve_frames_vs_oracle_and_frame_kernel(codec, w, h):
    budget = 32768 if w * h >= 640 * 480 else (8192 if w * h >= 160 * 112 else 2048)
    for n, amp, seed in ((1, 4, 1), (1, 8, 2), (2, 6, 3), (3, 2, 4), (5, 9, 5), (8, 5, 6), (12, 7, 7)):
        fr = O.synth_frames(w, h, n, seed=seed, amp=amp)
        _check(codec, w, h, fr, budget, budget, "c%d %dx%d n=%d amp=%d" % (codec, w, h, n, amp))


@pytest.mark.parametrize("codec", [0, 1, 2])
def test_budgets_even_odd_per_frame(codec):
    w, h = 320, 240
    fr = O.synth_frames(w, h, 6, seed=11, amp=6)
    for b in (8192, 8191, 4097, 16128, 18144, 30001):
        _check(codec, w, h, fr[:3], b, b, "uniform %d" % b)
    budgets = np.array([16128, 18144, 8191, 4097, 30000, 6002], np.int32)
    _check(codec, w, h, fr, budgets, 30000, "per-frame budgets")


@pytest.mark.parametrize("codec", [0, 1, 2])
def test_special_frames(codec):
    """flat fields (v3 DC ties), checkerboards (DC deltas near +-255, the v3dc wrap), hard edges (escapes), DC staircases"""
    sys.path.insert(0, os.path.join(O.ROOT, "tests", "golden"))
    from make_mdec_golden import special_frames
    for (w, h, budget) in ((48, 32, 4096), (320, 240, 30000), (320, 240, 9000)):
        fr = special_frames(w, h)
        enc = _encoder(codec, w, h, budget)
        from psxavenc_amd import _lib
        for k in range(fr.shape[0]):
            want, want_res, rc = O.mdec_encode(codec, w, h, fr[k:k + 1], budget)
            if rc == 0:
                out, res = enc.encode_frames_host(fr[k:k + 1], budget)
                assert np.array_equal(res, want_res) and np.array_equal(out, want), (codec, w, h, budget, k)
            else:
                with pytest.raises(_lib.PsxHipError) as e:
                    enc.encode_frames_host(fr[k:k + 1], budget)
                assert e.value.code == _lib.PSXHIP_ENOFIT
        enc.close()


@pytest.mark.parametrize("codec", [0, 1])
def test_answers_above_scale_16_take_further_rounds(codec):
    """noise that fits only at scales 17..63: the second, third, fourth round of sixteen scales"""
    w, h = 320, 240
    seen = set()
    for amp, budget in ((40, 8192), (60, 6000), (90, 6000), (120, 5000), (30, 4096)):
        fr = O.synth_frames(w, h, 2, seed=amp, amp=amp)
        want, want_res, rc = O.mdec_encode(codec, w, h, fr, budget)
        if rc != 0:
            continue
        seen.update(int(s) for s in want_res[:, 0])
        _check(codec, w, h, fr, budget, budget, "amp %d budget %d" % (amp, budget))
    assert max(seen) > 32 and any(16 < s <= 32 for s in seen), seen


def test_device_entry_point_small_launches_in_a_row():
    """psxhip_mdec_encode_frames_device with 1..8 frames per launch (a hinted one-frame call in between: another workspace), back to back on one stream and on two launch lanes: the
    workspace a launch leaves behind is the next one's"""
    import torch
    w, h, budget = 320, 240, 8192
    fr = O.synth_frames(w, h, 40, seed=77, amp=7)
    want, want_res, rc = O.mdec_encode(0, w, h, fr, budget)
    assert rc == 0
    d = torch.from_numpy(fr).to("cuda:0")
    for lanes in (1, 2):
        enc = _encoder(0, w, h, budget)
        if lanes > 1:
            enc.set_lanes(2)
        outs, at = [], 0
        for rep in range(3):
            at = 0
            outs = []
            for n in (1, 2, 3, 8, 1, 5, 7, 4, 1, 8):          # (40 frames)
                o, r = enc.encode_frames_device(d[at:at + n], budget)
                outs.append((at, n, o, r))
                at += n
        enc.fence()
        torch.cuda.synchronize()
        for at, n, o, r in outs:
            assert np.array_equal(r.cpu().numpy(), want_res[at:at + n]), (lanes, at, n)
            assert np.array_equal(o.cpu().numpy()[:, :budget], want[at:at + n]), (lanes, at, n)
        enc.close()


def test_segment_sizes_all_agree():
    """M = 1, 2, 4, 8, 16 macroblocks per workgroup (PSXHIP_MDEC_SPLIT_M, experiments): the same bytes -- in child processes, the
    switch is read once per process"""
    import subprocess
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import oracle_lib as O\n"
            "from psxavenc_amd.mdec import MdecEncoder\n"
            "ok = True\n"
            "for codec, w, h, b in ((0, 320, 240, 8192), (1, 320, 240, 8192), (2, 160, 112, 5000)):\n"
            "    fr = O.synth_frames(w, h, 3, seed=5, amp=6)\n"
            "    want, wr, rc = O.mdec_encode(codec, w, h, fr, b)\n"
            "    e = MdecEncoder(codec, w, h, max_frame_size=b, device=0)\n"
            "    out, res = e.encode_frames_host(fr, b)\n"
            "    ok = ok and rc == 0 and np.array_equal(out, want) and np.array_equal(res, wr)\n"
            "print('OK' if ok else 'DIFF')\n") % (O.ROOT, os.path.join(O.ROOT, "tests"))
    for m in (1, 2, 4, 8, 16):
        env = dict(os.environ, PSXHIP_MDEC_SPLIT_M=str(m))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (m, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("codec", [0, 1])
def test_one_frame_calls_across_scene_cuts_follow_the_hint_and_stay_exact(codec):
    """The reference's loop, one frame per call (filefmt.c:641-647), over cuts between quiet and noisy scenes: a call starts from the
    answer of the call before it (eight scales in its first round when that was <= 6; streams built at the hinted scale while the sums
    travel), so an answer that jumps from 2 to 20 and back takes the further rounds / the fallback emit -- same bytes either way."""
    w, h, budget = 320, 240, 8192
    amps = [2, 2, 30, 30, 3, 40, 2, 12, 20, 1, 50, 6, 18, 6]
    frames = np.concatenate([O.synth_frames(w, h, 1, seed=90 + i, amp=a) for i, a in enumerate(amps)])
    want, want_res, rc = O.mdec_encode(codec, w, h, frames, budget)
    assert rc == 0
    scales = want_res[:, 0].tolist()
    assert min(scales) <= 3 and max(scales) > 16 and any(8 < s <= 16 for s in scales), scales
    enc = _encoder(codec, w, h, budget)
    for rep in range(2):
        for k in range(len(amps)):
            out, res = enc.encode_frames_host(frames[k:k + 1], budget)
            assert np.array_equal(res[0], want_res[k]), (rep, k, res[0], want_res[k])
            assert np.array_equal(out[0], want[k]), (rep, k)
    enc.close()


@pytest.mark.parametrize("switch", ["PSXHIP_NO_BAR_WRITE", "PSXHIP_MDEC_SPLIT_MAX", "PSXHIP_NO_HDP_FLUSH", "PSXHIP_NO_PERCALL_PATH"])
def test_one_frame_calls_on_the_fallback_paths(switch):
    """the paths a system without a large BAR (frame through the page-locked block and a copy kernel), a context with the split kernel
    off (every launch to the frame kernel), and the batched path (copies + chunks) take for one frame per call: the same bytes"""
    from psxavenc_amd.mdec import MdecEncoder
    w, h, budget = 320, 240, 8192
    frames = O.synth_frames(w, h, 6, seed=55, amp=9)
    want, want_res, rc = O.mdec_encode(1, w, h, frames, budget)
    assert rc == 0
    old = os.environ.get(switch)
    os.environ[switch] = "0" if switch == "PSXHIP_MDEC_SPLIT_MAX" else "1"          # (read when the context / its call block is created)
    try:
        enc = MdecEncoder(1, w, h, max_frame_size=budget, device=0)
        for k in range(6):
            out, res = enc.encode_frames_host(frames[k:k + 1], budget)
            assert np.array_equal(res[0], want_res[k]) and np.array_equal(out[0], want[k]), (switch, k)
        enc.close()
    finally:
        if old is None:
            os.environ.pop(switch, None)
        else:
            os.environ[switch] = old
