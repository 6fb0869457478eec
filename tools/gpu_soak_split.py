"""Soak of the split kernel (one frame across many workgroups, csrc/mdec_split.inc): launches of 1..12 frames -- the one-frame call
through the host entry point (BAR write + flag) and device-resident launches of n frames on one or two lanes -- random geometry,
codec, per-frame budgets (even and odd), content from flat to heavy noise (answers 1..60, several rounds of scales), every byte and
result field against the oracle.   usage: gpu_soak_split.py [rounds [seed]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from psxavenc_amd.mdec import MdecEncoder

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
GEOS = [(320, 240), (320, 240), (160, 112), (640, 480), (48, 32), (16, 16), (640, 512), (336, 272)]
total = bad = calls = 0
hist = np.zeros(64, np.int64)
t0 = time.time()
for rnd in range(rounds):
    codec = int(rng.integers(0, 3))
    w, h = GEOS[int(rng.integers(0, len(GEOS)))]
    nmb = (w // 16) * (h // 16)
    n = int(rng.integers(1, 13))
    amp = int(rng.integers(0, 60)) if rng.random() < 0.8 else int(rng.integers(60, 128))
    frames = O.synth_frames(w, h, n, seed=int(rng.integers(1, 1 << 30)), amp=amp, first=int(rng.integers(0, 100000)))
    lo = 8 + 2 * ((nmb * 6 * 12 + 10 + 15) // 16)
    top = lo + 64 + int(rng.integers(200, 1 + max(201, min(60000, nmb * 160))))
    budgets = rng.integers(lo + 32, top, n).astype(np.int32)
    if rng.random() < 0.3:
        budgets[:] = int(budgets[0])
    keep = [k for k in range(n) if O.mdec_encode(codec, w, h, frames[k:k + 1], int(budgets[k]))[2] == 0]
    if not keep:
        continue
    frames, budgets = frames[keep], budgets[keep]
    n = len(keep)
    bmax = int(budgets.max())
    want, want_res, rc = O.mdec_encode(codec, w, h, frames, budgets, stride=bmax)
    assert rc == 0
    enc = MdecEncoder(codec, w, h, max_frame_size=bmax, device=0)
    mode = int(rng.integers(0, 3))
    ok = True
    if mode == 0:                      # the reference's pattern: one frame per call, twice over (the second call starts from a hint)
        for rep in range(2):
            for k in range(n):
                out, res = enc.encode_frames_host(frames[k:k + 1], int(budgets[k]))
                ok = ok and np.array_equal(out[0, :budgets[k]], want[k, :budgets[k]]) and np.array_equal(res[0], want_res[k])
                calls += 1
    else:                              # device-resident launches of n frames, one lane or two, three in a row
        if mode == 2:
            enc.set_lanes(2)
        d_fr, d_b = torch.from_numpy(frames).to("cuda:0"), torch.from_numpy(budgets).to("cuda:0")
        outs = [enc.encode_frames_device(d_fr, d_b) for _ in range(3)]
        enc.fence()
        torch.cuda.synchronize()
        for d_out, d_res in outs:
            out, res = d_out.cpu().numpy()[:, :bmax], d_res.cpu().numpy()
            for k in range(n):
                ok = ok and np.array_equal(out[k, :budgets[k]], want[k, :budgets[k]])
            ok = ok and np.array_equal(res, want_res)
        calls += 3
    enc.close()
    total += n
    bad += 0 if ok else 1
    np.add.at(hist, np.clip(want_res[:, 0], 0, 63), 1)
    if not ok or rnd % 20 == 0:
        print("round %3d codec %d %dx%d n %2d amp %3d mode %d scales %2d..%2d %s" % (rnd, codec, w, h, n, amp, mode, want_res[:, 0].min(), want_res[:, 0].max(), "ok" if ok else "MISMATCH"), flush=True)
print("answers by round of scales: 1-8 %d, 9-16 %d, 17-32 %d, 33-48 %d, 49-63 %d" % (hist[1:9].sum(), hist[9:17].sum(), hist[17:33].sum(), hist[33:49].sum(), hist[49:].sum()))
print("split soak: %d rounds, %d frames, %d launches / calls, %d mismatching rounds, %.0f s" % (rounds, total, calls, bad, time.time() - t0))
