"""Soak of the launch lanes and batch lists: one context, one stream, two lanes; K batches of seeded frames with per-frame budgets
are launched back to back without a fence in between (a launch's head overlaps its predecessor's tail; frames are handed on
between workgroups through the lanes' own retry queues), every other round as ONE call with a list of batches
(psxhip_mdec_encode_batches_device).  Every byte of every batch against the oracle.
usage: gpu_soak_lanes.py [rounds [seed [frames_per_batch]]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from psxavenc_amd.mdec import MdecEncoder

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 700
total = bad = 0
t0 = time.time()
for rnd in range(rounds):
    codec = rnd % 3
    w, h = [(320, 240), (160, 112), (320, 240), (640, 480)][rnd % 4]
    n = N if w <= 320 else max(40, N // 8)
    K = int(rng.integers(3, 7))
    lo = 8 + 2 * (((w // 16) * (h // 16) * 6 * 12 + 10 + 15) // 16)
    cap = lo + 300 + int(rng.integers(2000, 30000))
    batches = []
    for k in range(K):
        amp = int(rng.integers(0, 30))
        nk = int(rng.integers(max(1, n // 3), n + 1))
        frames = O.synth_frames(w, h, nk, seed=int(rng.integers(1, 1 << 30)), amp=amp, first=int(rng.integers(0, 100000)))
        budgets = rng.integers(lo + 300, cap + 1, nk).astype(np.int32)
        want, want_res, rc = O.mdec_encode(codec, w, h, frames, budgets, stride=cap)
        if rc != 0:
            keep = [i for i in range(nk) if O.mdec_encode(codec, w, h, frames[i:i + 1], int(budgets[i]))[2] == 0]
            frames, budgets = frames[keep], budgets[keep]
            want, want_res, rc = O.mdec_encode(codec, w, h, frames, budgets, stride=cap)
            nk = len(keep)
        batches.append((frames, budgets, want, want_res))
    enc = MdecEncoder(codec, w, h, max_frame_size=cap)
    ostride = (cap + 3) & ~3
    d_in = [torch.from_numpy(b[0]).to("cuda:0") for b in batches]
    d_bud = [torch.from_numpy(b[1]).to("cuda:0") for b in batches]
    d_out = [torch.zeros((len(b[1]), ostride), dtype=torch.uint8, device="cuda:0") for b in batches]
    d_res = [torch.zeros((len(b[1]), 4), dtype=torch.int32, device="cuda:0") for b in batches]
    as_list = bool(rnd & 1) and K <= 8
    if as_list:
        enc.encode_batches_device([(d_in[k], d_out[k], d_res[k], d_bud[k]) for k in range(K)], cap)
    else:
        enc.set_lanes(2)
        for rep in range(2):                      # twice over the same buffers: the second sweep starts on warm hints
            for k in range(K):
                enc.encode_frames_device(d_in[k], d_bud[k], d_out=d_out[k], d_results=d_res[k])
        enc.fence()
    torch.cuda.synchronize()
    ok = enc.watchdog() == 0
    for k, (frames, budgets, want, want_res) in enumerate(batches):
        out, res = d_out[k].cpu().numpy()[:, :cap], d_res[k].cpu().numpy()
        want = want[:, :cap].copy()
        for i in range(len(budgets)):
            out[i, budgets[i]:] = 0; want[i, budgets[i]:] = 0
        ok = ok and np.array_equal(out, want) and np.array_equal(res, want_res)
        total += len(budgets)
    bad += 0 if ok else 1
    print("round %3d codec %d %dx%d batches %d (%s) frames %s  %s" % (rnd, codec, w, h, K, "one call" if as_list else "two lanes", [len(b[1]) for b in batches], "ok" if ok else "MISMATCH"), flush=True)
    enc.close()
print("lanes / batch-list soak: %d frames, %d mismatching rounds, %.0f s" % (total, bad, time.time() - t0))
