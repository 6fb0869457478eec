#!/usr/bin/env python3
"""Per-launch times of 1000-frame launches, one at a time (HIP event pair and a synchronise around every launch), cycling over the four
batches of a content class: is the in-order figure of a class one slow launch in N, or every launch a little slower?
usage: python tools/gpu_launch_times.py [kind ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpu_r05_diag as D
from psxavenc_amd.mdec import MdecEncoder

for kind in (sys.argv[1:] or ["a4", "a8"]):
    bb = D.batches(kind)
    N, B = D.N, D.BUDGET
    outs = [(torch.zeros((N, B), dtype=torch.uint8, device="cuda"), torch.zeros((N, 4), dtype=torch.int32, device="cuda")) for _ in range(4)]
    enc = MdecEncoder(D.CODEC, D.W, D.H, max_frame_size=B, device=0)
    ts = []
    for k in range(240):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        enc.encode_frames_device(bb[k % 4], B, d_out=outs[k % 4][0], d_results=outs[k % 4][1])
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    enc.close()
    t = np.array(ts[40:]) * 1e3
    print(kind, "us per launch: min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f mean %.0f" % (t.min(), np.percentile(t, 10), np.percentile(t, 50), np.percentile(t, 90), t.max(), t.mean()))
    by = [np.round(t[i::4].mean()) for i in range(4)]
    print("   mean per batch of the cycle:", by, " histogram (10 us bins from %d):" % (int(t.min()) // 10 * 10), np.bincount(((t - int(t.min()) // 10 * 10) // 10).astype(int)).tolist())
