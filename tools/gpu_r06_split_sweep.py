#!/usr/bin/env python3
"""round 6: launches of n frames, device-resident -- the split kernel (one frame across many workgroups) against the frame kernel.
Where the split kernel stops paying sets PSXHIP_MDEC_SPLIT_MAX's default."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from psxavenc_amd import synth  # noqa: E402
from psxavenc_amd.mdec import MdecEncoder  # noqa: E402


def rate(codec, w, h, budget, n, split_max, amp=4, reps=300):
    os.environ["PSXHIP_MDEC_SPLIT_MAX"] = str(split_max)
    enc = MdecEncoder(codec, w, h, max_frame_size=budget, device=0)
    fr = synth.frames_device(w, h, 5, 0, n, amp, device=0)
    out = torch.zeros((n, (budget + 3) & ~3), dtype=torch.uint8, device="cuda:0")
    res = torch.zeros((n, 4), dtype=torch.int32, device="cuda:0")
    for _ in range(10):
        enc.encode_frames_device(fr, budget, d_out=out, d_results=res)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            enc.encode_frames_device(fr, budget, d_out=out, d_results=res)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    enc.close()
    return best, res[:, 0].tolist()[:2]


rows = []
for (codec, w, h, budget) in ((0, 320, 240, 8192), (1, 640, 480, 32768)):
    for n in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
        s, sc = rate(codec, w, h, budget, n, 64)
        f, _ = rate(codec, w, h, budget, n, 0)
        rows.append({"codec": codec, "w": w, "h": h, "budget": budget, "frames": n, "split_us": round(s, 2), "frame_kernel_us": round(f, 2), "scales": sc})
        print(rows[-1], flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r06_split_sweep.json"), "w"), indent=1)
