"""frames per launch -> frames/s (MDEC sbs v2 320x240, budget 8192, noise +-4): how much of a launch is ramp-up / tail"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psxavenc_amd import synth
from psxavenc_amd.mdec import MdecEncoder
w, h, budget = 320, 240, 8192
enc = MdecEncoder(0, w, h, max_frame_size=budget, device=0)
for n in (1, 64, 256, 300, 512, 768, 1000, 1024, 1536, 2048, 4096, 8192):
    d = synth.frames_device(w, h, 1, 0, n, 4, device=0)
    out = torch.zeros((n, budget), dtype=torch.uint8, device="cuda")
    res = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    for _ in range(5):
        enc.encode_frames_device(d, budget, d_out=out, d_results=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    e0.record()
    for _ in range(reps):
        enc.encode_frames_device(d, budget, d_out=out, d_results=res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%5d frames/launch: %.4f ms  %9.0f frames/s  (%.1f us per 512-frame round)" % (n, ms, n / ms * 1e3, ms * 1e3 / max(1.0, n / 512.0)))
