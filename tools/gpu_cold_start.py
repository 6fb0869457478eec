"""first launch of a fresh context (no answer of a previous batch to start from: pilot on every group's first frame) vs steady state"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from psxavenc_amd import synth
from psxavenc_amd.mdec import MdecEncoder
w, h, budget, n = 320, 240, 8192, 1000
for amp in (4, 8):
    d = synth.frames_device(w, h, 1, 0, n, amp, device=0)
    out = torch.zeros((n, budget), dtype=torch.uint8, device="cuda")
    res = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    warm = MdecEncoder(0, w, h, max_frame_size=budget, device=0)
    for _ in range(5):
        warm.encode_frames_device(d, budget, d_out=out, d_results=res)
    torch.cuda.synchronize()
    cold = []
    for _ in range(12):
        enc = MdecEncoder(0, w, h, max_frame_size=budget, device=0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); enc.encode_frames_device(d, budget, d_out=out, d_results=res); e1.record()
        torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1))
        enc.close()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        warm.encode_frames_device(d, budget, d_out=out, d_results=res)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    print("noise +-%d: cold first launch median %.4f ms (%.2f M frames/s), steady %.4f ms (%.2f M frames/s)" % (amp, np.median(cold), n / np.median(cold) / 1e3, ms, n / ms / 1e3))
