"""Throughput on a batch whose frames differ in cost (final quant scale), to exercise dynamic frame assignment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from psxavenc_amd import synth
from psxavenc_amd.mdec import MdecEncoder
w, h, budget = 320, 240, 8192
enc = MdecEncoder(0, w, h, max_frame_size=budget)
for name, amps in (("uniform +-4", [4] * 1000), ("first half +-4, second half +-24", [4] * 500 + [24] * 500),
                   ("every 8th frame +-30", [30 if i % 8 == 0 else 4 for i in range(1000)])):
    d = torch.empty((1000, w * h * 3 // 2), dtype=torch.uint8, device="cuda:0")
    a = np.array(amps)
    for amp in np.unique(a):
        idx = np.nonzero(a == amp)[0]
        # contiguous runs share one generator call
        runs = np.split(idx, np.nonzero(np.diff(idx) != 1)[0] + 1)
        for r in runs:
            synth.frames_device(w, h, 1, int(r[0]), len(r), int(amp), out=d[int(r[0]):int(r[-1]) + 1])
    for _ in range(3): out, res = enc.encode_frames_device(d, budget)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out, res = enc.encode_frames_device(d, budget)
    e1.record(); torch.cuda.synchronize()
    sc = res.cpu().numpy()[:, 0]
    print("%-36s %.4f ms/launch  %.2f M frames/s  scales min %d max %d mean %.1f" % (name, e0.elapsed_time(e1) / 10, 10 / e0.elapsed_time(e1), sc.min(), sc.max(), sc.mean()))
