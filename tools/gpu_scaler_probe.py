"""Scaler kernel probe: one geometry, frames/s (pictures resident in HBM).
usage: gpu_scaler_probe.py fmt sw sh dw dh [n [full_range]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from psxavenc_amd.frontend import Scaler

fmt, sw, sh, dw, dh = [int(x) for x in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 1000
full = bool(int(sys.argv[7])) if len(sys.argv) > 7 else True
sc = Scaler(fmt, sw, sh, dw, dh, src_full_range=full)
d_src = torch.randint(0, 256, (n, sc.source_bytes), dtype=torch.uint8, device="cuda:0")
d_frames = torch.empty((n, sc.frame_bytes), dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    sc.convert_device(d_src, d_frames)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    sc.convert_device(d_src, d_frames)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print("fmt %d %s %dx%d -> %dx%d n %d vsegs %s: %.4f ms  %.0f frames/s  %.1f GB/s" % (
    fmt, "full" if full else "limited", sw, sh, dw, dh, n, os.environ.get("PSXHIP_SCALER_VSEGS", "auto"), ms, n / ms * 1e3, (sc.source_bytes + sc.frame_bytes) * n / ms / 1e6))
