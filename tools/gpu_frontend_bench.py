"""Throughput of the colour-conversion / scaling front-end (psxhip_scaler_convert_device), pictures and frames resident in
HBM, and of the chain pictures -> NV21 -> BS frames on one stream."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from psxavenc_amd.frontend import Scaler
from psxavenc_amd.mdec import MdecEncoder

rows = []
for fmt, name, sw, sh, dw, dh in ((0, "rgb24", 640, 480, 320, 240), (1, "yuv420p", 640, 480, 320, 240), (1, "yuv420p", 1280, 720, 320, 176),
                                  (0, "rgb24", 320, 240, 320, 240), (1, "yuv420p", 720, 576, 640, 480)):
    n = 1000
    sc = Scaler(fmt, sw, sh, dw, dh)
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
    alg = (sc.source_bytes + sc.frame_bytes) * n
    rows.append({"case": "%s %dx%d -> %dx%d" % (name, sw, sh, dw, dh), "frames_per_sec": round(n / ms * 1e3, 1), "ms_per_1000": round(ms, 4),
                 "algorithmic_gbs": round(alg / ms / 1e6, 1), "frac_of_8tbs": round(alg / ms / 1e6 / 8000, 4)})
    print(rows[-1], flush=True)
    if (dw, dh) == (320, 240) and sw == 640 and fmt == 0:
        enc = MdecEncoder(0, dw, dh, max_frame_size=8192)
        d_out = torch.zeros((n, 8192), dtype=torch.uint8, device="cuda:0"); d_res = torch.zeros((n, 4), dtype=torch.int32, device="cuda:0")
        for _ in range(3):
            sc.convert_device(d_src, d_frames); enc.encode_frames_device(d_frames, 8192, d_out=d_out, d_results=d_res)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            sc.convert_device(d_src, d_frames); enc.encode_frames_device(d_frames, 8192, d_out=d_out, d_results=d_res)
        b.record(); torch.cuda.synchronize()
        ms2 = a.elapsed_time(b) / 20
        rows.append({"case": "chain rgb24 640x480 -> NV21 320x240 -> BS v2 (random pictures)", "frames_per_sec": round(n / ms2 * 1e3, 1), "ms_per_1000": round(ms2, 4)})
        print(rows[-1], flush=True)
        enc.close()
    sc.close()
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r03_frontend_bench.json"), "w"), indent=1)
