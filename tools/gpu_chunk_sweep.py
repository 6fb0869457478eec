"""host-path chunk size (PSXHIP_MDEC_CHUNK) -> PCIe-inclusive frames/s at 1000 and 8000 frames per call, pageable caller buffers"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from psxavenc_amd.mdec import MdecEncoder
w, h, budget = 320, 240, 8192
fr = O.synth_frames(w, h, 1000, seed=1, amp=4)
fr8 = np.concatenate([fr] * 8)
for chunk in [int(x) for x in sys.argv[1:]] or [1024]:
    os.environ["PSXHIP_MDEC_CHUNK"] = str(chunk)
    enc = MdecEncoder(0, w, h, max_frame_size=budget)
    row = []
    for data in (fr, fr8):
        n = data.shape[0]
        out = np.zeros((n, budget), np.uint8); res = np.zeros((n, 4), np.int32)
        for _ in range(2): enc.encode_frames_host(data, budget, out=out, res=res)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter(); enc.encode_frames_host(data, budget, out=out, res=res); best = min(best, time.perf_counter() - t)
        row.append("%d frames/call: %.0f frames/s (%.2f ms)" % (n, n / best, best * 1e3))
    print("chunk %5d  " % chunk + "  ".join(row), flush=True)
    enc.close()
