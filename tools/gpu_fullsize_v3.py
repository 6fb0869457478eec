#!/usr/bin/env python3
"""Config 'sbs v3' at full size on one GPU: this GPU's share of the 10 000 frames (frames [g*1250, (g+1)*1250) for GPU g,
SURVEY 8(d) config 4), 640x480, codec v3, at both budgets the survey names (8192 with noise +-4, 32768 with noise +-8).
Every frame: header fields, zero tail, decodability with the oracle's BS reader, bits/bytes/hwords consistency;
every 16th frame: byte-for-byte against the oracle.  Prints one JSON line per budget.
usage: python tools/gpu_fullsize_v3.py [gpu_index_of_8]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from psxavenc_amd import synth
from psxavenc_amd.mdec import MdecEncoder
from psxavenc_amd.parallel import shard_range

g = int(sys.argv[1]) if len(sys.argv) > 1 else 0
w, h = 640, 480
first, n = shard_range(10000, g, 8)
for budget, amp in ((8192, 4), (32768, 8)):
    enc = MdecEncoder(1, w, h, max_frame_size=budget)
    d = synth.frames_device(w, h, 1, first, n, amp, device=0)
    d_out, d_res = enc.encode_frames_device(d, budget)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        enc.encode_frames_device(d, budget, d_out=d_out, d_results=d_res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out, res = d_out.cpu().numpy(), d_res.cpu().numpy()
    ok = bool(((res[:, 0] >= 1) & (res[:, 0] <= 63)).all())
    ok &= bool((out[:, 2] == 0).all() and (out[:, 3] == 0x38).all() and (out[:, 6] == 3).all() and (out[:, 7] == 0).all())
    ok &= bool(np.array_equal(out[:, 4].astype(np.int32) | (out[:, 5].astype(np.int32) << 8), res[:, 0]))
    ok &= bool(np.array_equal(out[:, 0].astype(np.int32) | (out[:, 1].astype(np.int32) << 8), res[:, 2]))
    t0 = time.time()
    decoded = 0
    for k in range(n):
        ok &= not out[k, res[k, 1]:budget].any()
        rc2, levels, scale, version, nbits = O.mdec_decode(w, h, out[k, :budget])
        ok &= rc2 == 0 and version == 3 and scale == res[k, 0]
        ok &= res[k, 1] == ((8 + 2 * ((nbits + 15) // 16) + 3) & ~3)
        ok &= res[k, 3] == ((int(np.count_nonzero(levels[:, 1:])) + 2 * levels.shape[0] + 2 + 63) & ~63)
        decoded += 1
    idx = list(range(0, n, 16))
    fr = d[idx].cpu().numpy()
    want, want_res, rc = O.mdec_encode(1, w, h, fr, budget)
    exact = bool(rc == 0 and np.array_equal(out[idx][:, :budget], want) and np.array_equal(res[idx], want_res))
    sc, cnt = np.unique(res[:, 0], return_counts=True)
    print(json.dumps({"config": "sbs v3, frames [%d, %d) of 10000 (GPU %d of 8), 640x480, budget %d, noise +-%d" % (first, first + n, g, budget, amp),
                      "frames": n, "ms_per_launch": round(ms, 4), "frames_per_sec": round(n / ms * 1e3, 1),
                      "properties_ok_all_frames": bool(ok), "frames_decoded_with_oracle_reader": decoded,
                      "oracle_diff_sample": {"frames": len(idx), "bit_exact": exact},
                      "quant_scale_hist": {str(int(a)): int(b) for a, b in zip(sc, cnt)}, "check_seconds": round(time.time() - t0, 1)}), flush=True)
    enc.close()
