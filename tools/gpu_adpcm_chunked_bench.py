"""Chunked (speculate-and-verify) vs serial ADPCM on long chains -- development aid, numbers quoted in DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from psxavenc_amd import adpcm, synth

def run(n_chains, n_units, kind, chunk, warm):
    n = n_units * 28
    d = torch.empty((n_chains, n), dtype=torch.int16, device="cuda:0")
    for c in range(n_chains):
        synth.pcm_device(5, c, 0, n, kind, out=d[c])
    chains = adpcm.make_chains(np.arange(n_chains) * n, 1, n, n_units)
    base = np.arange(n_chains, dtype=np.int32) * n_units
    torch.cuda.synchronize()
    res = {}
    for label, cu in (("serial", 0), ("chunked", chunk)):
        adpcm.encode_chains_device(d.reshape(-1), chains, base, 4, 4, chunk_units=cu, warmup_units=warm)
        torch.cuda.synchronize()
        t = time.perf_counter()
        u, s, passes = adpcm.encode_chains_device(d.reshape(-1), chains, base, 4, 4, chunk_units=cu, warmup_units=warm)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        res[label] = (dt, passes, u, s)
    same = torch.equal(res["serial"][2], res["chunked"][2]) and torch.equal(res["serial"][3], res["chunked"][3])
    tot = n_chains * n_units
    print("kind %d chains %3d units/chain %8d chunk %3d warm %2d: serial %8.1f ms (%.2f Munits/s)  chunked %8.1f ms (%.1f Munits/s, %d passes)  identical=%s"
          % (kind, n_chains, n_units, chunk, warm, res["serial"][0] * 1e3, tot / res["serial"][0] / 1e6,
             res["chunked"][0] * 1e3, tot / res["chunked"][0] / 1e6, res["chunked"][1], same))

for kind in (0, 1, 2, 5, 4):
    run(16, 400000, kind, 64, 16)
run(16, 400000, 0, 256, 32)
run(16, 400000, 0, 32, 16)
run(16, 2000000, 0, 128, 32)
