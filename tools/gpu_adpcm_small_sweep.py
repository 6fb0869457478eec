"""chunking of a SHORT XA stream (the strcd size: 2 chains x 90 000 units): time and verify passes vs chunk / warm-up length"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from psxavenc_amd import adpcm, synth
for n_chains, n_units in ((2, 90000), (2, 22500), (1, 400000), (8, 90000)):
    n = n_units * 28
    d = torch.empty((n_chains, n), dtype=torch.int16, device="cuda:0")
    for kind in (0,):
        for c in range(n_chains):
            synth.pcm_device(1, c, 0, n, kind, out=d[c])
        chains = adpcm.make_chains(np.arange(n_chains) * n, 1, n, n_units)
        base = np.arange(n_chains, dtype=np.int32) * n_units
        for chunk, warm in ((64, 16), (64, 32), (128, 32), (128, 64), (256, 32), (256, 64), (512, 64), (1024, 64), (2048, 64)):
            adpcm.encode_chains_device(d.reshape(-1), chains, base, 4, 4, chunk_units=chunk, warmup_units=warm)
            torch.cuda.synchronize(); t = time.perf_counter()
            u, s, passes = adpcm.encode_chains_device(d.reshape(-1), chains, base, 4, 4, chunk_units=chunk, warmup_units=warm)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            print("%d chains x %6d units, chunk %4d warm %3d: %7.2f ms, %3d passes" % (n_chains, n_units, chunk, warm, dt * 1e3, passes), flush=True)
