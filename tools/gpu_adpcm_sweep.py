import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from psxavenc_amd import adpcm, synth
n_chains, n_units = 16, 1350 * int(os.environ.get("SWEEP_SECONDS", "600"))   # 10 min of 37800 Hz per chain (28 samples per unit): 37800*600/28 = 810000
n = n_units * 28
d = torch.empty((n_chains, n), dtype=torch.int16, device="cuda:0")
for kind in [int(k) for k in os.environ.get("SWEEP_KINDS", "0,2").split(",")]:
    for c in range(n_chains):
        synth.pcm_device(5, c, 0, n, kind, out=d[c])
    chains = adpcm.make_chains(np.arange(n_chains) * n, 1, n, n_units)
    base = np.arange(n_chains, dtype=np.int32) * n_units
    pairs = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(256, 16), (256, 32), (512, 16), (512, 32), (512, 64), (1024, 16), (1024, 32), (1024, 64)]
    for chunk, warm in pairs:      # usage: gpu_adpcm_sweep.py [chunk:warm ...]
        adpcm.encode_chains_device(d.reshape(-1), chains, base, 4, 4, chunk_units=chunk, warmup_units=warm)
        torch.cuda.synchronize(); t = time.perf_counter()
        u, s, passes = adpcm.encode_chains_device(d.reshape(-1), chains, base, 4, 4, chunk_units=chunk, warmup_units=warm)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("kind %d chunk %4d warm %3d: %7.1f ms, %2d passes, %.0f Munits/s" % (kind, chunk, warm, dt * 1e3, passes, n_chains * n_units / dt / 1e6))
