"""one XA stream of the strcd size (1250 sectors, stereo 4-bit) through psxhip_xa_encode_streams_host: where the time goes"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from psxavenc_amd import adpcm
from psxavenc_amd.mdec import register_host, unregister_host
s = adpcm.XaSettings(format=1, stereo=True, frequency=37800, bits_per_sample=4)
for sectors in (1250, 5000):
    n = sectors * adpcm.xa_get_samples_per_sector(s)
    rng = np.random.default_rng(1)
    pcm = (rng.integers(-8000, 8000, n * 2)).astype(np.int16)
    for mode in ("pageable", "page-locked"):
        if mode == "page-locked": register_host(pcm)
        for _ in range(2): adpcm.xa_encode_streams(s, pcm[None, :], n)
        t = time.perf_counter()
        for _ in range(5): adpcm.xa_encode_streams(s, pcm[None, :], n)
        dt = (time.perf_counter() - t) / 5
        print("%d sectors, %s input: %.2f ms per call (%.0f sectors/s)" % (sectors, mode, dt * 1e3, sectors / dt))
        if mode == "page-locked": unregister_host(pcm)
