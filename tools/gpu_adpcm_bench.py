"""ADPCM throughput on the device-resident path (development aid; numbers quoted in DESIGN.md)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from psxavenc_amd import _lib, adpcm, synth
import oracle_lib as O

L = adpcm._bind()
dev = torch.device("cuda", 0)

def run(n_chains, units_per_chain, filter_count, bits, kind=0, reps=5):
    n = units_per_chain * 28
    pcm = torch.empty((n_chains, n), dtype=torch.int16, device=dev)
    for c in range(min(n_chains, 64)):
        synth.pcm_device(7, c, 0, n, kind, out=pcm[c])
    if n_chains > 64:
        pcm[64:] = pcm[:64].repeat((n_chains + 63) // 64, 1)[:n_chains - 64]
    chains = np.zeros(n_chains, dtype=[("off", "<i8"), ("pitch", "<i4"), ("limit", "<i4"), ("units", "<i4"), ("stride", "<i4")])
    chains["off"] = np.arange(n_chains) * n; chains["pitch"] = 1; chains["limit"] = n; chains["units"] = units_per_chain; chains["stride"] = 1
    d_chains = torch.from_numpy(chains.view(np.uint8)).to(dev)
    d_base = torch.arange(n_chains, dtype=torch.int32, device=dev) * units_per_chain
    d_states = torch.zeros((n_chains, 2), dtype=torch.int32, device=dev)
    d_units = torch.zeros((n_chains * units_per_chain, 16 if bits == 4 else 32), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def go():
        d_states.zero_()
        _lib.check(L.psxhip_adpcm_encode_chains_device(0, pcm.data_ptr(), d_chains.data_ptr(), d_base.data_ptr(), n_chains, filter_count, bits, d_states.data_ptr(), d_units.data_ptr(), st))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    units = n_chains * units_per_chain
    print("chains %6d units/chain %6d filt %d bits %d: %8.3f ms  %10.0f units/s  (%.1f us/unit/chain)" % (n_chains, units_per_chain, filter_count, bits, ms, units / ms * 1e3, ms * 1e3 / units_per_chain))
    return pcm, d_units

run(16, 4000, 4, 4)          # xacd shape: 16 chains
run(64, 4000, 4, 4)
run(1024, 1000, 5, 4)        # many SPU streams
run(16384, 200, 5, 4)
pcm, d_units = run(4096, 500, 5, 4)
# CPU oracle speed for comparison (1 core)
x = pcm[0].cpu().numpy()
t = time.perf_counter(); 
for _ in range(20): O.spu_encode(x)
dt = time.perf_counter() - t
print("oracle 1 core: %.0f units/s" % (20 * 500 / dt))
if O.ref() is not None:
    t = time.perf_counter()
    for _ in range(20): O.ref_spu_encode(x)
    dt = time.perf_counter() - t
    print("reference build (oracle/_ref, -O3 -ffast-math) 1 core: %.0f units/s" % (20 * 500 / dt))
