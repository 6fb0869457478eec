"""PCIe-inclusive rates of the host-buffer entry points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from psxavenc_amd.mdec import MdecEncoder
w, h, budget = 320, 240, 8192
fr = O.synth_frames(w, h, 1000, seed=1, amp=4)
enc = MdecEncoder(0, w, h, max_frame_size=budget)
enc.frame_max_size = budget
for k in range(20): enc.encode_frame_bs(fr[k])
t = time.perf_counter()
for k in range(500): enc.encode_frame_bs(fr[k])
dt = time.perf_counter() - t
print("encode_frame_bs (one frame per call, pageable host buffers): %.0f frames/s (%.1f us per call)" % (500 / dt, dt / 500 * 1e6))
enc.encode_frames_host(fr, budget)
t = time.perf_counter()
for _ in range(5): enc.encode_frames_host(fr, budget)
dt = (time.perf_counter() - t) / 5
print("encode_frames_host (1000 frames per call, pageable host buffers, H2D+kernel+D2H): %.0f frames/s (%.2f ms per call)" % (1000 / dt, dt * 1e3))
fr8 = np.concatenate([fr] * 8)
out8 = enc.encode_frames_host(fr8, budget)
t = time.perf_counter()
for _ in range(3): enc.encode_frames_host(fr8, budget)
dt = (time.perf_counter() - t) / 3
print("encode_frames_host (8000 frames per call, pageable host buffers, chunked double-buffered pipeline): %.0f frames/s (%.2f ms per call)" % (8000 / dt, dt * 1e3))

# page-locked caller buffers: no staging copy, DMA straight from / to the caller's memory
from psxavenc_amd.mdec import register_host, unregister_host
out8 = np.zeros((8000, budget), dtype=np.uint8); res8 = np.zeros((8000, 4), dtype=np.int32)
t = time.perf_counter(); enc.encode_frames_host(fr8, budget, out=out8, res=res8); dt0 = time.perf_counter() - t
for _ in range(2): enc.encode_frames_host(fr8, budget, out=out8, res=res8)
t = time.perf_counter()
for _ in range(3): enc.encode_frames_host(fr8, budget, out=out8, res=res8)
dt = (time.perf_counter() - t) / 3
print("encode_frames_host (8000 frames per call, pageable, preallocated output): %.0f frames/s (%.2f ms per call)" % (8000 / dt, dt * 1e3))
want = out8.copy()
t = time.perf_counter(); register_host(fr8); register_host(out8); dtr = time.perf_counter() - t
enc.encode_frames_host(fr8, budget, out=out8, res=res8)
t = time.perf_counter()
for _ in range(3): enc.encode_frames_host(fr8, budget, out=out8, res=res8)
dt = (time.perf_counter() - t) / 3
print("encode_frames_host (8000 frames per call, PAGE-LOCKED caller buffers, DMA direct): %.0f frames/s (%.2f ms per call); registering 987 MB took %.0f ms; output identical: %s"
      % (8000 / dt, dt * 1e3, dtr * 1e3, np.array_equal(out8, want)))
unregister_host(fr8); unregister_host(out8)
