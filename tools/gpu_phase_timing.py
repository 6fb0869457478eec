"""Phase breakdown of the MDEC frame kernel (PSXHIP_MDEC_TIMING=1)."""
import os, sys, ctypes as C
os.environ["PSXHIP_MDEC_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from psxavenc_amd import synth, _lib
from psxavenc_amd.mdec import MdecEncoder
w, h, n, budget = 320, 240, 1000, 8192
for amp in (4, 8):
    enc = MdecEncoder(0, w, h, max_frame_size=budget)
    d = synth.frames_device(w, h, 1, 0, n, amp)
    for _ in range(3): enc.encode_frames_device(d, budget)
    t = (C.c_ulonglong * 16)()
    L = _lib.lib(); L.psxhip_mdec_read_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.psxhip_mdec_read_timing(enc._h, t, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(10): enc.encode_frames_device(d, budget)
    e1.record(); torch.cuda.synchronize()
    L.psxhip_mdec_read_timing(enc._h, t, 1)
    v = np.array(list(t), dtype=np.float64); tot = v[:7].sum()
    wgs = min(n, 512)
    print("amp", amp, "prologue cycles/WG %.0f" % (v[7] / (10 * wgs)), "ms/launch %.4f" % (e0.elapsed_time(e1) / 10), "phase %:", np.round(100 * v / tot, 1).tolist(), "cycles/frame %.0f" % (tot / (10 * n)), "wave0 loop(A): dct part %.0f cnt part %.0f cycles/frame" % (v[8] / (10 * n), v[9] / (10 * n)))
    enc.close()
