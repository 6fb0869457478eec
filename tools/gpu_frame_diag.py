"""per-frame search diagnostics of one launch (PSXHIP_MDEC_STATS build): first guess, first checkpoint verdict, answer, passes.
usage: python tools/gpu_frame_diag.py [noise amplitude] [launches before the one that is reported]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PSXHIP_MDEC_STATS"] = "1"
import numpy as np, torch
from psxavenc_amd import _lib, synth
from psxavenc_amd.mdec import MdecEncoder
w, h, budget, n, amp = 320, 240, 8192, 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = synth.frames_device(w, h, 1, 0, n, amp, device=0)
out = torch.zeros((n, budget), dtype=torch.uint8, device="cuda"); res = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
enc = MdecEncoder(0, w, h, max_frame_size=budget, device=0)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4): enc.encode_frames_device(d, budget, d_out=out, d_results=res)
torch.cuda.synchronize()
NT = 8 + 4 * 1024 + 16 + 2048
t = (C.c_ulonglong * NT)()
_lib.lib().psxhip_mdec_read_stats(enc._h, t, NT, 0)
fr = np.array(list(t)[8 + 4096 + 16:8 + 4096 + 16 + n], dtype=np.int64)
guess, ab, ans, np_ = fr & 0xFF, (fr >> 8) & 0xFF, (fr >> 16) & 0xFF, (fr >> 24) & 0xFF
import collections
c = collections.Counter(zip(guess.tolist(), ab.tolist(), ans.tolist(), np_.tolist()))
for k, v in sorted(c.items(), key=lambda x: -x[1])[:14]: print("guess %d abort->%d answer %d passes %d : %d frames" % (k + (v,)))
print("answers by frame index (runs):", [(int(a), int(l)) for a, l in zip(*[x for x in (lambda a: (a[np.r_[True, a[1:] != a[:-1]]], np.diff(np.r_[np.nonzero(np.r_[True, a[1:] != a[:-1]])[0], len(a)])))(ans)])][:30])
