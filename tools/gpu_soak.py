"""Soak: many seeded frames / budgets / codecs through the batched HIP path, every byte against the oracle.
usage: gpu_soak.py [rounds [seed [frames_small frames_large]]]; with frame counts above the number of workgroups in flight (512)
the whole batch goes through ONE device launch, so that frames are handed on between workgroups (the retry queue)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from psxavenc_amd.mdec import MdecEncoder
import ctypes as C
from psxavenc_amd import _lib


def budget_limit(codec, w, h):
    class Geo(C.Structure):
        _fields_ = [("fits", C.c_int32), ("groups_per_cu", C.c_int32), ("wavefronts_per_group", C.c_int32), ("frames_in_flight", C.c_int32),
                    ("max_frame_size_limit", C.c_int32), ("reserved", C.c_int32), ("lds_bytes_per_group", C.c_int64), ("lds_bytes_per_cu", C.c_int64)]
    geo = Geo()
    _lib.check(_lib.lib().psxhip_mdec_query_geometry(0, codec, w, h, 8192, C.byref(geo)))
    return int(geo.max_frame_size_limit)


rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)      # second argument: another seed, other frames and budgets
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 500
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 60
ONE_LAUNCH = len(sys.argv) > 3
total = bad = 0
t0 = time.time()
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    codec = rnd % 3
    w, h = [(320, 240), (320, 240), (160, 112), (640, 480)][rnd % 4]
    n = NS if w <= 320 else NL
    amp = int(rng.integers(0, 40))
    frames = O.synth_frames(w, h, n, seed=int(rng.integers(1, 1 << 30)), amp=amp, first=int(rng.integers(0, 100000)))
    lo = 8 + 2 * (((w // 16) * (h // 16) * 6 * 12 + 10 + 15) // 16)
    budgets = rng.integers(lo + 200, lo + 200 + int(rng.integers(500, 40000)), n).astype(np.int32)
    budgets = np.minimum(budgets, budget_limit(codec, w, h))     # a frame's working set has to fit the CU's LDS
    want, want_res, rc = O.mdec_encode(codec, w, h, frames, budgets, stride=int(budgets.max()))
    if rc != 0:
        # some frame fits no scale: encode one by one
        keep = []
        for k in range(n):
            _, _, r1 = O.mdec_encode(codec, w, h, frames[k:k + 1], int(budgets[k]))
            if r1 == 0: keep.append(k)
        frames, budgets = frames[keep], budgets[keep]
        want, want_res, rc = O.mdec_encode(codec, w, h, frames, budgets, stride=int(budgets.max()))
        n = len(keep)
    enc = MdecEncoder(codec, w, h, max_frame_size=int(budgets.max()))
    if ONE_LAUNCH:
        import torch
        d_out, d_res = enc.encode_frames_device(torch.from_numpy(frames).to("cuda:0"), torch.from_numpy(budgets).to("cuda:0"))
        torch.cuda.synchronize()
        out, res = d_out.cpu().numpy()[:, :int(budgets.max())], d_res.cpu().numpy()
        want = want[:, :int(budgets.max())]
        for k in range(n):            # bytes past a frame's own budget are not the encoder's
            out[k, budgets[k]:] = 0; want[k, budgets[k]:] = 0
    else:
        out, res = enc.encode_frames_host(frames, budgets)
    ok = np.array_equal(out, want) and np.array_equal(res, want_res)
    total += n; bad += 0 if ok else 1
    print("round %2d codec %d %dx%d amp %2d frames %3d scales %2d..%2d %s" % (rnd, codec, w, h, amp, n, res[:, 0].min(), res[:, 0].max(), "ok" if ok else "MISMATCH"), flush=True)
    enc.close()
print("soak: %d frames, %d mismatching rounds, %.0f s" % (total, bad, time.time() - t0))
