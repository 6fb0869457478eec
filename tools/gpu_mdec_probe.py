#!/usr/bin/env python3
"""Quick MDEC kernel probe on the GPU box: fps + passes-per-frame statistics for a few workloads.
usage: PSXHIP_MDEC_STATS=1 python tools/gpu_mdec_probe.py [name ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PSXHIP_MDEC_STATS", "1")
import numpy as np
import torch
from psxavenc_amd import _lib, synth
from psxavenc_amd.mdec import MdecEncoder

CASES = {
    "a4": (0, 320, 240, 8192, 1000, 4), "a8": (0, 320, 240, 8192, 1000, 8), "a2": (0, 320, 240, 8192, 1000, 2),
    "v2_16k": (0, 320, 240, 16128, 1000, 4), "v3_8k": (1, 640, 480, 8192, 1250, 4), "v3_32k": (1, 640, 480, 32768, 1250, 8),
    "a16": (0, 320, 240, 8192, 1000, 16), "a24_4k": (0, 320, 240, 4096, 1000, 24), "a6": (0, 320, 240, 8192, 1000, 6),
}


def run(name, reps=20):
    codec, w, h, budget, n, amp = CASES[name]
    enc = MdecEncoder(codec, w, h, max_frame_size=budget, device=0)
    d = synth.frames_device(w, h, 1, 0, n, amp, device=0)
    out = torch.zeros((n, (budget + 3) & ~3), dtype=torch.uint8, device="cuda")
    res = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    for _ in range(3):
        enc.encode_frames_device(d, budget, d_out=out, d_results=res)
    torch.cuda.synchronize()
    L = _lib.lib()
    st = (C.c_ulonglong * 8)()
    L.psxhip_mdec_read_stats(enc._h, st, 8, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        enc.encode_frames_device(d, budget, d_out=out, d_results=res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    L.psxhip_mdec_read_stats(enc._h, st, 8, 1)
    s = list(st)
    sc, cnt = np.unique(res[:, 0].cpu().numpy(), return_counts=True)
    print("%-8s %8.4f ms  %10.0f fps  passes/frame %.3f  hist(0,1,2,3,4,5+) %s  scales %s" % (
        name, ms, n / ms * 1e3, s[1] / max(1, s[0]), s[2:8], dict(zip(sc.tolist(), cnt.tolist()))), flush=True)
    if os.environ.get("PROBE_TRACE"):
        NT = 8 + 4 * 1024 + 16
        t = (C.c_ulonglong * NT)()
        enc.encode_frames_device(d, budget, d_out=out, d_results=res)
        torch.cuda.synchronize()
        L.psxhip_mdec_read_stats(enc._h, t, NT, 1)
        ph = np.array(list(t)[8 + 4096:], dtype=np.float64)
        print("   phase share %% (ticket/idle, reset+dc, pilot, passes, scan+merge, header+writeout): %s  total %.1f us/group" % (np.round(100 * ph[:6] / ph[:6].sum(), 1).tolist(), ph[:6].sum() / 100.0 / 512))
        print("   wavefront time at group barriers: %.1f%% of wavefront residency" % (100.0 * ph[6] / max(1.0, ph[7])))
        print("   by barrier (ticket+reset, dc+pilot, checkpoint, end of pass, search step, merge+writeout) %% of residency:", np.round(100.0 * ph[8:14] / max(1.0, ph[7]), 1).tolist())
        a = np.array(list(t)[8:8 + 4096], dtype=np.int64).reshape(-1, 4)
        a = a[a[:, 1] > 0]
        t0 = a[:, 0].min()
        st, en, nf = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2] & 0xFF      # microseconds
        pro, first = ((a[:, 2] >> 8) & 0xFFFFFF) / 100.0, ((a[:, 2] >> 32) & 0xFFFFFF) / 100.0
        print("   prologue us: p10 %.1f p50 %.1f p90 %.1f | first frame done (from group entry) us: p10 %.1f p50 %.1f p90 %.1f max %.1f"
              % (np.percentile(pro, 10), np.percentile(pro, 50), np.percentile(pro, 90), np.percentile(first, 10), np.percentile(first, 50), np.percentile(first, 90), first.max()))
        print("   groups %d  start us: p0 %.1f p50 %.1f p90 %.1f max %.1f | end us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | frames/group hist %s"
              % (len(a), st.min(), np.percentile(st, 50), np.percentile(st, 90), st.max(), en.min(), np.percentile(en, 10),
                 np.percentile(en, 50), np.percentile(en, 90), en.max(), np.bincount(nf).tolist()))
        print("   mean residency %.1f%% of the span" % (100 * (en - st).mean() / en.max()))
        idx = np.arange(len(en))
        for x in range(8):
            m = (idx % 8) == x
            print("   xcd-slot %d: end mean %.1f max %.1f frames %s" % (x, en[m].mean(), en[m].max(), np.bincount(nf[m], minlength=4).tolist()))
        hw = a[:, 3] & 0xFFFFFFFF
        xcc = (a[:, 3] >> 32) & 0xF
        slot = hw & 0xF
        cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
        where = xcc * 1000 + se * 100 + sh * 50 + cu          # one number per CU
        order = np.argsort(en)
        print("   slowest 16 groups (xcc, se, sh, cu, slot, end us):", [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(slot[i]), round(float(en[i]), 1)) for i in order[-16:]])
        for x in sorted(set(xcc.tolist())):
            m = xcc == x
            print("   xcc %d: groups %d  end mean %.1f max %.1f  CUs %d" % (x, m.sum(), en[m].mean(), en[m].max(), len(set(where[m].tolist()))))
        percu = {}
        for i in range(len(en)): percu.setdefault(int(where[i]), []).append(float(en[i]))
        cu_end = np.array([max(v) for v in percu.values()])
        print("   per-CU end time (max of its groups): CUs %d  p10 %.1f p50 %.1f p90 %.1f max %.1f | groups per CU hist %s"
              % (len(cu_end), np.percentile(cu_end, 10), np.percentile(cu_end, 50), np.percentile(cu_end, 90), cu_end.max(), np.bincount([len(v) for v in percu.values()]).tolist()))
        if os.environ.get("PROBE_TRACE_DUMP"):
            np.save(os.environ["PROBE_TRACE_DUMP"] + "_%s.npy" % name, np.stack([where, slot, st, en, nf], 1))
        for sl in sorted(set(slot.tolist())):
            m = slot == sl
            print("   wave slot %d: groups %d  end mean %.1f  frames %s" % (sl, m.sum(), en[m].mean(), np.bincount(nf[m], minlength=4).tolist()))
        order = np.argsort(en)
        print("   slowest 12 groups (idx, end us, frames):", [(int(i), round(float(en[i]), 1), int(nf[i])) for i in order[-12:]])
        print("   fastest 12 groups (idx, end us, frames):", [(int(i), round(float(en[i]), 1), int(nf[i])) for i in order[:12]])
        print("   end-time histogram (20us bins):", np.histogram(en, bins=np.arange(0, en.max() + 20, 20))[0].tolist())
    enc.close()


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["a4", "a8", "v2_16k", "v3_8k", "v3_32k"]):
        run(nm)
