"""Host-buffer path over a device LIST (psxhip_mdec_multi_encode_frames_host, psxhip_xa_encode_streams_host_multi):
rates for {0}, {0,0} and, when more GPUs are visible, {0..N-1}; static ranges vs the host ticket queue on a batch whose
second half is twice as expensive as its first.  PCIe-inclusive numbers -- never bench.py's `value`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from psxavenc_amd import multi
from psxavenc_amd.mdec import register_host, unregister_host

w, h, budget, n = 320, 240, 8192, 8000
ngpu = torch.cuda.device_count()
fr = np.concatenate([O.synth_frames(w, h, 1000, seed=1, amp=4)] * (n // 2000) + [O.synth_frames(w, h, 1000, seed=2, amp=8)] * (n // 2000))
out = np.zeros((n, budget), np.uint8); res = np.zeros((n, 4), np.int32)
register_host(fr); register_host(out)
lists = [(0,), (0, 0)] + ([tuple(range(ngpu))] if ngpu > 1 else [])
rows = []
want = None
for devs in lists:
    m = multi.MdecMulti(devs, 0, w, h, budget)
    for sched, name in ((multi.SCHED_STATIC, "static"), (multi.SCHED_TICKETS, "tickets")):
        m.encode_frames_host(fr, budget, schedule=sched, out=out, res=res)
        t = time.perf_counter()
        for _ in range(3):
            m.encode_frames_host(fr, budget, schedule=sched, out=out, res=res)
        dt = (time.perf_counter() - t) / 3
        if want is None:
            want = out.copy()
        rows.append({"devices": list(devs), "schedule": name, "frames_per_sec": round(n / dt, 1), "ms_per_call": round(dt * 1e3, 3),
                     "identical_to_single_device": bool(np.array_equal(out, want)),
                     "per_worker": [{"units": r["units"], "tickets": r["tickets"], "seconds": round(r["seconds"], 5)} for r in m.last_report]})
        print(rows[-1], flush=True)
    m.close()
unregister_host(fr); unregister_host(out)
json.dump({"workload": "%d frames 320x240 v2 budget 8192 (first half noise +-4, second half +-8), page-locked caller buffers" % n, "gpus_visible": ngpu,
           "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "r03_multi_host.json"), "w"), indent=1)
