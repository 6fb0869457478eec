"""priority turn-taking patterns (PSXHIP_MDEC_PRIO) x workloads -> ms per launch (min of 3 runs of 40 launches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psxavenc_amd import synth
from psxavenc_amd.mdec import MdecEncoder
CASES = {"a4": (0, 320, 240, 8192, 1000, 4), "a8": (0, 320, 240, 8192, 1000, 8), "a16": (0, 320, 240, 8192, 1000, 16),
         "v2_16k": (0, 320, 240, 16128, 1000, 4), "a2": (0, 320, 240, 8192, 1000, 2)}
pats = [int(x, 0) for x in sys.argv[1:]] or [0x2EE01]
data = {}
for name, (codec, w, h, budget, n, amp) in CASES.items():
    d = synth.frames_device(w, h, 1, 0, n, amp, device=0)
    out = torch.zeros((n, budget), dtype=torch.uint8, device="cuda")
    res = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    data[name] = (d, out, res)
for p in pats:
    os.environ["PSXHIP_MDEC_PRIO"] = hex(p)
    row = []
    for name, (codec, w, h, budget, n, amp) in CASES.items():
        d, out, res = data[name]
        enc = MdecEncoder(codec, w, h, max_frame_size=budget, device=0)
        for _ in range(5):
            enc.encode_frames_device(d, budget, d_out=out, d_results=res)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                enc.encode_frames_device(d, budget, d_out=out, d_results=res)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40)
        row.append("%s %.4f" % (name, best))
        enc.close()
    print(hex(p), " ".join(row), flush=True)
