"""quarter-pass checkpoint margin (PSXHIP_MDEC_CKMARGIN, thousandths of the projection's standard error) x workloads -> ms per launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psxavenc_amd import synth
from psxavenc_amd.mdec import MdecEncoder
CASES = {"a4": (0, 320, 240, 8192, 1000, 4), "a8": (0, 320, 240, 8192, 1000, 8), "a24_4k": (0, 320, 240, 4096, 1000, 24), "a12": (0, 320, 240, 8192, 1000, 12), "v3_8k": (1, 640, 480, 8192, 1250, 4), "v3_32k": (1, 640, 480, 32768, 1250, 8), "a6": (0, 320, 240, 8192, 1000, 6), "a10_12k": (0, 320, 240, 12000, 1000, 10), "v3_a10": (1, 320, 240, 8192, 1000, 10)}
for m in ([int(x) for x in sys.argv[1:]] or [1500, 800, 400, 300, 200, 100, 1]):
    os.environ["PSXHIP_MDEC_CKMARGIN"] = str(m)
    row = []
    for name, (codec, w, h, budget, n, amp) in CASES.items():
        enc = MdecEncoder(codec, w, h, max_frame_size=budget, device=0)
        d = synth.frames_device(w, h, 1, 0, n, amp, device=0)
        out = torch.zeros((n, budget), dtype=torch.uint8, device="cuda"); res = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        for _ in range(5): enc.encode_frames_device(d, budget, d_out=out, d_results=res)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): enc.encode_frames_device(d, budget, d_out=out, d_results=res)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30)
        row.append("%s %.4f" % (name, best)); enc.close()
    print("margin %4d / 1000 standard errors: " % m + "  ".join(row), flush=True)
