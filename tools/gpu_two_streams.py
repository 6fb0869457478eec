import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from psxavenc_amd import synth
from psxavenc_amd.mdec import MdecEncoder
dev = torch.device("cuda", 0)
w, h, n, budget = 320, 240, 1000, 8192
encs = [MdecEncoder(0, w, h, max_frame_size=budget, device=0) for _ in range(2)]
strs = [torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)]
if len(sys.argv) > 1: strs[0] = torch.cuda.Stream(device=dev)
outs = [(torch.zeros((n, budget), dtype=torch.uint8, device=dev), torch.zeros((n, 4), dtype=torch.int32, device=dev)) for _ in range(2)]
b4 = [synth.frames_device(w, h, 301 + b, 0, n, 4, device=0) for b in range(4)]
def both(launches):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(strs[0]); strs[1].wait_stream(strs[0])
    for k in range(launches):
        i = k & 1
        encs[i].encode_frames_device(b4[k % 4], budget, d_out=outs[i][0], d_results=outs[i][1], stream=strs[i])
    strs[0].wait_stream(strs[1]); b.record(strs[0]); torch.cuda.synchronize()
    return a.elapsed_time(b) / launches
for L in (8, 128, 800, 800):
    ms = both(L); print(L, ms, n / ms * 1e3)
