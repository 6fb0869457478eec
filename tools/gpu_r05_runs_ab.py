"""A/B of the frame hand-out on uniform content (product build): ms per 1000-frame launch, one lane and two, under a few settings of
PSXHIP_MDEC_RUN / _NO_SPARE / _NO_RETRY_QUEUE (read when a context is created).  usage: python tools/gpu_r05_runs_ab.py [n_frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psxavenc_amd import synth
from psxavenc_amd.mdec import MdecEncoder
W, H, B = 320, 240, 8192
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
amp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bb = [synth.frames_device(W, H, 101 + b, 0, N, amp, device=0) for b in range(4)]
outs = [(torch.zeros((N, B), dtype=torch.uint8, device="cuda"), torch.zeros((N, 4), dtype=torch.int32, device="cuda")) for _ in range(4)]
def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for k in range(reps): fn(k)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for env in ({"PSXHIP_MDEC_RUN": "1"}, {"PSXHIP_MDEC_RUN": "2"}, {"PSXHIP_MDEC_RUN": "2", "PSXHIP_MDEC_NO_SPARE": "1"},
            {"PSXHIP_MDEC_RUN": "2", "PSXHIP_MDEC_NO_RETRY_QUEUE": "1"}, {"PSXHIP_MDEC_RUN": "1", "PSXHIP_MDEC_NO_RETRY_QUEUE": "1"}, {"PSXHIP_MDEC_RUN": "4"}):
    for k in ("PSXHIP_MDEC_RUN", "PSXHIP_MDEC_NO_SPARE", "PSXHIP_MDEC_NO_RETRY_QUEUE"): os.environ.pop(k, None)
    os.environ.update(env)
    r = []
    for lanes in (1, 2):
        enc = MdecEncoder(0, W, H, max_frame_size=B, device=0)
        if lanes > 1: enc.set_lanes(2)
        def one(k): enc.encode_frames_device(bb[k % 4], B, d_out=outs[k % 4][0], d_results=outs[k % 4][1])
        timed(one, 8); enc.fence()
        r.append(min(timed(lambda k: (one(k), enc.fence() if k == 63 else None), 64) for _ in range(3)))
        enc.close()
    print("n %d amp %d %-70s lanes1 %.4f ms  lanes2 %.4f ms" % (N, amp, env, r[0], r[1]), flush=True)
