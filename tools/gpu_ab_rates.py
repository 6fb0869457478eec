#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box (box-to-box spread is about 1.5 %, more than a kernel revision's fixed costs):
one process per (library, round), alternating; each prints the in-order and two-lane time per 1000-frame launch of the content
class.  usage: python tools/gpu_ab_rates.py <libA.so> <libB.so> [more.so ...] [kind ...] [--rounds 3] [--json out.json]
(child: PSXAV_HIP_LIB=<lib> python tools/gpu_ab_rates.py --child kind)
An older revision's library: git worktree add /tmp/k36 <commit>; make -C /tmp/k36/psxavenc_amd/csrc; cp /tmp/k36/psxavenc_amd/libpsxav_hip.so
build_ab/libpsxav_hip_k36.so (build_ab/*.so is not tracked; it travels to the GPU box with the snapshot)."""
import json
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(kind):
    import torch
    sys.argv = [sys.argv[0]]
    import gpu_r05_diag as D
    from psxavenc_amd import _lib
    from psxavenc_amd.mdec import MdecEncoder
    bb = D.batches(kind)
    N, B = D.N, D.BUDGET
    outs = [(torch.zeros((N, B), dtype=torch.uint8, device="cuda"), torch.zeros((N, 4), dtype=torch.int32, device="cuda")) for _ in range(4)]
    r = {"library": _lib.lib().psxhip_version().decode()}
    for lanes in (1, 2):
        enc = MdecEncoder(D.CODEC, D.W, D.H, max_frame_size=B, device=0)
        if lanes > 1:
            enc.set_lanes(lanes)

        def one(k):
            enc.encode_frames_device(bb[k % 4], B, d_out=outs[k % 4][0], d_results=outs[k % 4][1])
        D.timed(one, 16)
        enc.fence()
        ms = sorted(D.timed(lambda k: (one(k), enc.fence() if k == 255 else None), 256) for _ in range(5))
        r["lanes%d_ms" % lanes] = [round(ms[0], 5), round(ms[2], 5)]
        enc.close()
    print("AB " + json.dumps(r), flush=True)


def main():
    argv = sys.argv[1:]
    if argv[0] == "--child":
        return child(argv[1])
    rounds, json_out = 3, None
    if "--rounds" in argv:
        i = argv.index("--rounds"); rounds = int(argv[i + 1]); del argv[i:i + 2]
    if "--json" in argv:
        i = argv.index("--json"); json_out = argv[i + 1]; del argv[i:i + 2]
    libs = [os.path.abspath(a) for a in argv if a.endswith(".so")]
    kinds = [a for a in argv if not a.endswith(".so")] or ["a4"]
    res = {}
    for kind in kinds:
        for rnd in range(rounds):
            for lib in libs:
                env = dict(os.environ, PSXAV_HIP_LIB=lib)
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind], env=env, capture_output=True, text=True, timeout=600)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("AB ")]
                if not line:
                    print(kind, os.path.basename(lib), "FAILED", out.stderr[-400:], flush=True)
                    continue
                d = json.loads(line[0][3:])
                res.setdefault(kind, {}).setdefault(os.path.basename(lib), []).append(d)
                print(kind, rnd, os.path.basename(lib), d, flush=True)
    if json_out:
        json.dump(res, open(json_out, "w"), indent=1)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
