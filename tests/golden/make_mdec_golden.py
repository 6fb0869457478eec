#!/usr/bin/env python3
"""Generate tests/golden/mdec_selfgolden.npz from this repo's CPU restatement (oracle/mdec_oracle.c).

NOT reference output: psxavenc/mdec.c cannot be compiled in this image (it needs FFmpeg's
libavcodec/avdct.h and stand-in headers are not allowed), so these vectors pin the restatement
against regressions and give the GPU box fixed expectations; the MDEC parity claim stays
"unpinned at the FDCT" (DESIGN.md).  Inputs are regenerated at test time from oracle/synth.c.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

CASES = []
for codec in (0, 1, 2):
    for (w, h) in ((320, 240), (640, 480), (16, 16), (48, 32)):
        for budget in (8192, 8191, 16128, 18144, 4096, 32768):
            for amp in (0, 4, 8):
                CASES.append((codec, w, h, budget, amp))


def special_frames(w, h):
    """hand-made stress frames: hard edges (escape codes), DC ties for v3, flat, near-white-noise"""
    n = w * h
    out = []
    yy, xx = np.mgrid[0:h, 0:w]
    f = np.full(n * 3 // 2, 128, np.uint8)                      # flat mid-grey
    out.append(f.copy())
    f = np.full(n * 3 // 2, 128, np.uint8)                      # flat value giving dc = 2 (mod 4) ties: (130-128)*64/16 = 8 -> no; use 129 -> dc=4; 128+x
    f[:n] = 128 + 1                                              # dc = round(64/16)=4
    out.append(f.copy())
    for v in (3, 5, 7, 250, 1):                                  # assorted flat levels: dc = round(64*(v-128)/16) hits 2 mod 4 for odd multiples
        f = np.full(n * 3 // 2, 128, np.uint8)
        f[:n] = v
        f[n:] = 255 - v
        out.append(f.copy())
    f = np.full(n * 3 // 2, 128, np.uint8)                      # checkerboard of 8x8 tiles, full contrast: big DC swings (v3 deltas near +-255)
    f[:n] = (((yy // 8 + xx // 8) & 1) * 255).astype(np.uint8).ravel()
    out.append(f.copy())
    f = np.full(n * 3 // 2, 128, np.uint8)                      # vertical hard edges every 5 px: large AC levels / escapes
    f[:n] = (((xx // 5) & 1) * 255).astype(np.uint8).ravel()
    out.append(f.copy())
    f = np.full(n * 3 // 2, 128, np.uint8)                      # per-block DC staircase in steps of 2 quant units (tie-heavy for v3)
    f[:n] = np.clip(96 + ((yy // 8) * 3 + (xx // 8) * 5) % 64, 0, 255).astype(np.uint8).ravel()
    out.append(f.copy())
    return np.stack(out)


def main():
    out = {}
    rows = []
    for (codec, w, h, budget, amp) in CASES:
        n = 2 if w >= 640 else 4
        fr = O.synth_frames(w, h, n, seed=100 + amp, amp=amp, first=3)
        data, res, rc = O.mdec_encode(codec, w, h, fr, budget)
        sha = hashlib.sha256(data.tobytes()).digest() if rc == 0 else b"\0" * 32
        rows.append([codec, w, h, budget, amp, n, rc] + (res.ravel().tolist() if rc == 0 else [0] * (4 * n)) + [0] * (4 * (4 - n)))
        out["sha_c%d_%dx%d_b%d_a%d" % (codec, w, h, budget, amp)] = np.frombuffer(sha, np.uint8)
    out["table"] = np.array(rows, np.int32)
    # stress frames at 320x240 and 48x32, all codecs, generous + tight budgets; keep full outputs for the small size
    for codec in (0, 1, 2):
        for (w, h, budget) in ((48, 32, 4096), (320, 240, 30000), (320, 240, 9000)):
            fr = special_frames(w, h)
            rcs, shas, ress = [], [], []
            for k in range(fr.shape[0]):
                data, res, rc = O.mdec_encode(codec, w, h, fr[k:k + 1], budget)
                rcs.append(rc)
                shas.append(np.frombuffer(hashlib.sha256(data.tobytes()).digest(), np.uint8))
                ress.append(res[0])
                if w == 48 and codec == 1 and rc == 0:
                    out["full_c1_48x32_k%d" % k] = data[0, :res[0, 1]]
            key = "special_c%d_%dx%d_b%d" % (codec, w, h, budget)
            out[key + "_rc"] = np.array(rcs, np.int32)
            out[key + "_sha"] = np.stack(shas)
            out[key + "_res"] = np.stack(ress).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "mdec_selfgolden.npz"), **out)
    t = out["table"]
    print("wrote mdec_selfgolden.npz:", len(CASES), "cases; rc histogram", {int(v): int((t[:, 6] == v).sum()) for v in np.unique(t[:, 6])})
    print("scale range", t[t[:, 6] == 0][:, 7].min(), t[t[:, 6] == 0][:, 7].max())


if __name__ == "__main__":
    main()
