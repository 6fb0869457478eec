"""Soak of the frame kernel's hint / pilot / trust-policy paths (mdec-k3.7): scene-structured sequences (psxavenc_amd/mixed.py: runs of
5..30 similar frames, cuts between noise amplitudes, hand-made flat / hard-edge / escape frames) with random per-frame budgets, several
consecutive launches per encoder context (the verdict on foreign hints travels from launch to launch), one lane and two, single-frame
tickets and runs of 2 / 4 -- every output byte and result field against the oracle (encoded on all host cores).
usage: gpu_soak_mixed.py [rounds [seed [frames per launch]]]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
from psxavenc_amd import mixed
from psxavenc_amd.mdec import MdecEncoder

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 555)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 900
O.lib()
pool = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1))


def oracle(codec, w, h, frames, budgets, stride):
    n = frames.shape[0]
    parts = [(a, min(n, a + 64)) for a in range(0, n, 64)]
    res = list(pool.map(lambda ab: O.mdec_encode(codec, w, h, frames[ab[0]:ab[1]], budgets[ab[0]:ab[1]], stride=stride), parts))
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res]), [r[2] for r in res]


total = bad = 0
t0 = time.time()
for rnd in range(rounds):
    codec = int(rng.integers(0, 3))
    w, h = [(320, 240), (320, 240), (160, 112), (640, 480)][rnd % 4]
    n = N if w <= 320 else max(64, N // 4)
    launches = int(rng.integers(2, 5))
    run = int(rng.choice([1, 1, 2, 4]))
    lanes = int(rng.integers(1, 3))
    seq = mixed.frames_host(O, w, h, int(rng.integers(1, 1 << 20)), int(rng.integers(0, 5000)), n * launches)
    lo = 8 + 2 * (((w // 16) * (h // 16) * 6 * 12 + 10 + 15) // 16)
    base = int(rng.integers(6000, 20000)) * (w * h) // (320 * 240) + lo
    budgets = (base + rng.integers(0, 3000, n * launches)).astype(np.int32)
    budgets = np.minimum(budgets, {(320, 240): 100000, (160, 112): 100000, (640, 480): 80000}[(w, h)])      # a frame's working set has to fit the CU's LDS (psxhip_mdec_query_geometry)
    stride = int(budgets.max())
    want, want_res, rcs = oracle(codec, w, h, seq, budgets, stride)
    keep = want_res[:, 0] < 64                      # (frames that fit no scale: the reference asserts; leave them out)
    if not keep.all():
        seq, budgets, want, want_res = seq[keep], budgets[keep], want[keep], want_res[keep]
    m = seq.shape[0] // launches
    if m < 8:
        continue
    os.environ["PSXHIP_MDEC_RUN"] = str(run)
    enc = MdecEncoder(codec, w, h, max_frame_size=stride)
    del os.environ["PSXHIP_MDEC_RUN"]
    if lanes > 1:
        enc.set_lanes(2)
    d_seq, d_bud = torch.from_numpy(seq).to("cuda:0"), torch.from_numpy(budgets).to("cuda:0")
    outs = []
    for k in range(launches):
        outs.append(enc.encode_frames_device(d_seq[k * m:(k + 1) * m], d_bud[k * m:(k + 1) * m]))
    enc.fence()
    torch.cuda.synchronize()
    ok = True
    for k, (d_out, d_res) in enumerate(outs):
        out, res = d_out.cpu().numpy()[:, :stride], d_res.cpu().numpy()
        wk, wr, bk = want[k * m:(k + 1) * m].copy(), want_res[k * m:(k + 1) * m], budgets[k * m:(k + 1) * m]
        for i in range(m):                          # bytes past a frame's own budget are not the encoder's
            out[i, bk[i]:] = 0; wk[i, bk[i]:] = 0
        ok = ok and np.array_equal(out, wk) and np.array_equal(res, wr)
    enc.close()
    total += m * launches
    bad += 0 if ok else 1
    sc = np.unique(want_res[:m * launches, 0])
    print("round %3d codec %d %dx%d %d x %d frames run %d lanes %d scales %d..%d (%d distinct): %s   [%d frames, %.0f s]"
          % (rnd, codec, w, h, launches, m, run, lanes, sc.min(), sc.max(), sc.size, "ok" if ok else "MISMATCH", total, time.time() - t0), flush=True)
print("soak (mixed content): %d frames in %d rounds, %d mismatching rounds" % (total, rounds, bad))
sys.exit(1 if bad else 0)
