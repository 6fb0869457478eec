#!/usr/bin/env python3
"""CPU study tool (uses the oracle, never the product): AC bits(scale) and the refinement lower bound of synthetic
frames, fed through the search-policy simulator (tests/cpu/search_sim.cpp)."""
import ctypes as C
import os
import re
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O


def luts():
    src = open(os.path.join(ROOT, "psxavenc_amd/csrc/bs_vlc_lut.h")).read()
    m = re.search(r"bs_ac_len16_lut\[\d+\] = \{(.*?)\};", src, re.S)
    v = np.array([int(x, 16) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]).reshape(42, 63)
    q = re.search(r"bs_quant_zz\[64\] = \{(.*?)\};", src, re.S)
    quant = np.array([int(x) for x in q.group(1).replace("\n", " ").split(",") if x.strip()])
    z = re.search(r"bs_zagzig\[64\] = \{(.*?)\};", src, re.S)
    zz = np.array([int(x) for x in z.group(1).replace("\n", " ").split(",") if x.strip()])
    return v & 0xFF, v >> 8, quant, zz


def curves(w, h, frame, scales=range(1, 64)):
    """AC bits and deficits per scale for one frame: returns (tb_ac[64], def[64])"""
    lens, defs, quant, zz = luts()
    co = O.mdec_coefs(w, h, frame).reshape(-1, 64)[:, zz].astype(np.int64)     # blocks x 64, zig-zag order
    co[:, 0] = 0
    tb = np.zeros(64, np.int64)
    df = np.zeros(64, np.int64)
    a = np.abs(co)
    for s in scales:
        d = quant * s
        q = (2 * a + d) // (2 * d)
        nz = q != 0
        nz[:, 0] = False
        # run before each nonzero: position - previous nonzero position - 1 (previous = 0 for the DC slot)
        pos = np.where(nz, np.arange(64)[None, :], 0)
        prev = np.maximum.accumulate(pos, axis=1)
        prevpos = np.concatenate([np.zeros((co.shape[0], 1), np.int64), prev[:, :-1]], axis=1)
        run = np.arange(64)[None, :] - prevpos - 1
        lv = np.minimum(q, 41)
        tb[s] = lens[lv[nz], run[nz]].sum()
        df[s] = defs[lv[nz], run[nz]].sum()
    return tb, df


if __name__ == "__main__":
    w, h, budget, amp, n = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else [640, 480, 8192, 4, 8])]
    so = "/tmp/libsearch_sim.so"
    subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests/cpu/search_sim.cpp")], check=True)
    L = C.CDLL(so)
    ip = C.POINTER(C.c_int)
    nblk = (w // 16) * (h // 16) * 6
    fixed = 12 * nblk + 10
    limit = 16 * ((budget - 8) // 2)
    fr = O.synth_frames(w, h, n, seed=1, amp=amp)
    for i in range(n):
        tb, df = curves(w, h, fr[i], range(1, 40))
        tb[40:] = tb[39]
        t = (tb + fixed).astype(np.int32)
        t[0] = 0
        f = (t - df).astype(np.int32)
        want = next((s for s in range(1, 64) if t[s] <= limit), 64)
        out = []
        for g in range(1, 24):
            npass, lo, hi = C.c_int(), C.c_int(), C.c_int()
            r = L.search_sim(t.ctypes.data_as(ip), f.ctypes.data_as(ip), limit, fixed, g, limit + 32 * nblk // 6, C.byref(npass), C.byref(lo), C.byref(hi))
            assert r == want, (r, want)
            out.append(npass.value)
        print("frame %d want %d  limit %d fixed %d  tb[want-2..want+1]=%s  passes by guess 1..23: %s" % (i, want, limit, fixed, t[max(1, want - 2):want + 2].tolist(), out))
