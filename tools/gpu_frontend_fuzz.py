"""Seeded fuzz of the colour-conversion / scaling front-end against its CPU statement: random source sizes (odd RGB widths ->
the unaligned staging path, tiny and huge ratios -> every tile shape), targets, formats and ranges.
usage: python tools/gpu_frontend_fuzz.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from psxavenc_amd import _lib
from psxavenc_amd.frontend import Scaler

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
bad = done = refused = 0
t0 = time.time()
for case in range(n_cases):
    fmt = int(rng.integers(0, 2))
    dw, dh = 16 * int(rng.integers(1, 41)), 16 * int(rng.integers(1, 31))
    ratio_w, ratio_h = float(rng.choice([0.3, 0.5, 0.77, 1.0, 1.25, 1.5, 2.0, 2.6, 3.0, 4.0, 6.5])), float(rng.choice([0.3, 0.5, 0.77, 1.0, 1.25, 1.5, 2.0, 2.6, 3.0, 4.0, 6.5]))
    sw, sh = max(2, int(dw * ratio_w) + int(rng.integers(-3, 4))), max(2, int(dh * ratio_h) + int(rng.integers(-3, 4)))
    if fmt == 1:
        sw, sh = sw & ~1, sh & ~1
        sw, sh = max(2, sw), max(2, sh)
    full = bool(rng.integers(0, 2)) if fmt == 1 else True
    if sw * sh > 2500 * 1600:
        continue
    try:
        sc = Scaler(fmt, sw, sh, dw, dh, src_full_range=full)
    except _lib.PsxHipError:
        refused += 1
        continue
    nbytes = sw * sh * 3 if fmt == 0 else sw * sh * 3 // 2
    pics = rng.integers(0, 256, (2, nbytes), dtype=np.uint8)
    # smooth half of the picture so that the filters see structure as well as noise
    pics[0, : nbytes // 2] = (np.arange(nbytes // 2) // 7 % 256).astype(np.uint8)
    # vertical segments per band: the launch's own choice for two pictures is close to one per tile; one segment walks the whole
    # picture through the ring of intermediates
    segs = [None, "1", "2", "3"][case % 4]
    if segs is None:
        os.environ.pop("PSXHIP_SCALER_VSEGS", None)
    else:
        os.environ["PSXHIP_SCALER_VSEGS"] = segs
    got = sc.convert_host(pics)
    want = O.scaler_convert(fmt, sw, sh, full, dw, dh, pics)
    ok = np.array_equal(got, want)
    done += 1
    bad += 0 if ok else 1
    print("case %3d %s %4dx%-4d -> %4dx%-4d %s %s" % (case, "rgb" if fmt == 0 else "yuv", sw, sh, dw, dh, ("full" if full else "limited") + " segs " + str(segs), "ok" if ok else "MISMATCH (%d bytes)" % int((got != want).sum())), flush=True)
    sc.close()
print("front-end fuzz: %d cases compared, %d refused geometries, %d mismatching, %.0f s" % (done, refused, bad, time.time() - t0))
sys.exit(1 if bad else 0)
