"""Ad-hoc GPU check: HIP MDEC path vs oracle on a few configs (development aid)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from psxavenc_amd.mdec import MdecEncoder

bad = 0
for (w, h, bud, amp, codec, n) in [(16, 16, 4096, 4, 0, 3), (48, 32, 4096, 8, 0, 3), (320, 240, 8192, 4, 0, 8), (320, 240, 8192, 8, 0, 8),
                                   (320, 240, 8192, 4, 1, 8), (320, 240, 8191, 8, 2, 8), (640, 480, 32768, 8, 1, 4),
                                   (640, 480, 8192, 4, 1, 4), (320, 240, 3000, 8, 0, 4)]:
    fr = O.synth_frames(w, h, n, seed=1, amp=amp)
    ref, rres, rc = O.mdec_encode(codec, w, h, fr, bud)
    enc = MdecEncoder(codec, w, h, max_frame_size=max(bud, 8192))
    t = time.time()
    try:
        out, res = enc.encode_frames_host(fr, bud)
    except Exception as e:
        print(w, h, bud, amp, codec, "EXC", e, "oracle rc", rc); continue
    dt = time.time() - t
    ok = (out == ref).all() and (res == rres).all()
    print(w, h, bud, amp, codec, "OK" if ok else "MISMATCH", res[:3].tolist(), rres[:3].tolist(), "%.1f ms" % (dt * 1e3))
    if not ok:
        bad += 1
        for k in range(n):
            d = np.nonzero(out[k] != ref[k])[0]
            if d.size: print("   frame", k, "first diff at byte", d[0], "ndiff", d.size, out[k][d[0]:d[0]+8], ref[k][d[0]:d[0]+8])
    enc.close()
print("BAD" if bad else "ALL OK")
