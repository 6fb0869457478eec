import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, oracle_lib as O
from psxavenc_amd.mdec import MdecEncoder
for (w, h, bud) in [(16, 1024, 6000), (1024, 16, 6000), (16, 16, 64), (16, 16, 63), (32, 16, 9000), (1024, 1024, 60000), (640, 512, 16128), (176, 144, 3000)]:
    for codec in (0, 2):
        try:
            fr = O.synth_frames(w, h, 3, seed=5, amp=6)
            want, wres, rc = O.mdec_encode(codec, w, h, fr, bud)
            enc = MdecEncoder(codec, w, h, max_frame_size=bud)
            if rc == 0:
                out, res = enc.encode_frames_host(fr, bud)
                print(w, h, bud, codec, "OK" if (np.array_equal(out, want) and np.array_equal(res, wres)) else "MISMATCH", res[:, 0].tolist())
            else:
                try:
                    enc.encode_frames_host(fr, bud); print(w, h, bud, codec, "expected failure, got success")
                except Exception as e:
                    print(w, h, bud, codec, "no-fit reported by both")
            enc.close()
        except Exception as e:
            print(w, h, bud, codec, "EXC", str(e)[:100])
