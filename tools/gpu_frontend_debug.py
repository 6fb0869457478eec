import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from psxavenc_amd.frontend import Scaler
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_frontend.py"))
T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
fmt, sw, sh, dw, dh = [int(x) for x in sys.argv[1:6]]
pics = T._pictures(fmt, sw, sh, 1, seed=sw + dh)
sc = Scaler(fmt, sw, sh, dw, dh)
got = sc.convert_host(pics)[0]; want = O.scaler_convert(fmt, sw, sh, True, dw, dh, pics)[0]
ly = (got[:dw * dh] != want[:dw * dh]).reshape(dh, dw)
print("luma bad rows:", np.nonzero(ly.any(axis=1))[0][:40], "bad cols:", np.nonzero(ly.any(axis=0))[0][:40], ly.sum())
c = (got[dw * dh:] != want[dw * dh:]).reshape(dh // 2, dw)
print("chroma bad rows:", np.nonzero(c.any(axis=1))[0][:40], "bad cols:", np.nonzero(c.any(axis=0))[0][:40], c.sum())
for w_ in range(4):
    print(w_, sc.filter(w_)[0], sc.filter(w_)[1][:20])
g = got[:dw * dh].reshape(dh, dw); w = want[:dw * dh].reshape(dh, dw)
for r in (12, 13, 14, 15, 29):
    print(r, "got ", g[r, :16].tolist()); print(r, "want", w[r, :16].tolist())
