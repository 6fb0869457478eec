"""Per-call rates of the audio drop-ins next to the CPU they replace: the reference calls
psx_audio_spu_encode once per 28 samples (filefmt.c:243) and psx_audio_xa_encode once per sector (:184); each such call
is a synchronous H2D + launch + D2H here."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from psxavenc_amd import _lib

L = _lib.lib()


class Chan(C.Structure):
    _fields_ = [("qerr", C.c_int), ("mse", C.c_uint64), ("prev1", C.c_int), ("prev2", C.c_int)]


class State(C.Structure):
    _fields_ = [("left", Chan), ("right", Chan)]


class XaSettings(C.Structure):
    _fields_ = [("format", C.c_int), ("stereo", C.c_bool), ("frequency", C.c_int), ("bits_per_sample", C.c_int),
                ("file_number", C.c_int), ("channel_number", C.c_int)]


L.psx_audio_spu_encode.argtypes = [C.POINTER(Chan), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.psx_audio_xa_encode.argtypes = [XaSettings, C.POINTER(State), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
R = O.ref()
out = {}
pcm = O.synth_pcm(1, 0, 0, 28 * 4000, 0)
blk = np.zeros(16, np.uint8)
for name, lib in (("gpu_dropin", L), ("cpu_reference", R)):
    if lib is None:
        continue
    st = Chan() if lib is L else O.RefChan()
    fn = lib.psx_audio_spu_encode
    for k in range(50):
        fn(C.byref(st), pcm[k * 28:].ctypes.data_as(O.i16p) if lib is R else pcm[k * 28:].ctypes.data, 28, 1, blk.ctypes.data_as(O.u8p) if lib is R else blk.ctypes.data)
    n = 3000
    t = time.perf_counter()
    for k in range(n):
        fn(C.byref(st), pcm[k * 28:].ctypes.data_as(O.i16p) if lib is R else pcm[k * 28:].ctypes.data, 28, 1, blk.ctypes.data_as(O.u8p) if lib is R else blk.ctypes.data)
    dt = time.perf_counter() - t
    out["spu_28_samples_per_call_" + name] = {"blocks_per_sec": round(n / dt, 1), "us_per_call": round(dt / n * 1e6, 2)}
x = np.zeros((2016 * 400 + 4032) * 2, np.int16)
x[0:2 * 2016 * 400:2] = O.synth_pcm(1, 0, 0, 2016 * 400, 0)
x[1:2 * 2016 * 400:2] = O.synth_pcm(1, 1, 0, 2016 * 400, 0)
sec = np.zeros(2352, np.uint8)
for name, lib in (("gpu_dropin", L), ("cpu_reference", R)):
    if lib is None:
        continue
    s = XaSettings(1, True, 37800, 4, 1, 0) if lib is L else O.RefXaSettings(1, True, 37800, 4, 1, 0)
    st = State() if lib is L else O.RefState()
    fn = lib.psx_audio_xa_encode
    arg = (lambda k: x[k * 4032:].ctypes.data) if lib is L else (lambda k: x[k * 4032:].ctypes.data_as(O.i16p))
    o = sec.ctypes.data if lib is L else sec.ctypes.data_as(O.u8p)
    for k in range(20):
        fn(s, C.byref(st), arg(k), 2016, k, o)
    n = 300
    t = time.perf_counter()
    for k in range(n):
        fn(s, C.byref(st), arg(k), 2016, k, o)
    dt = time.perf_counter() - t
    out["xa_sector_per_call_" + name] = {"sectors_per_sec": round(n / dt, 1), "us_per_call": round(dt / n * 1e6, 2)}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_percall_audio.json"), "w"), indent=1)
