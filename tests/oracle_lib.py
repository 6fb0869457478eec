"""ctypes bindings for the CPU checker (oracle/liboracle.so, oracle/_ref/libpsxav_ref.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by psxavenc_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

BS_V2, BS_V3, BS_V3DC = 0, 1, 2
FMT_STR, FMT_STRCD, FMT_STRV = 6, 7, 9


class MdecResult(C.Structure):
    _fields_ = [("quant_scale", C.c_int), ("bytes_used", C.c_int),
                ("blocks_used", C.c_int), ("uncomp_hwords_used", C.c_int)]


class StrState(C.Structure):
    _fields_ = [("frame_index", C.c_int), ("frame_data_offset", C.c_int), ("frame_max_size", C.c_int),
                ("base_overflow", C.c_int), ("overflow_num", C.c_int), ("overflow_den", C.c_int),
                ("bytes_used", C.c_int), ("quant_scale_sum", C.c_int), ("frame_output", C.c_void_p)]


class Chan(C.Structure):
    _fields_ = [("prev1", C.c_int), ("prev2", C.c_int)]


class State(C.Structure):
    _fields_ = [("left", Chan), ("right", Chan)]


class XaSettings(C.Structure):
    _fields_ = [("format", C.c_int), ("stereo", C.c_int), ("frequency", C.c_int),
                ("bits_per_sample", C.c_int), ("file_number", C.c_int), ("channel_number", C.c_int)]


# --- the reference's own structs (libpsxav.h:44-62), for oracle/_ref -------------------
class RefXaSettings(C.Structure):
    _fields_ = [("format", C.c_int), ("stereo", C.c_bool), ("frequency", C.c_int),
                ("bits_per_sample", C.c_int), ("file_number", C.c_int), ("channel_number", C.c_int)]


class RefChan(C.Structure):
    _fields_ = [("qerr", C.c_int), ("mse", C.c_uint64), ("prev1", C.c_int), ("prev2", C.c_int)]


class RefState(C.Structure):
    _fields_ = [("left", RefChan), ("right", RefChan)]


u8p = C.POINTER(C.c_uint8)
i16p = C.POINTER(C.c_int16)
intp = C.POINTER(C.c_int)


def ptr(a, t):
    return a.ctypes.data_as(t)


_lib = None
_ref = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_mdec_encode_frame.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_int, u8p, C.POINTER(MdecResult)]
        L.orc_mdec_encode_frames.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_int, intp, C.c_int, u8p,
                                             C.POINTER(MdecResult)]
        L.orc_mdec_frame_to_coefs.argtypes = [C.c_int, C.c_int, u8p, i16p]
        L.orc_fdct_islow8.argtypes = [i16p]
        L.orc_mdec_ac_code.restype = C.c_uint32
        L.orc_mdec_dc_code.restype = C.c_uint32
        L.orc_mdec_decode_frame.argtypes = [C.c_int, C.c_int, u8p, C.c_int, i16p, intp, intp, intp, C.c_int]
        L.orc_mdec_reconstruct.argtypes = [C.c_int, C.c_int, i16p, C.c_int, u8p]
        L.orc_mdec_encode_sector_str.argtypes = [C.POINTER(StrState), C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_uint16, u8p, u8p]
        L.orc_synth_frame.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, u8p]
        L.orc_synth_pcm.argtypes = [C.c_uint32, C.c_uint32, C.c_int64, C.c_int, C.c_int, i16p]
        L.orc_spu_encode.argtypes = [C.POINTER(Chan), i16p, C.c_int, C.c_int, u8p]
        L.orc_spu_encode_simple.argtypes = [i16p, C.c_int, u8p, C.c_int]
        L.orc_xa_encode.argtypes = [XaSettings, C.POINTER(State), i16p, C.c_int, C.c_int, u8p]
        L.orc_xa_encode_finalize.argtypes = [XaSettings, u8p, C.c_int]
        L.orc_xa_samples_per_sector.argtypes = [XaSettings]
        L.orc_xa_sector_size.argtypes = [XaSettings]
        L.orc_xa_sector_interleave.argtypes = [XaSettings]
        L.orc_edc_crc32.argtypes = [u8p, C.c_int]
        L.orc_edc_crc32.restype = C.c_uint32
        L.orc_cdrom_init_sector.argtypes = [u8p, C.c_int, C.c_int]
        L.orc_cdrom_calculate_checksums.argtypes = [u8p, C.c_int]
        L.orc_scaler_filter.argtypes = [C.c_int, C.c_int, intp, C.POINTER(C.c_int32), i16p, C.c_int]
        L.orc_scaler_convert.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p]
        _lib = L
    return _lib


def ref():
    """The reference's own libpsxav compiled unchanged (None when oracle/_ref is absent)."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libpsxav_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.psx_audio_spu_encode.argtypes = [C.POINTER(RefChan), i16p, C.c_int, C.c_int, u8p]
        R.psx_audio_spu_encode_simple.argtypes = [i16p, C.c_int, u8p, C.c_int]
        R.psx_audio_xa_encode.argtypes = [RefXaSettings, C.POINTER(RefState), i16p, C.c_int, C.c_int, u8p]
        R.psx_audio_xa_encode_simple.argtypes = [RefXaSettings, i16p, C.c_int, C.c_int, u8p]
        R.psx_audio_xa_encode_finalize.argtypes = [RefXaSettings, u8p, C.c_int]
        for f in ("psx_audio_xa_get_buffer_size_per_sector", "psx_audio_xa_get_samples_per_sector",
                  "psx_audio_xa_get_sector_interleave"):
            getattr(R, f).argtypes = [RefXaSettings]
            getattr(R, f).restype = C.c_uint32
        R.psx_audio_xa_get_buffer_size.argtypes = [RefXaSettings, C.c_int]
        R.psx_audio_xa_get_buffer_size.restype = C.c_uint32
        R.psx_audio_spu_get_buffer_size.argtypes = [C.c_int]
        R.psx_audio_spu_get_buffer_size.restype = C.c_uint32
        R.psx_cdrom_init_sector.argtypes = [u8p, C.c_int, C.c_int]
        R.psx_cdrom_calculate_checksums.argtypes = [u8p, C.c_int]
        _ref = R
    return _ref


def ref_settings(s):
    return RefXaSettings(s.format, bool(s.stereo), s.frequency, s.bits_per_sample, s.file_number, s.channel_number)


# ---------------------------------------------------------------- MDEC helpers
def synth_frames(w, h, n, seed=1, amp=4, first=0):
    out = np.empty((n, w * h * 3 // 2), dtype=np.uint8)
    L = lib()
    for i in range(n):
        L.orc_synth_frame(w, h, seed, first + i, amp, ptr(out[i], u8p))
    return out


def synth_pcm(seed, chain, first, n, kind):
    out = np.empty(n, dtype=np.int16)
    lib().orc_synth_pcm(seed, chain, first, n, kind, ptr(out, i16p))
    return out


def mdec_encode(codec, w, h, frames, budgets, stride=None):
    """frames: (n, w*h*3/2) u8; budgets: int or sequence.
    Returns (out (n, stride) u8, results (n, 4) int32 [scale, bytes, blocks, hwords], rc)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n = frames.shape[0]
    budgets = np.full(n, budgets, dtype=np.int32) if np.isscalar(budgets) else np.ascontiguousarray(budgets, dtype=np.int32)
    stride = int(budgets.max()) if stride is None else stride
    out = np.zeros((n, stride), dtype=np.uint8)
    res = (MdecResult * n)()
    rc = lib().orc_mdec_encode_frames(codec, w, h, ptr(frames, u8p), n, ptr(budgets, intp), stride, ptr(out, u8p), res)
    r = np.array([[x.quant_scale, x.bytes_used, x.blocks_used, x.uncomp_hwords_used] for x in res], dtype=np.int32)
    return out, r, rc


def mdec_coefs(w, h, frame):
    nmb = (w // 16) * (h // 16)
    coefs = np.empty((6, nmb, 64), dtype=np.int16)
    lib().orc_mdec_frame_to_coefs(w, h, ptr(np.ascontiguousarray(frame), u8p), ptr(coefs, i16p))
    return coefs


def mdec_decode(w, h, bs, v3dc_wrap=0):
    nmb = (w // 16) * (h // 16)
    bs = np.ascontiguousarray(bs, dtype=np.uint8)
    levels = np.empty((nmb * 6, 64), dtype=np.int16)
    q, v, nb = C.c_int(), C.c_int(), C.c_int()
    rc = lib().orc_mdec_decode_frame(w, h, ptr(bs, u8p), bs.size, ptr(levels, i16p), C.byref(q), C.byref(v),
                                     C.byref(nb), v3dc_wrap)
    return rc, levels, q.value, v.value, nb.value


def mdec_reconstruct(w, h, levels, scale):
    out = np.empty(w * h * 3 // 2, dtype=np.uint8)
    lib().orc_mdec_reconstruct(w, h, ptr(np.ascontiguousarray(levels), i16p), scale, ptr(out, u8p))
    return out


# ---------------------------------------------------------------- ADPCM helpers
def spu_encode(samples, pitch=1, state=None, n=None):
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    n = samples.size // pitch if n is None else n
    st = state if state is not None else Chan(0, 0)
    out = np.zeros(((n + 27) // 28) * 16, dtype=np.uint8)
    ln = lib().orc_spu_encode(C.byref(st), ptr(samples, i16p), n, pitch, ptr(out, u8p))
    return out[:ln], st


def xa_encode(settings, samples, sample_count, lba=0, state=None):
    """Output buffer is zero-initialised first (the reference leaves bytes unwritten)."""
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    st = state if state is not None else State()
    sps = lib().orc_xa_samples_per_sector(settings)
    nsec = max(1, (sample_count + sps - 1) // sps)
    out = np.zeros(nsec * 2352 + 16, dtype=np.uint8)
    ln = lib().orc_xa_encode(settings, C.byref(st), ptr(samples, i16p), sample_count, lba, ptr(out, u8p))
    return out[:ln], st


def ref_spu_encode(samples, pitch=1, state=None, n=None):
    R = ref()
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    n = samples.size // pitch if n is None else n
    st = state if state is not None else RefChan()
    out = np.zeros(((n + 27) // 28) * 16, dtype=np.uint8)
    ln = R.psx_audio_spu_encode(C.byref(st), ptr(samples, i16p), n, pitch, ptr(out, u8p))
    return out[:ln], st


def ref_xa_encode(settings, samples, sample_count, lba=0, state=None):
    R = ref()
    rs = ref_settings(settings)
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    st = state if state is not None else RefState()
    sps = R.psx_audio_xa_get_samples_per_sector(rs)
    nsec = max(1, (sample_count + sps - 1) // sps)
    out = np.zeros(nsec * 2352 + 16, dtype=np.uint8)
    ln = R.psx_audio_xa_encode(rs, C.byref(st), ptr(samples, i16p), sample_count, lba, ptr(out, u8p))
    return out[:ln], st


# ---------------------------------------------------------------- front-end helpers (oracle/frontend_oracle.c)
PIX_RGB24, PIX_YUV420P = 0, 1


def scaler_filter(src, dst):
    """(taps, left (dst,), coef (dst, taps)) of the front-end's bicubic filter bank"""
    left = np.zeros(dst, np.int32)
    coef = np.zeros(dst * 64, np.int16)
    taps = C.c_int()
    rc = lib().orc_scaler_filter(src, dst, C.byref(taps), left.ctypes.data_as(C.POINTER(C.c_int32)), ptr(coef, i16p), coef.size)
    assert rc == 0
    return taps.value, left, coef[:dst * taps.value].reshape(dst, taps.value)


def scaler_convert(fmt, src_w, src_h, full_range, dst_w, dst_h, pictures):
    """pictures: (n, bytes per picture) uint8 -> (n, dst_w * dst_h * 3 / 2) NV21"""
    pictures = np.ascontiguousarray(pictures, dtype=np.uint8)
    out = np.zeros((pictures.shape[0], dst_w * dst_h * 3 // 2), np.uint8)
    for i in range(pictures.shape[0]):
        rc = lib().orc_scaler_convert(fmt, src_w, src_h, int(full_range), dst_w, dst_h, ptr(pictures[i], u8p), ptr(out[i], u8p))
        assert rc == 0, rc
    return out
