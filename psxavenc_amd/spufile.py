"""SPU / VAG / SPUI / VAGI container framing -- Python mirror of psxhip_spu_file_* (include/psxav_hip.h).

Reference surface: ``encode_file_spu`` / ``encode_file_spui`` and the .vag header writer
(psxavenc/filefmt.c:95-162,212-389) around ``psx_audio_spu_encode`` (libpsxav/adpcm.c:356-376)."""
import ctypes as C

import numpy as np

from . import _lib

FORMAT_SPU, FORMAT_VAG, FORMAT_SPUI, FORMAT_VAGI = 2, 3, 4, 5


class SpuFileSettings(C.Structure):
    """psxhip_spu_file_settings_t; field names follow args_t (psxavenc/args.h)"""
    _fields_ = [("format", C.c_int32), ("audio_frequency", C.c_int32), ("audio_channels", C.c_int32),
                ("audio_interleave", C.c_int32), ("alignment", C.c_int32), ("audio_loop_point", C.c_int32),
                ("enable_loop", C.c_int32), ("no_leading_dummy", C.c_int32), ("name", C.c_char * 16)]


def settings(fmt, channels=None, freq=44100, interleave=2048, alignment=None, loop_point=-1, enable_loop=False,
             no_dummy=False, name="out.vag"):
    """defaults as init_default_args (psxavenc/args.c:149-187)"""
    mono = fmt in (FORMAT_SPU, FORMAT_VAG)
    if channels is None:
        channels = 1 if mono else 2
    if alignment is None:
        alignment = 64 if mono else 2048
    return SpuFileSettings(fmt, freq, channels, interleave, alignment, loop_point, int(enable_loop), int(no_dummy),
                           name.encode()[:16])


def _bind():
    L = _lib.lib()
    L.psxhip_spu_file_size.argtypes = [C.POINTER(SpuFileSettings), C.c_int64]
    L.psxhip_spu_file_size.restype = C.c_int64
    L.psxhip_spu_file_encode_host.argtypes = [C.c_int, C.POINTER(SpuFileSettings), C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t]
    L.psxhip_spu_file_encode_host.restype = C.c_int64
    return L


def encode(s, pcm, device=0):
    """pcm: int16, channels interleaved.  Returns the file's bytes (uint8 array)."""
    L = _bind()
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    n = pcm.size // s.audio_channels
    size = L.psxhip_spu_file_size(C.byref(s), n)
    if size < 0:
        _lib.check(int(size))
    out = np.zeros(size, np.uint8)
    rc = L.psxhip_spu_file_encode_host(device, C.byref(s), pcm.ctypes.data if pcm.size else None, n, out.ctypes.data, out.size)
    if rc < 0:
        _lib.check(int(rc))
    assert rc == size
    return out
