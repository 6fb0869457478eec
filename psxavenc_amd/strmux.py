"""STR / STRCD / STRV muxer -- Python mirror of psxhip_str_* (include/psxav_hip.h).

Reference surface: ``encode_file_str`` (psxavenc/filefmt.c:391-520) around ``encode_sector_str``
(psxavenc/mdec.c:757-836) and ``psx_audio_xa_encode`` (libpsxav/adpcm.c:293-332)."""
import ctypes as C

import numpy as np

from . import _lib

FORMAT_STR, FORMAT_STRCD, FORMAT_STRV = 6, 7, 9


class StrSettings(C.Structure):
    """psxhip_str_settings_t; field names follow args_t (psxavenc/args.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "format", "video_codec", "video_width", "video_height", "str_fps_num", "str_fps_den", "str_cd_speed",
        "str_video_id", "trailing_audio", "audio_channels", "audio_frequency", "audio_bit_depth", "audio_xa_file",
        "audio_xa_channel")]


class StrPlan(C.Structure):
    _fields_ = [("n_sectors", C.c_int32), ("n_video_sectors", C.c_int32), ("n_audio_sectors", C.c_int32),
                ("sector_size", C.c_int32), ("interleave", C.c_int32), ("audio_samples_per_sector", C.c_int32),
                ("max_frame_size", C.c_int32), ("reserved", C.c_int32), ("quant_scale_sum", C.c_int64)]


def settings(fmt=FORMAT_STRCD, codec=0, width=320, height=240, fps_num=15, fps_den=1, cd_speed=2, video_id=0x8001,
             trailing_audio=False, channels=2, frequency=37800, bits=4, xa_file=1, xa_channel=0):
    """defaults = config 'strcd v2' (args.c:149-187 + SURVEY 3.2)"""
    return StrSettings(fmt, codec, width, height, fps_num, fps_den, cd_speed, video_id, int(trailing_audio), channels,
                       frequency, bits, xa_file, xa_channel)


def _bind():
    L = _lib.lib()
    L.psxhip_str_plan.argtypes = [C.POINTER(StrSettings), C.c_int, C.POINTER(StrPlan)]
    L.psxhip_str_frame_budgets.argtypes = [C.POINTER(StrSettings), C.c_int, C.c_int, C.c_void_p]
    L.psxhip_str_encode_host.argtypes = [C.c_int, C.POINTER(StrSettings), C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_size_t, C.POINTER(StrPlan)]
    return L


def plan(s, n_frames):
    p = StrPlan()
    _lib.check(_bind().psxhip_str_plan(C.byref(s), n_frames, C.byref(p)))
    return p


def frame_budgets(s, first_frame, n_frames):
    out = np.zeros(n_frames, np.int32)
    _lib.check(_bind().psxhip_str_frame_budgets(C.byref(s), first_frame, n_frames, out.ctypes.data))
    return out


def encode(s, frames, pcm=None, device=0, out=None):
    """frames: (n, w*h*3/2) uint8; pcm: int16, interleaved when stereo.  Returns (sectors (n_sectors, sector_size) uint8, plan).
    `out`: optional preallocated (n_sectors, sector_size) uint8 array to write into (a caller muxing stream after stream
    reuses its buffer instead of faulting in 23 MB of fresh pages per call)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n = frames.shape[0]
    p = plan(s, n)
    if out is None:
        out = np.zeros((p.n_sectors, p.sector_size), np.uint8)
    assert out.dtype == np.uint8 and out.flags.c_contiguous and out.shape == (p.n_sectors, p.sector_size)
    if pcm is None:
        pcm = np.zeros(0, np.int16)
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    per_ch = pcm.size // max(1, s.audio_channels)
    rc = _bind().psxhip_str_encode_host(device, C.byref(s), frames.ctypes.data, n, pcm.ctypes.data if pcm.size else None,
                                        per_ch, out.ctypes.data, out.size, C.byref(p))
    _lib.check(rc)
    return out, p
