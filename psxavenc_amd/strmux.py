"""STR / STRCD / STRV muxer -- Python mirror of psxhip_str_* (include/psxav_hip.h).

Reference surface: ``encode_file_str`` (psxavenc/filefmt.c:391-520) around ``encode_sector_str``
(psxavenc/mdec.c:757-836) and ``psx_audio_xa_encode`` (libpsxav/adpcm.c:293-332)."""
import ctypes as C

import numpy as np

from . import _lib

FORMAT_STR, FORMAT_STRCD, FORMAT_STRV = 6, 7, 9
TAIL_REFERENCE, TAIL_COMPLETE = 0, 1      # PSXHIP_STR_TAIL_*: how the stream ends (see include/psxav_hip.h)
PLENTY_OF_AUDIO = 1 << 40                 # plan(): "the audio never ends before the video does"


class StrSettings(C.Structure):
    """psxhip_str_settings_t; field names follow args_t (psxavenc/args.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "format", "video_codec", "video_width", "video_height", "str_fps_num", "str_fps_den", "str_cd_speed",
        "str_video_id", "trailing_audio", "audio_channels", "audio_frequency", "audio_bit_depth", "audio_xa_file",
        "audio_xa_channel", "tail_mode", "reserved")]


class StrPlan(C.Structure):
    _fields_ = [("n_sectors", C.c_int32), ("n_video_sectors", C.c_int32), ("n_audio_sectors", C.c_int32),
                ("sector_size", C.c_int32), ("interleave", C.c_int32), ("audio_samples_per_sector", C.c_int32),
                ("max_frame_size", C.c_int32), ("n_frames_encoded", C.c_int32), ("quant_scale_sum", C.c_int64)]


def settings(fmt=FORMAT_STRCD, codec=0, width=320, height=240, fps_num=15, fps_den=1, cd_speed=2, video_id=0x8001,
             trailing_audio=False, channels=2, frequency=37800, bits=4, xa_file=1, xa_channel=0, tail=TAIL_REFERENCE):
    """defaults = config 'strcd v2' (args.c:149-187 + SURVEY 3.2); tail = the reference's end-of-input model"""
    return StrSettings(fmt, codec, width, height, fps_num, fps_den, cd_speed, video_id, int(trailing_audio), channels,
                       frequency, bits, xa_file, xa_channel, tail, 0)


def _bind():
    L = _lib.lib()
    L.psxhip_str_plan.argtypes = [C.POINTER(StrSettings), C.c_int, C.c_int64, C.POINTER(StrPlan)]
    L.psxhip_str_frame_budgets.argtypes = [C.POINTER(StrSettings), C.c_int, C.c_int, C.c_void_p]
    L.psxhip_str_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]
    L.psxhip_str_destroy.argtypes = [C.c_void_p]
    L.psxhip_str_destroy.restype = None
    L.psxhip_str_encode_host.argtypes = [C.c_void_p, C.POINTER(StrSettings), C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_size_t, C.POINTER(StrPlan)]
    L.psxhip_str_encode_device.argtypes = [C.c_void_p, C.POINTER(StrSettings), C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                           C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.POINTER(StrPlan), C.c_void_p]
    return L


def plan(s, n_frames, pcm_samples_per_channel=PLENTY_OF_AUDIO):
    p = StrPlan()
    _lib.check(_bind().psxhip_str_plan(C.byref(s), n_frames, pcm_samples_per_channel, C.byref(p)))
    return p


class StrSector(C.Structure):
    """psxhip_str_sector_t"""
    _fields_ = [("kind", C.c_int32), ("frame", C.c_int32), ("index", C.c_int32), ("eof", C.c_int32)]


SECTOR_VIDEO, SECTOR_AUDIO, SECTOR_EMPTY = 0, 1, 2


def plan_sectors(s, n_frames, pcm_samples_per_channel=PLENTY_OF_AUDIO):
    """what each sector of the stream holds: (n_sectors, 4) int32 [kind, frame, index, eof]"""
    L = _bind()
    L.psxhip_str_plan_sectors.argtypes = [C.POINTER(StrSettings), C.c_int, C.c_int64, C.c_void_p, C.c_int]
    n = L.psxhip_str_plan_sectors(C.byref(s), n_frames, pcm_samples_per_channel, None, 0)
    if n < 0:
        _lib.check(n)
    out = np.zeros((n, 4), np.int32)
    if n:
        n2 = L.psxhip_str_plan_sectors(C.byref(s), n_frames, pcm_samples_per_channel, out.ctypes.data, n)
        assert n2 == n
    return out


def frame_budgets(s, first_frame, n_frames):
    out = np.zeros(n_frames, np.int32)
    _lib.check(_bind().psxhip_str_frame_budgets(C.byref(s), first_frame, n_frames, out.ctypes.data))
    return out


class StrMuxer:
    """psxhip_str_ctx_t: owns what is kept between calls (encoder contexts of the listed devices, pinned buffers)."""

    def __init__(self, devices=(0,)):
        self.devices = tuple(int(d) for d in devices)
        self._h = C.c_void_p()
        arr = (C.c_int * len(self.devices))(*self.devices)
        _lib.check(_bind().psxhip_str_create(C.byref(self._h), arr, len(self.devices)))

    def close(self):
        if self._h:
            _bind().psxhip_str_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, s, frames, pcm=None, out=None):
        """frames: (n, w*h*3/2) uint8; pcm: int16, interleaved when stereo.  Returns (sectors (n_sectors, sector_size) uint8, plan).
        `out`: optional preallocated (n_sectors, sector_size) uint8 array to write into (a caller muxing stream after stream
        reuses its buffer instead of faulting in 23 MB of fresh pages per call)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n = frames.shape[0]
        if pcm is None:
            pcm = np.zeros(0, np.int16)
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        per_ch = pcm.size // max(1, s.audio_channels)
        p = plan(s, n, per_ch)
        if out is None:
            out = np.zeros((p.n_sectors, p.sector_size), np.uint8)
        assert out.dtype == np.uint8 and out.flags.c_contiguous and out.shape == (p.n_sectors, p.sector_size)
        rc = _bind().psxhip_str_encode_host(self._h, C.byref(s), frames.ctypes.data, n, pcm.ctypes.data if pcm.size else None,
                                            per_ch, out.ctypes.data, out.size, C.byref(p))
        _lib.check(rc)
        return out, p


    def encode_device(self, s, d_frames, d_pcm=None, d_out=None, stream=None):
        """psxhip_str_encode_device: everything stays in HBM.  d_frames: uint8 CUDA tensor (S, n_frames, w*h*3/2) (or (n_frames, ..) for
        one stream); d_pcm: int16 CUDA tensor (S, samples_per_channel * channels), interleaved when stereo, or None.
        Returns (d_out (S, n_sectors, sector_size) uint8 CUDA tensor, plan); the call returns when the sectors are complete."""
        import torch
        if d_frames.dim() == 2:
            d_frames = d_frames.unsqueeze(0)
        assert d_frames.is_cuda and d_frames.dtype == torch.uint8 and d_frames.is_contiguous()
        n_streams, n = d_frames.shape[0], d_frames.shape[1]
        ch = max(1, s.audio_channels)
        per_ch = 0
        if d_pcm is not None:
            if d_pcm.dim() == 1:
                d_pcm = d_pcm.unsqueeze(0)
            assert d_pcm.is_cuda and d_pcm.dtype == torch.int16 and d_pcm.is_contiguous() and d_pcm.shape[0] == n_streams
            per_ch = d_pcm.shape[1] // ch if s.audio_channels else 0
        p = plan(s, n, per_ch)
        if d_out is None:
            d_out = torch.zeros((n_streams, p.n_sectors, p.sector_size), dtype=torch.uint8, device=d_frames.device)
        assert d_out.is_cuda and d_out.dtype == torch.uint8 and d_out.is_contiguous() and tuple(d_out.shape) == (n_streams, p.n_sectors, p.sector_size)
        st = stream if stream is not None else torch.cuda.current_stream(d_frames.device)
        rc = _bind().psxhip_str_encode_device(self._h, C.byref(s), n_streams, d_frames.data_ptr(), d_frames.stride(0), n,
                                              d_pcm.data_ptr() if d_pcm is not None and d_pcm.numel() else None,
                                              d_pcm.stride(0) if d_pcm is not None else 0, per_ch, d_out.data_ptr(),
                                              d_out.stride(0) if p.n_sectors else 4, C.byref(p), st.cuda_stream)
        _lib.check(rc)
        return d_out, p


_muxers = {}


def encode(s, frames, pcm=None, device=0, out=None, devices=None):
    """convenience for tests / bench: one cached StrMuxer per device list"""
    key = tuple(devices) if devices is not None else (int(device),)
    if key not in _muxers:
        _muxers[key] = StrMuxer(key)
    return _muxers[key].encode(s, frames, pcm, out=out)


def release():
    for m in _muxers.values():
        m.close()
    _muxers.clear()
