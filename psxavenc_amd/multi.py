"""Several devices behind one call -- Python mirror of the psxhip_*_multi entry points (include/psxav_hip.h).

The reference's host side is one C loop (psxavenc/filefmt.c:633-662); these take a device list and shard inside the
call: contiguous ranges (``shard_range``) or a host-side ticket queue of frame ranges."""
import ctypes as C

import numpy as np

from . import _lib

SCHED_STATIC, SCHED_TICKETS = 0, 1


class MultiReport(C.Structure):
    """psxhip_multi_report_t"""
    _fields_ = [("device", C.c_int32), ("units", C.c_int64), ("tickets", C.c_int32), ("seconds", C.c_double)]


def _bind():
    L = _lib.lib()
    i64p = C.POINTER(C.c_int64)
    L.psxhip_shard_range.argtypes = [C.c_int64, C.c_int, C.c_int, i64p, i64p]
    L.psxhip_shard_range.restype = None
    L.psxhip_ticket_queue_create.argtypes = [C.c_int64, C.c_int64]
    L.psxhip_ticket_queue_create.restype = C.c_void_p
    L.psxhip_ticket_queue_next.argtypes = [C.c_void_p, i64p, i64p]
    L.psxhip_ticket_queue_destroy.argtypes = [C.c_void_p]
    L.psxhip_ticket_queue_destroy.restype = None
    L.psxhip_mdec_multi_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.psxhip_mdec_multi_destroy.argtypes = [C.c_void_p]
    L.psxhip_mdec_multi_destroy.restype = None
    L.psxhip_mdec_multi_device_count.argtypes = [C.c_void_p]
    L.psxhip_mdec_multi_encode_frames_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                                       C.c_void_p, C.c_int, C.c_int, C.POINTER(MultiReport)]
    L.psxhip_xa_encode_streams_host_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                      C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_int64, C.c_int, C.POINTER(MultiReport)]
    return L


def shard_range(n_units, rank, world):
    f, c = C.c_int64(), C.c_int64()
    _bind().psxhip_shard_range(n_units, rank, world, C.byref(f), C.byref(c))
    return f.value, c.value


class TicketQueue:
    """psxhip_ticket_queue_t: [0, n_units) in ranges of ticket_units, each handed out exactly once, to any number of threads"""

    def __init__(self, n_units, ticket_units):
        self._q = _bind().psxhip_ticket_queue_create(n_units, ticket_units)
        if not self._q:
            raise ValueError("bad ticket queue geometry")

    def next(self):
        f, c = C.c_int64(), C.c_int64()
        if not _bind().psxhip_ticket_queue_next(self._q, C.byref(f), C.byref(c)):
            return None
        return f.value, c.value

    def close(self):
        if self._q:
            _bind().psxhip_ticket_queue_destroy(self._q)
            self._q = None

    def __del__(self):
        self.close()


def _reports(rep, n):
    return [{"device": r.device, "units": r.units, "tickets": r.tickets, "seconds": r.seconds} for r in rep[:n]]


class MdecMulti:
    """psxhip_mdec_multi_t: one encoder context per listed device (a device may be listed twice)."""

    def __init__(self, devices, video_codec, video_width, video_height, max_frame_size=65536):
        self.devices = tuple(int(d) for d in devices)
        self.video_width, self.video_height = video_width, video_height
        self._h = C.c_void_p()
        arr = (C.c_int * len(self.devices))(*self.devices)
        _lib.check(_bind().psxhip_mdec_multi_create(C.byref(self._h), arr, len(self.devices), video_codec, video_width, video_height,
                                                    max_frame_size))
        self.last_report = None

    def close(self):
        if self._h:
            _bind().psxhip_mdec_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_frames_host(self, frames, frame_max_sizes, schedule=SCHED_STATIC, ticket_frames=0, out=None, res=None):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n = frames.shape[0]
        assert frames.shape[1] == self.video_width * self.video_height * 3 // 2
        if np.isscalar(frame_max_sizes):
            sizes_p, uniform, stride = None, int(frame_max_sizes), int(frame_max_sizes)
        else:
            sizes = np.ascontiguousarray(frame_max_sizes, dtype=np.int32)
            sizes_p, uniform, stride = sizes.ctypes.data, 0, int(sizes.max())
        if out is None:
            out = np.zeros((n, stride), dtype=np.uint8)
        if res is None:
            res = np.zeros((n, 4), dtype=np.int32)
        rep = (MultiReport * len(self.devices))()
        rc = _bind().psxhip_mdec_multi_encode_frames_host(self._h, frames.ctypes.data, n, sizes_p, uniform, out.ctypes.data,
                                                          out.strides[0], res.ctypes.data, schedule, ticket_frames, rep)
        self.last_report = _reports(rep, len(self.devices))
        _lib.check(rc)
        return out, res


def xa_encode_streams_multi(devices, settings, pcm, lbas=None, states=None, finalize=False):
    """pcm: (n_streams, samples * channels) int16.  settings: psxavenc_amd.adpcm.XaSettings.  Returns (sectors, states, report)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    n, per = pcm.shape
    ch = 2 if settings.stereo else 1
    spc = per // ch
    L = _bind()
    from . import adpcm
    bytes_per = adpcm.xa_get_buffer_size(settings, spc)
    out = np.zeros((n, bytes_per), np.uint8)
    if states is None:
        states = np.zeros((n * ch, 2), np.int32)
    states = np.ascontiguousarray(states, dtype=np.int32)
    lb = None if lbas is None else np.ascontiguousarray(lbas, dtype=np.int32)
    arr = (C.c_int * len(devices))(*[int(d) for d in devices])
    rep = (MultiReport * len(devices))()
    rc = L.psxhip_xa_encode_streams_host_multi(arr, len(devices), settings.format, int(bool(settings.stereo)), settings.frequency,
                                               settings.bits_per_sample, settings.file_number, settings.channel_number, pcm.ctypes.data, n,
                                               per, spc, None if lb is None else lb.ctypes.data, states.ctypes.data, out.ctypes.data,
                                               bytes_per, int(finalize), rep)
    if rc < 0:
        _lib.check(rc)
    return out, states, _reports(rep, min(len(devices), max(n, 1)))
