"""Synthetic inputs generated directly in HBM (psxhip_synth_*): the device twin of oracle/synth.c."""
import ctypes as C

import torch

from . import _lib


def _bind():
    L = _lib.lib()
    L.psxhip_synth_frames_device.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                             C.c_int, C.c_int, C.c_void_p]
    L.psxhip_synth_pcm_device.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64, C.c_int,
                                          C.c_int, C.c_void_p]
    return L


def frames_device(w, h, seed, first, n, amp, device=0, out=None):
    """(n, w*h*3/2) uint8 NV21 frames first..first+n-1 on cuda:<device>."""
    L = _bind()
    dev = torch.device("cuda", device)
    if out is None:
        out = torch.empty((n, w * h * 3 // 2), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    _lib.check(L.psxhip_synth_frames_device(device, out.data_ptr(), out.stride(0), w, h, seed, first, n, amp, st.cuda_stream))
    return out


def pcm_device(seed, chain, first, n, kind, device=0, out=None, pitch=1):
    L = _bind()
    dev = torch.device("cuda", device)
    if out is None:
        out = torch.zeros(n * pitch, dtype=torch.int16, device=dev)
    st = torch.cuda.current_stream(dev)
    _lib.check(L.psxhip_synth_pcm_device(device, out.data_ptr(), seed, chain, first, n, kind, pitch, st.cuda_stream))
    return out
