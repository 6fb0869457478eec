"""Colour conversion + scaling front-end -- Python mirror of psxhip_scaler_* (include/psxav_hip.h).

Reference surface: the libswscale context psxavenc configures and drives (psxavenc/decoding.c:287-311,463-475): decoded
pictures -> NV21 frames of the encoder's size, BT.601 full range.  Parity with libswscale itself is unpinned (FFmpeg is
absent); the arithmetic is specified in DESIGN.md section 9 and checked against oracle/frontend_oracle.c."""
import ctypes as C

import numpy as np

from . import _lib

PIX_RGB24, PIX_YUV420P = 0, 1


def _bind():
    L = _lib.lib()
    L.psxhip_scaler_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.psxhip_scaler_destroy.argtypes = [C.c_void_p]
    L.psxhip_scaler_destroy.restype = None
    L.psxhip_scaler_source_bytes.argtypes = [C.c_void_p]
    L.psxhip_scaler_source_bytes.restype = C.c_size_t
    L.psxhip_scaler_convert_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.psxhip_scaler_convert_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.psxhip_scaler_filter.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int]
    return L


class Scaler:
    """psxhip_scaler_t"""

    def __init__(self, src_format, src_width, src_height, dst_width, dst_height, src_full_range=True, device=0):
        self._h = C.c_void_p()
        self.dst_width, self.dst_height, self.device = dst_width, dst_height, device
        _lib.check(_bind().psxhip_scaler_create(C.byref(self._h), device, src_format, src_width, src_height, int(bool(src_full_range)),
                                                dst_width, dst_height))
        self.source_bytes = int(_bind().psxhip_scaler_source_bytes(self._h))
        self.frame_bytes = dst_width * dst_height * 3 // 2

    def close(self):
        if self._h:
            _bind().psxhip_scaler_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def filter(self, which):
        """(taps, left (n,), coef (n, taps)); which: 0 luma h, 1 luma v, 2 chroma h, 3 chroma v"""
        L = _bind()
        taps = C.c_int()
        n = L.psxhip_scaler_filter(self._h, which, C.byref(taps), None, None, 0)
        left = np.zeros(n, np.int32)
        coef = np.zeros((n, taps.value), np.int16)
        assert L.psxhip_scaler_filter(self._h, which, C.byref(taps), left.ctypes.data, coef.ctypes.data, coef.size) == n
        return taps.value, left, coef

    def convert_host(self, pictures):
        pictures = np.ascontiguousarray(pictures, dtype=np.uint8)
        n = pictures.shape[0]
        assert pictures.shape[1] == self.source_bytes
        out = np.zeros((n, self.frame_bytes), np.uint8)
        _lib.check(_bind().psxhip_scaler_convert_host(self._h, pictures.ctypes.data, n, out.ctypes.data))
        return out

    def convert_device(self, d_pictures, d_frames=None, stream=None):
        """d_pictures: uint8 CUDA tensor (n, >= source_bytes) -> NV21 frames (n, frame_bytes) on the same device, asynchronous"""
        import torch
        assert d_pictures.is_cuda and d_pictures.dtype == torch.uint8 and d_pictures.dim() == 2
        n = d_pictures.shape[0]
        if d_frames is None:
            d_frames = torch.empty((n, self.frame_bytes), dtype=torch.uint8, device=d_pictures.device)
        st = stream if stream is not None else torch.cuda.current_stream(d_pictures.device)
        _lib.check(_bind().psxhip_scaler_convert_device(self._h, d_pictures.data_ptr(), d_pictures.stride(0), n, d_frames.data_ptr(),
                                                        d_frames.stride(0), st.cuda_stream))
        return d_frames
