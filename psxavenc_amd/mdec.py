"""MDEC BS frame encoder -- Python mirror of include/psxav_mdec.h / psxav_hip.h.

Reference surface: ``init_mdec_encoder / encode_frame_bs / destroy_mdec_encoder``
(psxavenc/mdec.h:65-74).  ``MdecEncoder`` keeps the reference's state fields
(frame_max_size, quant_scale, bytes_used, blocks_used, uncomp_hwords_used, quant_scale_sum,
frame_index) and adds the batched, device-resident entry point used for throughput.
"""
import ctypes as C

import numpy as np

from . import _lib

try:  # torch is plumbing (device memory, streams); the host-buffer path works without it
    import torch
except Exception:  # pragma: no cover
    torch = None


class MdecEncoder:
    """One encoder object = one ``mdec_encoder_t`` (psxavenc/mdec.h:57-63)."""

    def __init__(self, video_codec, video_width, video_height, max_frame_size=65536, device=0):
        self._h = C.c_void_p()
        self.video_codec, self.video_width, self.video_height = video_codec, video_width, video_height
        self.device = device
        self.max_frame_size = max_frame_size
        _lib.check(_lib.lib().psxhip_mdec_create(C.byref(self._h), device, video_codec, video_width, video_height,
                                                 max_frame_size))
        # mdec_encoder_state_t fields the reference's callers read or poke (filefmt.c:428-440,512,637-640,655)
        self.frame_index = 0
        self.frame_max_size = 0
        self.quant_scale = 0
        self.quant_scale_sum = 0
        self.bytes_used = 0
        self.blocks_used = 0
        self.uncomp_hwords_used = 0
        self.frame_output = None

    @property
    def frame_bytes(self):
        return self.video_width * self.video_height * 3 // 2

    def close(self):
        if self._h:
            _lib.lib().psxhip_mdec_destroy(self._h)
            self._h = C.c_void_p()

    destroy = close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference call pattern: one frame per call, host buffers -------------------------------
    def encode_frame_bs(self, video_frame):
        """encode_frame_bs (mdec.c:580): host NV21 frame in, ``frame_output`` (frame_max_size bytes) out."""
        assert self.frame_max_size >= 8, "caller sets frame_max_size first (filefmt.c:637-640)"
        out, res = self.encode_frames_host(np.asarray(video_frame, dtype=np.uint8).reshape(1, -1), self.frame_max_size)
        self.quant_scale, self.bytes_used, self.blocks_used, self.uncomp_hwords_used = (int(v) for v in res[0])
        self.quant_scale_sum += self.quant_scale
        self.frame_output = out[0]
        return self.frame_output

    # ---- batched host path ------------------------------------------------------------------------
    def encode_frames_host(self, frames, frame_max_sizes, out=None, res=None):
        """frames: (n, w*h*3/2) uint8 host array.  `out` / `res`: optional preallocated (n, stride) uint8 / (n, 4) int32
        arrays (e.g. page-locked ones, see register_host()) -- the library then writes into them instead of fresh memory."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n = frames.shape[0]
        assert frames.shape[1] == self.frame_bytes
        if np.isscalar(frame_max_sizes):
            sizes_p, uniform, stride = None, int(frame_max_sizes), int(frame_max_sizes)
        else:
            sizes = np.ascontiguousarray(frame_max_sizes, dtype=np.int32)
            sizes_p, uniform, stride = sizes.ctypes.data, 0, int(sizes.max())
        if out is None:
            out = np.zeros((n, stride), dtype=np.uint8)
        if res is None:
            res = np.zeros((n, 4), dtype=np.int32)
        assert out.dtype == np.uint8 and out.flags.c_contiguous and out.shape == (n, stride)
        assert res.dtype == np.int32 and res.flags.c_contiguous and res.shape == (n, 4)
        rc = _lib.lib().psxhip_mdec_encode_frames_host(self._h, frames.ctypes.data, n, sizes_p, uniform,
                                                       out.ctypes.data, stride, res.ctypes.data)
        _lib.check(rc)
        return out, res

    # ---- batched device path (torch tensors on the encoder's device) -----------------------------
    def encode_frames_device(self, d_frames, frame_max_sizes, d_out=None, d_results=None, stream=None):
        """d_frames: uint8 CUDA tensor (n, frame_stride).  frame_max_sizes: int or int32 CUDA tensor (n,).
        Returns (d_out (n, out_stride) uint8, d_results (n, 4) int32); asynchronous on the stream."""
        assert torch is not None and d_frames.is_cuda and d_frames.dtype == torch.uint8 and d_frames.dim() == 2
        n, fstride = d_frames.shape
        if isinstance(frame_max_sizes, int):
            sizes_p, uniform, mx = None, frame_max_sizes, frame_max_sizes
        else:
            assert frame_max_sizes.is_cuda and frame_max_sizes.dtype == torch.int32
            sizes_p, uniform, mx = frame_max_sizes.data_ptr(), 0, self.max_frame_size
        ostride = (mx + 3) & ~3
        if d_out is None:
            d_out = torch.zeros((n, ostride), dtype=torch.uint8, device=d_frames.device)
        if d_results is None:
            d_results = torch.zeros((n, 4), dtype=torch.int32, device=d_frames.device)
        st = stream if stream is not None else torch.cuda.current_stream(d_frames.device)
        rc = _lib.lib().psxhip_mdec_encode_frames_device(self._h, d_frames.data_ptr(), d_frames.stride(0), n, sizes_p,
                                                         uniform, d_out.data_ptr(), d_out.stride(0),
                                                         d_results.data_ptr(), st.cuda_stream)
        _lib.check(rc)
        return d_out, d_results


    def encode_batches_device(self, batches, frame_max_size, stream=None):
        """psxhip_mdec_encode_batches_device: several batches, ONE launch.  batches: sequence of (d_frames, d_out, d_results) or
        (d_frames, d_out, d_results, d_sizes) CUDA tensors with common row strides; frame_max_size: the uniform budget."""
        assert torch is not None and len(batches) > 0
        arr = (MdecBatch * len(batches))()
        fstride = ostride = None
        for i, b in enumerate(batches):
            d_frames, d_out, d_res = b[0], b[1], b[2]
            d_sizes = b[3] if len(b) > 3 else None
            assert d_frames.is_cuda and d_frames.dtype == torch.uint8 and d_out.dtype == torch.uint8 and d_res.dtype == torch.int32
            n = d_frames.shape[0]
            assert d_out.shape[0] == n and d_res.shape[0] == n
            if n:
                fstride = d_frames.stride(0) if fstride is None else fstride
                ostride = d_out.stride(0) if ostride is None else ostride
                assert d_frames.stride(0) == fstride and d_out.stride(0) == ostride, "batches share their row strides"
            arr[i] = MdecBatch(d_frames.data_ptr(), n, 0, d_sizes.data_ptr() if d_sizes is not None else None, d_out.data_ptr(), d_res.data_ptr())
        st = stream if stream is not None else torch.cuda.current_stream(batches[0][0].device)
        L = _lib.lib()
        L.psxhip_mdec_encode_batches_device.argtypes = [C.c_void_p, C.POINTER(MdecBatch), C.c_int, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p]
        _lib.check(L.psxhip_mdec_encode_batches_device(self._h, arr, len(batches), fstride or self.frame_bytes, int(frame_max_size),
                                                       ostride or ((int(frame_max_size) + 3) & ~3), st.cuda_stream))

    def set_lanes(self, lanes):
        """psxhip_mdec_set_lanes: 2 = consecutive launches of this context on one caller stream may overlap; the results of the
        LAST launch are ordered into the stream by the next encode call or by fence()"""
        L = _lib.lib()
        L.psxhip_mdec_set_lanes.argtypes = [C.c_void_p, C.c_int]
        _lib.check(L.psxhip_mdec_set_lanes(self._h, int(lanes)))

    def fence(self, stream=None):
        """psxhip_mdec_fence: order `stream` behind every launch of this context issued so far"""
        L = _lib.lib()
        L.psxhip_mdec_fence.argtypes = [C.c_void_p, C.c_void_p]
        st = stream if stream is not None else torch.cuda.current_stream(torch.device("cuda", self.device))
        _lib.check(L.psxhip_mdec_fence(self._h, st.cuda_stream))

    def watchdog(self):
        """psxhip_mdec_watchdog: frames the retry queue's watchdog gave up (0 on a healthy device); synchronises"""
        L = _lib.lib()
        L.psxhip_mdec_watchdog.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
        lost = C.c_uint(0)
        _lib.check(L.psxhip_mdec_watchdog(self._h, C.byref(lost)))
        return int(lost.value)


class MdecBatch(C.Structure):
    """psxhip_mdec_batch_t"""
    _fields_ = [("d_frames", C.c_void_p), ("n_frames", C.c_int32), ("reserved", C.c_int32), ("d_frame_max_sizes", C.c_void_p),
                ("d_out", C.c_void_p), ("d_results", C.c_void_p)]


def register_host(a):
    """Page-lock a numpy array's memory (psxhip_host_register): the host entry points then move it by DMA, no staging copy."""
    L = _lib.lib()
    L.psxhip_host_register.argtypes = [C.c_void_p, C.c_size_t]
    _lib.check(L.psxhip_host_register(a.ctypes.data, a.nbytes))


def unregister_host(a):
    L = _lib.lib()
    L.psxhip_host_unregister.argtypes = [C.c_void_p]
    _lib.check(L.psxhip_host_unregister(a.ctypes.data))


class MdecGeometry(C.Structure):
    """psxhip_mdec_geometry_t"""
    _fields_ = [("fits", C.c_int32), ("groups_per_cu", C.c_int32), ("wavefronts_per_group", C.c_int32),
                ("frames_in_flight", C.c_int32), ("max_frame_size_limit", C.c_int32), ("image_tile_bytes", C.c_int32),
                ("lds_bytes_per_group", C.c_int64), ("lds_bytes_per_cu", C.c_int64)]


def query_geometry(video_codec, video_width, video_height, max_frame_size, device=0):
    """psxhip_mdec_query_geometry: fit / shape / image tile of a geometry before creating a context for it"""
    L = _lib.lib()
    L.psxhip_mdec_query_geometry.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(MdecGeometry)]
    g = MdecGeometry()
    _lib.check(L.psxhip_mdec_query_geometry(device, video_codec, video_width, video_height, max_frame_size, C.byref(g)))
    return g
