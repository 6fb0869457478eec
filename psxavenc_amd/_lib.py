"""ctypes loader for libpsxav_hip.so (the C-ABI product library).

There is deliberately no fallback: if the HIP extension has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C psxavenc_amd/csrc``) importing any operator raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PSXAV_HIP_LIB: another build of the same library, for A/B measurements of kernel revisions on one box)
LIB_PATH = os.environ.get("PSXAV_HIP_LIB") or os.path.join(_HERE, "libpsxav_hip.so")

PSXHIP_OK, PSXHIP_EINVAL, PSXHIP_EDEVICE, PSXHIP_ENOMEM, PSXHIP_ENOFIT = 0, -1, -2, -3, -4


class MdecResult(C.Structure):
    """psxhip_mdec_result_t (include/psxav_hip.h)"""
    _fields_ = [("quant_scale", C.c_int32), ("bytes_used", C.c_int32),
                ("blocks_used", C.c_int32), ("uncomp_hwords_used", C.c_int32)]


class AdpcmState(C.Structure):
    _fields_ = [("prev1", C.c_int32), ("prev2", C.c_int32)]


class AdpcmChain(C.Structure):
    _fields_ = [("sample_offset", C.c_int64), ("pitch", C.c_int32), ("sample_limit", C.c_int32),
                ("n_units", C.c_int32), ("reserved", C.c_int32)]


class PsxHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("psxhip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "psxavenc_amd: %s is missing -- build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % LIB_PATH)
    # PyTorch ships its own HIP runtime; whichever libamdhip64 is loaded first serves the whole process.  Import
    # torch first (when present) so that the library and the tensors it is handed share one runtime and one view
    # of the devices -- loading this library before torch leaves torch-allocated memory invisible to it.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, u8p = C.c_void_p, C.c_size_t, C.c_int, C.c_void_p
    L.psxhip_last_error.restype = C.c_char_p
    L.psxhip_version.restype = C.c_char_p
    L.psxhip_mdec_kernel_name.restype = C.c_char_p
    L.psxhip_mdec_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32]
    L.psxhip_mdec_destroy.argtypes = [vp]
    L.psxhip_mdec_destroy.restype = None
    L.psxhip_mdec_encode_frames_device.argtypes = [vp, u8p, sz, i32, vp, i32, u8p, sz, vp, vp]
    L.psxhip_mdec_encode_frames_host.argtypes = [vp, u8p, i32, vp, i32, u8p, sz, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise PsxHipError(rc, lib().psxhip_last_error().decode("utf-8", "replace"))
