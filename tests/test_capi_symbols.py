"""The C-ABI product library loads (no GPU needed) and exports every function include/*.h declares;
struct layouts match the reference's."""
import ctypes as C
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DECL = re.compile(r"^\s*(?:const\s+)?(?:unsigned\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**\s*\b([a-z_][a-z0-9_]*)\s*\(", re.M)


def declared_functions(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set()
    for m in DECL.finditer(text):
        name = m.group(1)
        if name not in ("defined", "sizeof", "_Static_assert"):
            names.add(name)
    return names


def test_library_exports_every_declared_function():
    from psxavenc_amd import _lib
    L = _lib.lib()
    total = 0
    for hdr in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        names = declared_functions(hdr)
        assert names, hdr
        for n in sorted(names):
            assert hasattr(L, n), "%s declares %s but libpsxav_hip.so does not export it" % (os.path.basename(hdr), n)
            total += 1
    assert total >= 30


def test_reference_surface_names_present():
    """exactly the functions the reference's FFI for this path would bind (mdec.h:65-74, libpsxav.h:73-101,174-176)"""
    from psxavenc_amd import _lib
    L = _lib.lib()
    for n in ("init_mdec_encoder", "destroy_mdec_encoder", "encode_frame_bs", "encode_sector_str",
              "psx_audio_xa_get_buffer_size", "psx_audio_spu_get_buffer_size", "psx_audio_xa_get_buffer_size_per_sector",
              "psx_audio_xa_get_samples_per_sector", "psx_audio_xa_get_sector_interleave", "psx_audio_xa_encode",
              "psx_audio_xa_encode_simple", "psx_audio_spu_encode", "psx_audio_spu_encode_simple",
              "psx_audio_xa_encode_finalize", "psx_cdrom_init_xa_subheader", "psx_cdrom_init_sector",
              "psx_cdrom_calculate_checksums"):
        assert hasattr(L, n), n


def test_struct_layouts_compile_to_reference_sizes(tmp_path):
    """sizeof(mdec_encoder_t) = 168, state = 152, channel state 24, stereo state 48, xa settings 24"""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "psxav_mdec.h"\n#include "psxav_audio.h"\n#include "psxav_hip.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mdec_encoder_t), sizeof(mdec_encoder_state_t),'
                   'sizeof(psx_audio_encoder_channel_state_t), sizeof(psx_audio_encoder_state_t), sizeof(psx_audio_xa_settings_t),'
                   'offsetof(mdec_encoder_state_t, frame_output), offsetof(mdec_encoder_t, state), sizeof(psxhip_adpcm_chain_t));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == [168, 152, 24, 48, 24, 40, 16, 24]


def test_size_helpers_and_cdrom_helpers_run_without_gpu(oracle):
    """pure host functions of the drop-in layer against the reference-pinned oracle"""
    import numpy as np
    from psxavenc_amd import _lib
    L = _lib.lib()

    class Settings(C.Structure):
        _fields_ = [("format", C.c_int), ("stereo", C.c_bool), ("frequency", C.c_int), ("bits_per_sample", C.c_int),
                    ("file_number", C.c_int), ("channel_number", C.c_int)]
    for f in ("psx_audio_xa_get_buffer_size_per_sector", "psx_audio_xa_get_samples_per_sector", "psx_audio_xa_get_sector_interleave"):
        getattr(L, f).argtypes = [Settings]
        getattr(L, f).restype = C.c_uint32
    L.psx_audio_xa_get_buffer_size.argtypes = [Settings, C.c_int]
    L.psx_audio_xa_get_buffer_size.restype = C.c_uint32
    L.psx_audio_spu_get_buffer_size.restype = C.c_uint32
    O = oracle
    for t in range(16):
        s = Settings(t & 1, bool(t & 2), 37800 if t & 4 else 18900, 8 if t & 8 else 4, 0, 0)
        os_ = O.XaSettings(t & 1, (t >> 1) & 1, s.frequency, s.bits_per_sample, 0, 0)
        assert L.psx_audio_xa_get_samples_per_sector(s) == O.lib().orc_xa_samples_per_sector(os_)
        assert L.psx_audio_xa_get_buffer_size_per_sector(s) == O.lib().orc_xa_sector_size(os_)
        assert L.psx_audio_xa_get_sector_interleave(s) == O.lib().orc_xa_sector_interleave(os_)
        sps = O.lib().orc_xa_samples_per_sector(os_)
        assert L.psx_audio_xa_get_buffer_size(s, 3 * sps + 1) == 4 * O.lib().orc_xa_sector_size(os_)
    assert L.psx_audio_spu_get_buffer_size(29) == 32
    rng = np.random.default_rng(3)
    for typ in (1, 2):
        a = rng.integers(0, 256, 2352).astype(np.uint8)
        b = a.copy()
        L.psx_cdrom_init_sector(a.ctypes.data_as(C.c_void_p), 12345, typ)
        O.lib().orc_cdrom_init_sector(O.ptr(b, O.u8p), 12345, typ)
        L.psx_cdrom_calculate_checksums(a.ctypes.data_as(C.c_void_p), typ)
        O.lib().orc_cdrom_calculate_checksums(O.ptr(b, O.u8p), typ)
        assert np.array_equal(a, b), typ


def test_no_gpu_means_loud_failure():
    """without a device the product path must fail (PSXHIP_EDEVICE), never fall back to the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from psxavenc_amd import _lib
    from psxavenc_amd.mdec import MdecEncoder
    with pytest.raises(_lib.PsxHipError) as e:
        MdecEncoder(0, 320, 240)
    assert e.value.code == _lib.PSXHIP_EDEVICE
