"""Host-side logic of the batched callers (no GPU needed): the STR sector plan / frame budgets and the SPU-file
layouts are pure functions; the encode entry points must fail loudly without a device (no CPU fallback)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_str_plan_and_budgets_config_strcd():
    """config 'strcd v2' (SURVEY 3.2): 320x240 @15 fps, 2x, 37800 Hz 4-bit stereo: interleave 8, budgets cycling
    16128 / 18144 x3 (mdec.c:768-775), stream ends with the last frame's last sector (filefmt.c:450)"""
    from psxavenc_amd import strmux
    from psxavenc_amd.parallel import str_frame_budgets
    s = strmux.settings()
    p = strmux.plan(s, 160)
    assert (p.interleave, p.sector_size, p.audio_samples_per_sector, p.max_frame_size) == (8, 2352, 2016, 18144)
    b = strmux.frame_budgets(s, 0, 160)
    assert b[:5].tolist() == [16128, 18144, 18144, 18144, 16128]
    assert b.tolist() == str_frame_budgets(160, 75 * 2 * 7 * 1, 8 * 15)
    assert strmux.frame_budgets(s, 37, 9).tolist() == b[37:46].tolist()          # any rank can budget its own range
    assert p.n_video_sectors == int(b.sum()) // 2016 == 1400
    assert p.n_sectors == 1600 and p.n_audio_sectors == 200
    # video only (strv): every sector is video, 2336-byte sectors
    v = strmux.plan(strmux.settings(fmt=strmux.FORMAT_STRV, channels=0), 30)
    assert v.n_audio_sectors == 0 and v.n_sectors == v.n_video_sectors and v.sector_size == 2336 and v.interleave == 1
    # trailing audio: the audio sector closes each block
    t = strmux.plan(strmux.settings(trailing_audio=True), 160)
    assert t.n_video_sectors == 1400 and t.n_sectors == 1599
    with pytest.raises(Exception):
        strmux.plan(strmux.settings(width=321), 4)


def test_spu_file_sizes_match_reference_golden():
    """psxhip_spu_file_size against the sizes of the files the reference's framing produces (tests/golden/spufile_ref.npz)"""
    from psxavenc_amd import _lib, spufile
    spec = importlib.util.spec_from_file_location("make_spufile_golden", os.path.join(ROOT, "tests", "golden", "make_spufile_golden.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "spufile_ref.npz"))
    L = spufile._bind()
    for key, fmt, opts, rec in G.CASES:
        s = spufile.settings(fmt, channels=rec["channels"], interleave=opts.get("interleave", 2048), alignment=opts.get("alignment"),
                             loop_point=opts.get("loop_point", -1), enable_loop=opts.get("enable_loop", False),
                             no_dummy=opts.get("no_dummy", False))
        assert L.psxhip_spu_file_size(C.byref(s), rec["n"]) == int(gold[key + "_size"][0]), key
    assert L.psxhip_spu_file_size(C.byref(spufile.settings(spufile.FORMAT_SPU)), 22050) == 12672     # config 'spu'
    bad = spufile.settings(spufile.FORMAT_SPU, channels=2)
    assert L.psxhip_spu_file_size(C.byref(bad), 100) == _lib.PSXHIP_EINVAL


def test_batched_callers_fail_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from psxavenc_amd import _lib, spufile, strmux
    with pytest.raises(_lib.PsxHipError) as e:
        strmux.encode(strmux.settings(width=32, height=32), np.zeros((2, 32 * 32 * 3 // 2), np.uint8), np.zeros(9000, np.int16))
    assert e.value.code == _lib.PSXHIP_EDEVICE
    with pytest.raises(_lib.PsxHipError) as e:
        spufile.encode(spufile.settings(spufile.FORMAT_SPU), np.zeros(280, np.int16))
    assert e.value.code == _lib.PSXHIP_EDEVICE
    L = _lib.lib()
    L.psxhip_mdec_fdct_host.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    b = np.zeros(64, np.int16)
    assert L.psxhip_mdec_fdct_host(0, b.ctypes.data, 1, b.ctypes.data) == _lib.PSXHIP_EDEVICE
