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
    16128 / 18144 x3 (mdec.c:768-775).  COMPLETE tail: the stream ends with the last frame's last sector; REFERENCE tail
    (default): the CLI's loop stops once <= frames_needed = 2 frames are left (filefmt.c:443-450)"""
    from psxavenc_amd import strmux
    from psxavenc_amd.parallel import str_frame_budgets
    s = strmux.settings(tail=strmux.TAIL_COMPLETE)
    p = strmux.plan(s, 160)
    assert (p.interleave, p.sector_size, p.audio_samples_per_sector, p.max_frame_size) == (8, 2352, 2016, 18144)
    b = strmux.frame_budgets(s, 0, 160)
    assert b[:5].tolist() == [16128, 18144, 18144, 18144, 16128]
    assert b.tolist() == str_frame_budgets(160, 75 * 2 * 7 * 1, 8 * 15)
    assert strmux.frame_budgets(s, 37, 9).tolist() == b[37:46].tolist()          # any rank can budget its own range
    assert p.n_video_sectors == int(b.sum()) // 2016 == 1400
    assert p.n_sectors == 1600 and p.n_audio_sectors == 200 and p.n_frames_encoded == 160
    # the reference's tail: 158 of the 160 frames, and the budgets do not depend on the tail
    r = strmux.plan(strmux.settings(), 160)
    assert r.n_frames_encoded == 158 and r.n_video_sectors == int(b[:158].sum()) // 2016
    assert strmux.frame_budgets(strmux.settings(), 0, 160).tolist() == b.tolist()
    # video only (strv): every sector is video, 2336-byte sectors
    v = strmux.plan(strmux.settings(fmt=strmux.FORMAT_STRV, channels=0, tail=strmux.TAIL_COMPLETE), 30)
    assert v.n_audio_sectors == 0 and v.n_sectors == v.n_video_sectors and v.sector_size == 2336 and v.interleave == 1
    # trailing audio: the audio sector closes each block
    t = strmux.plan(strmux.settings(trailing_audio=True, tail=strmux.TAIL_COMPLETE), 160)
    assert t.n_video_sectors == 1400 and t.n_sectors == 1599
    with pytest.raises(Exception):
        strmux.plan(strmux.settings(width=321), 4)
    with pytest.raises(Exception):
        strmux.plan(strmux.settings(tail=7), 4)


@pytest.mark.parametrize("tail", [0, 1])
@pytest.mark.parametrize("channels,trailing", [(0, False), (2, False), (2, True), (1, True)])
def test_str_plan_never_takes_a_frame_it_was_not_given(channels, trailing, tail):
    """n_frames = 0 (the reference asserts in its decoder there -- nothing to mirror) must not plan frame 0; n_frames = 1 plans it"""
    from psxavenc_amd import strmux
    for pcm in (0, 1000, 5000, strmux.PLENTY_OF_AUDIO):
        s = strmux.settings(channels=channels, trailing_audio=trailing, tail=tail, fmt=strmux.FORMAT_STRCD if channels else strmux.FORMAT_STRV)
        p0 = strmux.plan(s, 0, pcm)
        assert p0.n_frames_encoded == 0 and p0.n_video_sectors == 0
        rows = strmux.plan_sectors(s, 0, pcm)
        assert rows.shape[0] == p0.n_sectors and not (rows[:, 0] == strmux.SECTOR_VIDEO).any()
        p1 = strmux.plan(s, 1, pcm)
        assert p1.n_frames_encoded <= 1
        rows = strmux.plan_sectors(s, 1, pcm)
        assert (rows[rows[:, 0] == strmux.SECTOR_VIDEO][:, 1] == 0).all()
        for n in (2, 3, 7):
            assert strmux.plan(s, n, pcm).n_frames_encoded <= n


def _stream_structure(stream, fmt):
    """(kind, frame, index, eof) per sector, read back from the bytes the reference's loop produced"""
    at = {6: 0x08, 7: 0x18, 9: 0x00}[fmt]
    sub = 0x12 if fmt == 7 else 0x02
    rows = []
    audio = 0
    for sec in stream:
        if sec[at] == 0x60 and sec[at + 1] == 0x01:                      # STR chunk header, mdec.c:786-787
            frame = int(sec[at + 8]) | int(sec[at + 9]) << 8 | int(sec[at + 10]) << 16
            rows.append((0, frame - 1, int(sec[at + 4]) | int(sec[at + 5]) << 8, 0))
        elif not sec.any():
            rows.append((2, -1, -1, 0))
        else:
            rows.append((1, -1, audio, int(sec[sub] >> 7)))
            audio += 1
    return np.array(rows, np.int32).reshape(-1, 4)


@pytest.mark.parametrize("fmt,channels,bits,freq,speed,fps,trailing", [
    (7, 2, 4, 37800, 2, (15, 1), False),      # config 'strcd v2'
    (7, 2, 4, 37800, 2, (15, 1), True),
    (6, 1, 4, 37800, 2, (15, 1), False),
    (6, 2, 8, 18900, 1, (10, 1), False),
    (9, 0, 4, 37800, 2, (15, 1), False),
    (7, 2, 4, 37800, 2, (30000, 1001), False),
    (6, 1, 8, 37800, 2, (25, 1), True),
])
def test_str_plan_follows_the_reference_sector_loop(oracle, fmt, channels, bits, freq, speed, fps, trailing):
    """psxhip_str_plan_sectors (the product's dry run of the sector loop, REFERENCE tail) against the structure of the
    stream that encode_file_str restated over the oracle (tests/str_reference_loop.py, filefmt.c:391-520 + decoding.c:510-586)
    produces: which sector is video / audio / an empty audio slot, frame and chunk numbers, EOF flags, where the stream
    ends -- for plenty of audio, audio that ends first (incl. mid-sector and exactly on a sector), and tiny inputs."""
    import str_reference_loop as R
    from psxavenc_amd import strmux
    w, h = 16, 16                              # the structure does not depend on the picture
    s = strmux.settings(fmt=fmt, codec=0, width=w, height=h, fps_num=fps[0], fps_den=fps[1], cd_speed=speed,
                        trailing_audio=trailing, channels=channels, frequency=freq, bits=bits)
    sps = strmux.plan(s, 4).audio_samples_per_sector
    for n_frames in (1, 2, 3, 4, 7, 24):
        frames = oracle.synth_frames(w, h, n_frames, seed=3, amp=2)
        lengths = [0] if not channels else [10 ** 6, 0, 1, sps - 1, sps, sps + 1, 2 * sps, 3 * sps + 77, 7 * sps]
        for n_audio in lengths:
            pcm = np.zeros(n_audio * max(1, channels), np.int16)
            pcm[:] = (np.arange(pcm.size) * 37 % 2001 - 1000)
            want, _, frames_encoded = R.encode_file_str(fmt, 0, w, h, fps[0], fps[1], speed, frames, pcm, channels=channels, freq=freq,
                                                        bits=bits, trailing_audio=trailing)
            got = strmux.plan_sectors(s, n_frames, n_audio)
            p = strmux.plan(s, n_frames, n_audio)
            ctx = (n_frames, n_audio)
            assert got.shape[0] == want.shape[0] == p.n_sectors, ctx
            assert p.n_frames_encoded == frames_encoded, ctx
            ws = _stream_structure(want, fmt)
            assert np.array_equal(got, ws), (ctx, np.nonzero((got != ws).any(axis=1))[0][:5])


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


def test_fdct_pin_kit_still_compiles_against_its_declared_ffmpeg_surface(tmp_path):
    """tools/check_fdct_vs_ffmpeg.c is the one way to close the MDEC parity gap (the FDCT is FFmpeg's, mdec.c:640) and it
    cannot be built here (no FFmpeg).  Keep it from rotting: -fsyntax-only against a prototype list of exactly the FFmpeg
    surface it uses -- written down HERE as part of the test, not shipped as headers -- plus the repository's real headers
    for everything else (oracle/mdec_oracle.h, include/psxav_hip.h)."""
    import subprocess
    inc = tmp_path / "inc"
    (inc / "libavcodec").mkdir(parents=True)
    (inc / "libavutil").mkdir()
    (inc / "libavcodec" / "avcodec.h").write_text("unsigned avcodec_version(void);\nconst char *avcodec_configuration(void);\n#define FF_DCT_INT 2\n")
    (inc / "libavcodec" / "avdct.h").write_text(
        "#include <stdint.h>\n#include <stddef.h>\n"
        "typedef struct AVDCT { const void *av_class; void (*idct)(int16_t *block); void (*fdct)(int16_t *block);\n"
        "  int dct_algo; int idct_algo; void (*get_pixels)(int16_t *block, const uint8_t *pixels, ptrdiff_t line_size);\n"
        "  int bits_per_sample; } AVDCT;\n"
        "AVDCT *avcodec_dct_alloc(void);\nint avcodec_dct_init(AVDCT *);\n")
    (inc / "libavcodec" / "version.h").write_text("#define LIBAVCODEC_IDENT \"Lavc (prototype list of the test)\"\n#define LIBAVCODEC_VERSION_MAJOR 0\n"
                                                  "#define LIBAVCODEC_VERSION_MINOR 0\n#define LIBAVCODEC_VERSION_MICRO 0\n")
    (inc / "libavutil" / "mem.h").write_text("void av_free(void *ptr);\n")
    (inc / "libavutil" / "opt.h").write_text("#include <stdint.h>\nint av_opt_set_int(void *obj, const char *name, int64_t val, int search_flags);\n")
    src = os.path.join(ROOT, "tools", "check_fdct_vs_ffmpeg.c")
    for extra in ([], ["-DWITH_DEVICE"]):
        r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-I", str(inc),
                            "-I", os.path.join(ROOT, "oracle"), "-I", os.path.join(ROOT, "include"), src] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_pin_fdct_reports_unavailable_in_one_line_without_ffmpeg():
    """tools/pin_fdct.py is the one command that turns MDEC parity from "unpinned at the FDCT" into "pinned" (README): with an FFmpeg it
    prints one PASS / FAIL line with the SHA-256 of AVDCT's output vector; in this image (no FFmpeg) it has to say UNAVAILABLE in one
    line and exit 2 -- not a traceback, and never PASS"""
    import shutil
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_fdct.py"), "--no-device", "--blocks", "1000"], capture_output=True, text=True)
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("PIN_FDCT "), (r.stdout, r.stderr[-500:])
    have_ffmpeg = any(os.path.exists(os.path.join(d, "libavcodec", "avdct.h")) for d in ("/usr/include", "/usr/local/include", "/usr/include/x86_64-linux-gnu"))
    if not have_ffmpeg:
        assert r.returncode == 2 and "UNAVAILABLE" in lines[0]
    else:
        assert r.returncode in (0, 1) and "sha256=" in lines[0]
