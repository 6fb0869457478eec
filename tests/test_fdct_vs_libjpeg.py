"""The oracle's 8x8 forward DCT against an implementation nobody here wrote.

FFmpeg's FDCT (the one arithmetic of the MDEC path that is absent from /root/reference and from this image) is the IJG "jfdctint"
integer DCT with ONE constant changed: libavcodec/jfdctint_template.c keeps 4 extra bits after the row pass for 8-bit samples
(PASS1_BITS 4, OUT_SHIFT = PASS1_BITS) where the IJG original keeps 2.  The IJG original's compiled form is in this image:
libjpeg-turbo exports jpeg_fdct_islow (jfdctint.c, DCTELEM = short).  oracle/mdec_oracle.c states the butterfly once, with the
number of extra bits as a parameter; this test holds its 2-bit instance to libjpeg-turbo bit for bit -- constants, butterfly
order, rounding, int16 stores between the passes -- so that what the MDEC oracle leaves unchecked is the value of that one
parameter in libavcodec 8.0.1's build, not the algorithm.  (Not a pin to FFmpeg: DESIGN.md section 2 still says "unpinned".)"""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

import oracle_lib as O

i16p = C.POINTER(C.c_int16)


def _libjpeg():
    for name in ("libjpeg.so.8", "libjpeg.so.62", ctypes.util.find_library("jpeg")):
        if not name:
            continue
        try:
            lib = C.CDLL(name)
            lib.jpeg_fdct_islow.argtypes = [C.c_void_p]
            lib.jpeg_fdct_islow.restype = None
            return lib
        except (OSError, AttributeError):
            continue
    return None


def _blocks():
    rng = np.random.default_rng(20260929)
    out = [np.full(64, v, np.int16) for v in (-128, 127, 0, 1, -1)]
    yy, xx = np.mgrid[0:8, 0:8]
    out += [(((xx + yy) % 2) * 255 - 128).astype(np.int16).ravel(), ((xx % 2) * 255 - 128).astype(np.int16).ravel(),
            ((yy % 2) * 255 - 128).astype(np.int16).ravel(), (xx * 36 - 128).astype(np.int16).ravel(), (yy * 36 - 128).astype(np.int16).ravel()]
    for k in range(64):                                        # single samples at both extremes
        for v in (-128, 127):
            b = np.zeros(64, np.int16); b[k] = v; out.append(b)
    out += list(rng.integers(-128, 128, (60000, 64)).astype(np.int16))            # noise
    smooth = np.clip(rng.integers(-128, 128, (20000, 1)) + rng.integers(-6, 7, (20000, 64)), -128, 127).astype(np.int16)
    out += list(smooth)                                                             # flat blocks: rounding near zero
    return np.stack(out)


def test_two_bit_instance_of_the_oracle_dct_equals_libjpeg_turbo():
    J = _libjpeg()
    if J is None:
        pytest.skip("no libjpeg with jpeg_fdct_islow in this image")
    L = O.lib()
    L.orc_fdct_islow8_pass1.argtypes = [i16p, C.c_int]
    blocks = _blocks()
    # DCTELEM is short in SIMD-enabled builds of libjpeg-turbo and int otherwise: find out on a block whose answer is known
    probe = np.zeros(128, np.int16); probe[:64] = 100
    J.jpeg_fdct_islow(probe.ctypes.data)
    if probe[0] != 100 * 64 or probe[1:64].any():
        pytest.skip("this libjpeg's DCTELEM is not 16 bits wide")
    diff_to_ffmpeg_form = 0
    for blk in blocks:
        a = blk.copy(); L.orc_fdct_islow8_pass1(a.ctypes.data_as(i16p), 2)
        b = blk.copy(); J.jpeg_fdct_islow(b.ctypes.data)
        assert np.array_equal(a, b), (blk.reshape(8, 8), a.reshape(8, 8), b.reshape(8, 8))
        c = blk.copy(); L.orc_fdct_islow8(c.ctypes.data_as(i16p))
        d = blk.copy(); L.orc_fdct_islow8_pass1(d.ctypes.data_as(i16p), 4)
        assert np.array_equal(c, d)                           # the encoder's DCT is the 4-bit instance of the same text
        diff_to_ffmpeg_form = max(diff_to_ffmpeg_form, int(np.abs(c.astype(np.int32) - a.astype(np.int32)).max()))
    # the two instances are the same transform up to the rounding of the intermediate: never further apart than 2 units of an
    # output that carries a factor of 8 (and they DO differ: the parameter matters to the bytes)
    assert 1 <= diff_to_ffmpeg_form <= 2, diff_to_ffmpeg_form
