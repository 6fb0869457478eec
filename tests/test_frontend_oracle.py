"""The CPU checker of the colour-conversion / scaling front-end (oracle/frontend_oracle.c) against what can
be known without libswscale (parity with the reference's scaler is UNPINNED -- FFmpeg is absent): the filter bank's
invariants, exact identities, and closeness to a double-precision bicubic + BT.601 full-range model.  Tolerances are the
fixed-point format's: 14-bit coefficients, 15-bit intermediates."""
import numpy as np
import pytest


def _cubic(x, B=0.0, Cc=0.6):
    x = np.abs(x)
    w1 = ((12 - 9 * B - 6 * Cc) * x ** 3 + (-18 + 12 * B + 6 * Cc) * x ** 2 + (6 - 2 * B)) / 6
    w2 = ((-B - 6 * Cc) * x ** 3 + (6 * B + 30 * Cc) * x ** 2 + (-12 * B - 48 * Cc) * x + (8 * B + 24 * Cc)) / 6
    return np.where(x < 1, w1, np.where(x < 2, w2, 0.0))


def _float_scale(plane, dw, dh):
    """double-precision separable bicubic with the same geometry (centres, support widening, edge replication)"""
    def one(p, d):
        s = p.shape[1]
        ratio = s / d
        scale = max(1.0, ratio)
        out = np.zeros((p.shape[0], d))
        for i in range(d):
            c = (i + 0.5) * ratio - 0.5
            lo = int(np.floor(c - 2 * scale)) + 1
            ks = np.arange(lo, lo + int(np.ceil(4 * scale)) + 1)
            w = _cubic((ks - c) / scale)
            w /= w.sum()
            out[:, i] = (p[:, np.clip(ks, 0, s - 1)] * w).sum(axis=1)
        return out
    return one(one(plane.astype(np.float64), dw).T, dh).T


@pytest.mark.parametrize("src,dst", [(640, 320), (320, 320), (352, 320), (1280, 320), (160, 320), (720, 304), (853, 320), (17, 16)])
def test_filter_bank_invariants(oracle, src, dst):
    taps, left, coef = oracle.scaler_filter(src, dst)
    assert (coef.astype(np.int64).sum(axis=1) == 16384).all()
    ratio = max(1.0, src / dst)
    assert taps == int(np.ceil(4 * ((src * 65536 + dst // 2) // dst if src > dst else 65536) / 65536)) or taps >= 4
    centres = (np.arange(dst) + 0.5) * src / dst - 0.5
    assert (left <= np.floor(centres)).all() and (left + taps - 1 >= np.floor(centres)).all()
    # against the real-valued kernel: quantisation to 14 bits, and positions that advance by a 16.16 increment (as
    # libswscale's do): up to dst * 2^-17 source pixels off at the far end
    for i in (0, dst // 3, dst - 1):
        ks = left[i] + np.arange(taps)
        w = _cubic((ks - centres[i]) / ratio)
        w = w / w.sum() * 16384
        assert np.abs(coef[i] - w).max() < 12 + 16384 * 1.5 * dst / 131072, (i, coef[i], w)
    if src == dst:
        assert (coef.max(axis=1) == 16384).all()          # identity: one tap of weight 1


def test_identity_geometry_is_exact(oracle):
    """same size in and out: YUV420P full range comes back byte for byte (as NV21); RGB grey ramps give Y = the grey"""
    rng = np.random.default_rng(1)
    w, h = 64, 48
    pic = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
    out = oracle.scaler_convert(oracle.PIX_YUV420P, w, h, True, w, h, pic[None])[0]
    assert np.array_equal(out[:w * h], pic[:w * h])
    u, v = pic[w * h:w * h + w * h // 4], pic[w * h + w * h // 4:]
    assert np.array_equal(out[w * h::2], v) and np.array_equal(out[w * h + 1::2], u)        # NV21: Cr first
    grey = np.repeat(np.tile(np.arange(w, dtype=np.uint8) * 4, h)[:, None], 3, axis=1).reshape(-1)
    out = oracle.scaler_convert(oracle.PIX_RGB24, w, h, True, w, h, grey[None])[0]
    assert np.array_equal(out[:w * h], np.tile(np.arange(w, dtype=np.uint8) * 4, h))
    assert (out[w * h:] == 128).all()


@pytest.mark.parametrize("fmt,sw,sh,dw,dh", [(0, 640, 480, 320, 240), (1, 640, 480, 320, 240), (1, 352, 288, 320, 240), (0, 200, 150, 320, 240),
                                             (1, 1280, 720, 320, 176)])
def test_close_to_the_real_valued_model(oracle, fmt, sw, sh, dw, dh):
    """|fixed point - double| <= 1 everywhere and < 0.6 in the mean: what 14-bit coefficients and a 15-bit intermediate give"""
    rng = np.random.default_rng(sw + dw)
    yy, xx = np.mgrid[0:sh, 0:sw]
    base = (96 + 64 * np.sin(xx / 17.0) * np.cos(yy / 23.0) + 40 * ((xx // 32 + yy // 32) % 2)).astype(np.float64)
    if fmt == 0:
        rgb = np.stack([np.clip(base + rng.integers(-6, 7, base.shape), 0, 255), np.clip(base * 0.8 + 20, 0, 255),
                        np.clip(255 - base, 0, 255)], axis=-1).astype(np.uint8)
        out = oracle.scaler_convert(fmt, sw, sh, True, dw, dh, rgb.reshape(1, -1))[0]
        r, g, b = (rgb[..., k].astype(np.float64) for k in range(3))
        # the model scales the same 8-bit planes the integer path builds (the colour matrix's own rounding is checked apart)
        y8 = np.floor((19595 * r + 38470 * g + 7471 * b + 32768) / 65536)
        cb8 = np.clip(np.floor((-11059 * r - 21709 * g + 32768 * b + 32768) / 65536) + 128, 0, 255)
        cr8 = np.clip(np.floor((32768 * r - 27439 * g - 5329 * b + 32768) / 65536) + 128, 0, 255)
        assert np.abs(y8 - (0.299 * r + 0.587 * g + 0.114 * b)).max() <= 1.0
        assert np.abs(cb8 - (128 - 0.168736 * r - 0.331264 * g + 0.5 * b)).max() <= 1.0
        planes = [(y8, dw, dh), (cr8, dw // 2, dh // 2), (cb8, dw // 2, dh // 2)]
    else:
        y = np.clip(base + rng.integers(-6, 7, base.shape), 0, 255).astype(np.uint8)
        u = np.clip(128 + 40 * np.sin(xx[::2, ::2] / 29.0), 0, 255).astype(np.uint8)
        v = np.clip(128 - 50 * np.cos(yy[::2, ::2] / 31.0), 0, 255).astype(np.uint8)
        out = oracle.scaler_convert(fmt, sw, sh, True, dw, dh, np.concatenate([y.ravel(), u.ravel(), v.ravel()])[None])[0]
        planes = [(y.astype(np.float64), dw, dh), (v.astype(np.float64), dw // 2, dh // 2), (u.astype(np.float64), dw // 2, dh // 2)]
    got = [out[:dw * dh].reshape(dh, dw), out[dw * dh::2].reshape(dh // 2, dw // 2), out[dw * dh + 1::2].reshape(dh // 2, dw // 2)]
    for (p, w_, h_), g_ in zip(planes, got):
        want = np.clip(_float_scale(p, w_, h_), 0, 255)
        err = np.abs(g_.astype(np.float64) - want)
        assert err.max() <= 1.0 + 1e-9 and err.mean() < 0.6, (err.max(), err.mean())


def test_limited_range_input_expands_to_full_range(oracle):
    """MPEG-range YUV (16..235 / 16..240) -> the full range the encoder expects (decoding.c:301-311: dst range = JPEG)"""
    w, h = 64, 32
    for yv, want in ((16, 0), (235, 255), (126, 128)):
        pic = np.concatenate([np.full(w * h, yv, np.uint8), np.full(w * h // 2, 128, np.uint8)])
        out = oracle.scaler_convert(oracle.PIX_YUV420P, w, h, False, w, h, pic[None])[0]
        assert abs(int(out[0]) - want) <= 1 and (out[:w * h] == out[0]).all()
        assert (np.abs(out[w * h:].astype(int) - 128) <= 1).all()
    pic = np.concatenate([np.full(w * h, 128, np.uint8), np.full(w * h // 4, 240, np.uint8), np.full(w * h // 4, 16, np.uint8)])
    out = oracle.scaler_convert(oracle.PIX_YUV420P, w, h, False, w, h, pic[None])[0]
    assert (out[w * h::2] <= 1).all() and (out[w * h + 1::2] >= 254).all()       # Cr from V = 16, Cb from U = 240
