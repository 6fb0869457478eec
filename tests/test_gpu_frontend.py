"""GPU tests of the colour-conversion / scaling front-end (psxhip_scaler_*) through the C ABI: bit-exact
against oracle/frontend_oracle.c (the CPU statement of this library's own arithmetic -- parity with the reference's
libswscale is UNPINNED, FFmpeg is absent), filter banks tap by tap, edge cases, and the whole chain pictures -> NV21 in
HBM -> BS frames against oracle scaler + oracle encoder."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _pictures(fmt, w, h, n, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for k in range(n):
        base = 110 + 70 * np.sin((xx + 7 * k) / 19.0) * np.cos(yy / 27.0) + 35 * (((xx // 24) + (yy // 24) + k) % 2)
        if fmt == O.PIX_RGB24:
            pic = np.stack([np.clip(base + rng.integers(-5, 6, base.shape), 0, 255), np.clip(base * 0.7 + 40 + rng.integers(-5, 6, base.shape), 0, 255),
                            np.clip(250 - base, 0, 255)], axis=-1).astype(np.uint8).reshape(-1)
        else:
            y = np.clip(base + rng.integers(-5, 6, base.shape), 0, 255).astype(np.uint8)
            u = np.clip(128 + 45 * np.sin(xx[::2, ::2] / 31.0 + k), 0, 255).astype(np.uint8)
            v = np.clip(128 - 55 * np.cos(yy[::2, ::2] / 23.0), 0, 255).astype(np.uint8)
            pic = np.concatenate([y.ravel(), u.ravel(), v.ravel()])
        out.append(pic)
    return np.stack(out)


@pytest.mark.parametrize("fmt,sw,sh,dw,dh,full", [
    (0, 640, 480, 320, 240, True),        # the common case: VGA RGB -> the encoder's 320x240
    (1, 640, 480, 320, 240, True),
    (1, 640, 480, 320, 240, False),       # MPEG-range video, expanded
    (1, 352, 288, 320, 240, True),        # CIF, non-integer ratio
    (0, 200, 150, 320, 240, True),        # enlarging
    (1, 1280, 720, 320, 176, False),      # 4x down, wide taps
    (0, 320, 240, 320, 240, True),        # same size: colour conversion only
    (1, 720, 576, 640, 480, False),       # PAL -> the sbs v3 size
    (0, 97, 61, 48, 32, True),            # odd source, tiny target, partial tiles
    (1, 1920, 1080, 336, 192, True),      # 5.7x down: the smaller tile shape
])
def test_scaler_bit_exact_against_the_cpu_statement(fmt, sw, sh, dw, dh, full, monkeypatch):
    """every geometry three ways: the launch's own choice of vertical segments (a handful of pictures: nearly one segment per
    tile, every tile stages its whole reach), ONE segment (a band walks the whole picture: every tile but the first takes most of
    its rows from the ring the tiles before it filled), and two."""
    from psxavenc_amd.frontend import Scaler
    pics = _pictures(fmt, sw, sh, 3, seed=sw + dh)
    sc = Scaler(fmt, sw, sh, dw, dh, src_full_range=full)
    want = O.scaler_convert(fmt, sw, sh, full, dw, dh, pics)
    for segs in ("1", "2"):
        monkeypatch.setenv("PSXHIP_SCALER_VSEGS", segs)
        got = sc.convert_host(pics)
        assert np.array_equal(got, want), "vertical segments = %s: %d bytes differ" % (segs, int((got != want).sum()))
    monkeypatch.delenv("PSXHIP_SCALER_VSEGS")
    for which, (s_, d_) in enumerate([(sw, dw), (sh, dh), ((sw // 2) if fmt else sw, dw // 2), ((sh // 2) if fmt else sh, dh // 2)]):
        taps, left, coef = sc.filter(which)
        ot, ol, oc = O.scaler_filter(s_, d_)
        assert taps == ot and np.array_equal(left, ol) and np.array_equal(coef, oc), which
    got = sc.convert_host(pics)
    if not np.array_equal(got, want):
        bad = np.nonzero(got != want)
        raise AssertionError("%d bytes differ; first at frame %d byte %d (%d vs %d)" % (bad[0].size, bad[0][0], bad[1][0], got[bad[0][0], bad[1][0]], want[bad[0][0], bad[1][0]]))
    sc.close()


def test_scaler_rejects_what_it_cannot_do():
    from psxavenc_amd import _lib
    from psxavenc_amd.frontend import Scaler
    for args in ((0, 640, 480, 321, 240), (1, 641, 480, 320, 240), (0, 640, 480, 0, 0), (0, 16384 * 2, 480, 320, 240), (0, 8000, 480, 320, 240), (7, 64, 64, 64, 64)):
        with pytest.raises(_lib.PsxHipError):
            Scaler(*args)


def test_pictures_to_bs_frames_without_leaving_hbm():
    """the chain this row exists for: RGB pictures resident in HBM -> psxhip_scaler_convert_device -> the encoder's d_frames ->
    psxhip_mdec_encode_frames_device, all on one stream; against oracle scaler + oracle encoder"""
    import torch
    from psxavenc_amd.frontend import Scaler
    from psxavenc_amd.mdec import MdecEncoder
    sw, sh, w, h, n, budget = 640, 480, 320, 240, 300, 8192
    pics = _pictures(O.PIX_RGB24, sw, sh, 12, seed=5)
    pics = np.concatenate([pics] * (n // 12))
    sc = Scaler(O.PIX_RGB24, sw, sh, w, h)
    enc = MdecEncoder(0, w, h, max_frame_size=budget)
    d_pics = torch.from_numpy(pics).to("cuda:0")
    d_frames = sc.convert_device(d_pics)
    d_out, d_res = enc.encode_frames_device(d_frames, budget)
    torch.cuda.synchronize()
    frames = O.scaler_convert(O.PIX_RGB24, sw, sh, True, w, h, pics[:12])
    assert np.array_equal(d_frames.cpu().numpy()[:12], frames) and np.array_equal(d_frames.cpu().numpy()[-12:], frames)
    want, want_res, rc = O.mdec_encode(0, w, h, frames, budget)
    assert rc == 0
    got, res = d_out.cpu().numpy()[:, :budget], d_res.cpu().numpy()
    for k in range(0, n, 12):
        assert np.array_equal(got[k:k + 12], want) and np.array_equal(res[k:k + 12], want_res), k
    sc.close()
    enc.close()
