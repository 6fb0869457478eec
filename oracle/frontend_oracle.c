/*
 * oracle/frontend_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU checker of the colour-conversion / scaling front-end
 * (SURVEY 8(f4)).
 *
 * PARITY UNPINNED, and it cannot be otherwise here: the reference hands this step to FFmpeg's libswscale
 * (psxavenc/decoding.c:287-311: sws_getContext(src fmt -> AV_PIX_FMT_NV21, SWS_BICUBIC), sws_setColorspaceDetails(dst =
 * ITU-R BT.601 coefficients, full range); :463-475: sws_scale into the frame buffer), and libswscale is neither under
 * /root/reference nor installed in this image.  There is no reference arithmetic to restate, no golden vector to check
 * against.  This file therefore states the arithmetic the PRODUCT defines for that step ("psxhip front-end v1":
 * psxavenc_amd/csrc/frontend_kernels.hip) in plain C, written independently of the kernel from the specification in
 * DESIGN.md section 9, and the tests hold the kernel to it bit for bit; what relates either of them to swscale is
 * only structure (same filter family, same separable two-pass fixed-point layout, same output format and colour matrix)
 * and the float sanity bounds of tests/test_frontend_oracle.py.
 *
 * Specification (all integer; ">>" on negative values is an arithmetic shift = floor):
 *   planes    RGB24  -> full-resolution Y, Cb, Cr, BT.601 full range (JPEG), 16-bit coefficients:
 *                       Y  = (19595 R + 38470 G +  7471 B + 32768) >> 16
 *                       Cb = ((-11059 R - 21709 G + 32768 B + 32768) >> 16) + 128      (clipped to 0..255)
 *                       Cr = (( 32768 R - 27439 G -  5329 B + 32768) >> 16) + 128
 *             YUV420P -> Y (w x h), U, V (w/2 x h/2) as they are
 *   targets   luma -> dst_w x dst_h; each chroma plane -> dst_w/2 x dst_h/2 (from full resolution for RGB input)
 *   filter    bicubic, B = 0, C = 0.6 (libswscale's SWS_BICUBIC defaults), widened by the scale factor when shrinking;
 *             positions as libswscale places them: xInc = ((src << 16) + dst / 2) / dst, centre of output i in source
 *             coordinates (16.16) c = i * xInc + ((xInc - 65536) >> 1); support R = 2 * max(65536, xInc);
 *             taps T = (2 R + 65535) >> 16; first tap left = floor((c - R) / 65536) + 1; weight of tap k from
 *             x = |((left + k) << 16) - c| * 65536 / max(65536, xInc)  (16.16, truncating division):
 *               x < 1:  W = (14 x^3 >> 32) - (24 x^2 >> 16) + (10 << 16)
 *               x < 2:  W = -(6 x^3 >> 32) + (30 x^2 >> 16) - 48 x + (24 << 16)          else 0
 *             coefficients coef_k = W_k * 16384 / sum(W) (truncating), the remainder added to the first largest tap: every
 *             row sums to exactly 16384.  Source indices are clamped to the plane (edge replication).
 *   pass 1    horizontal: t = clamp((sum coef_k * sample) >> 7, 0, 32767)            (15-bit intermediate)
 *   range     limited-range YUV input only, on the intermediates (libswscale's lumRangeToJpeg / chrRangeToJpeg constants):
 *               luma   t = (min(t, 30189) * 19077 - 39057361) >> 14
 *               chroma t = (min(t, 30775) *  4663 -  9289992) >> 12
 *   pass 2    vertical: out = clamp((sum coef_k * t + (1 << 20)) >> 21, 0, 255)
 *   output    NV21: Y plane, then rows of interleaved Cr, Cb (psxavenc/mdec.c:585-594,627-628)
 */
#include "frontend_oracle.h"

#include <stdlib.h>
#include <string.h>

static int64_t floor_div(int64_t a, int64_t b) {
	int64_t q = a / b;
	if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
	return q;
}

static int64_t cubic_w(int64_t x) {      /* x: 16.16, >= 0 */
	const int64_t one = 65536;
	if (x < one) return ((14 * x * x * x) >> 32) - ((24 * x * x) >> 16) + (10 * one);
	if (x < 2 * one) return -((6 * x * x * x) >> 32) + ((30 * x * x) >> 16) - 48 * x + 24 * one;
	return 0;
}

int orc_scaler_filter(int src, int dst, int *taps_out, int32_t *left, int16_t *coef, int cap) {
	if (src < 1 || dst < 1) return -1;
	const int64_t xinc = (((int64_t)src << 16) + dst / 2) / dst;
	const int64_t scale = xinc > 65536 ? xinc : 65536;
	const int64_t R = 2 * scale;
	const int taps = (int)((2 * R + 65535) >> 16);
	if (taps_out) *taps_out = taps;
	if (taps > 64 || (int64_t)dst * taps > cap) return -1;
	int64_t W[64];
	for (int i = 0; i < dst; i++) {
		const int64_t c = (int64_t)i * xinc + ((xinc - 65536) >> 1);
		const int64_t l = floor_div(c - R, 65536) + 1;
		int64_t sum = 0;
		int best = 0;
		for (int k = 0; k < taps; k++) {
			int64_t d = ((l + k) << 16) - c;
			if (d < 0) d = -d;
			W[k] = cubic_w(d * 65536 / scale);
			sum += W[k];
			if (W[k] > W[best]) best = k;
		}
		int64_t got = 0;
		for (int k = 0; k < taps; k++) {
			const int64_t q = W[k] * 16384 / sum;
			coef[(size_t)i * taps + k] = (int16_t)q;
			got += q;
		}
		coef[(size_t)i * taps + best] = (int16_t)(coef[(size_t)i * taps + best] + (16384 - got));
		left[i] = (int32_t)l;
	}
	return 0;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* one plane: src (sw x sh, row pitch sp) -> dst (dw x dh), written with pixel stride dstep and row pitch dp */
static int scale_plane(const uint8_t *src, int sw, int sh, int sp, uint8_t *dst, int dw, int dh, int dstep, int dp,
                       int range_mode /* 0 none, 1 luma, 2 chroma */) {
	int th, tv;
	int32_t *lh = malloc(sizeof(int32_t) * (size_t)dw), *lv = malloc(sizeof(int32_t) * (size_t)dh);
	int16_t *ch = malloc(sizeof(int16_t) * (size_t)dw * 64), *cv = malloc(sizeof(int16_t) * (size_t)dh * 64);
	int16_t *tmp = malloc(sizeof(int16_t) * (size_t)sh * dw);
	int rc = -1;
	if (lh && lv && ch && cv && tmp && orc_scaler_filter(sw, dw, &th, lh, ch, dw * 64) == 0 &&
	    orc_scaler_filter(sh, dh, &tv, lv, cv, dh * 64) == 0) {
		for (int y = 0; y < sh; y++)
			for (int i = 0; i < dw; i++) {
				int64_t acc = 0;
				for (int k = 0; k < th; k++) acc += (int64_t)ch[(size_t)i * th + k] * src[(size_t)y * sp + clampi(lh[i] + k, 0, sw - 1)];
				int t = clampi((int)(acc >> 7), 0, 32767);
				if (range_mode == 1) t = (int)((((int64_t)(t < 30189 ? t : 30189)) * 19077 - 39057361) >> 14);
				if (range_mode == 2) t = (int)((((int64_t)(t < 30775 ? t : 30775)) * 4663 - 9289992) >> 12);
				tmp[(size_t)y * dw + i] = (int16_t)t;
			}
		for (int j = 0; j < dh; j++)
			for (int i = 0; i < dw; i++) {
				int64_t acc = 0;
				for (int k = 0; k < tv; k++) acc += (int64_t)cv[(size_t)j * tv + k] * tmp[(size_t)clampi(lv[j] + k, 0, sh - 1) * dw + i];
				dst[(size_t)j * dp + (size_t)i * dstep] = (uint8_t)clampi((int)((acc + (1 << 20)) >> 21), 0, 255);
			}
		rc = 0;
	}
	free(lh); free(lv); free(ch); free(cv); free(tmp);
	return rc;
}

int orc_scaler_convert(int src_format, int src_w, int src_h, int src_full_range, int dst_w, int dst_h,
                       const uint8_t *src, uint8_t *out) {
	if (src_w < 2 || src_h < 2 || dst_w < 16 || dst_h < 16 || (dst_w % 16) || (dst_h % 16)) return -1;
	uint8_t *cplane = out + (size_t)dst_w * dst_h;
	if (src_format == ORC_PIX_YUV420P) {
		if ((src_w & 1) || (src_h & 1)) return -1;
		const uint8_t *Y = src, *U = src + (size_t)src_w * src_h, *V = U + (size_t)(src_w / 2) * (src_h / 2);
		const int lim = !src_full_range;
		if (scale_plane(Y, src_w, src_h, src_w, out, dst_w, dst_h, 1, dst_w, lim ? 1 : 0)) return -1;
		if (scale_plane(V, src_w / 2, src_h / 2, src_w / 2, cplane + 0, dst_w / 2, dst_h / 2, 2, dst_w, lim ? 2 : 0)) return -1;    /* Cr first */
		if (scale_plane(U, src_w / 2, src_h / 2, src_w / 2, cplane + 1, dst_w / 2, dst_h / 2, 2, dst_w, lim ? 2 : 0)) return -1;
		return 0;
	}
	if (src_format != ORC_PIX_RGB24) return -1;
	const size_t n = (size_t)src_w * src_h;
	uint8_t *planes = malloc(3 * n);
	if (!planes) return -1;
	for (size_t p = 0; p < n; p++) {
		const int r = src[3 * p], g = src[3 * p + 1], b = src[3 * p + 2];
		planes[p] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
		planes[n + p] = (uint8_t)clampi(((-11059 * r - 21709 * g + 32768 * b + 32768) >> 16) + 128, 0, 255);
		planes[2 * n + p] = (uint8_t)clampi(((32768 * r - 27439 * g - 5329 * b + 32768) >> 16) + 128, 0, 255);
	}
	int rc = scale_plane(planes, src_w, src_h, src_w, out, dst_w, dst_h, 1, dst_w, 0);
	if (!rc) rc = scale_plane(planes + 2 * n, src_w, src_h, src_w, cplane + 0, dst_w / 2, dst_h / 2, 2, dst_w, 0);
	if (!rc) rc = scale_plane(planes + n, src_w, src_h, src_w, cplane + 1, dst_w / 2, dst_h / 2, 2, dst_w, 0);
	free(planes);
	return rc;
}
