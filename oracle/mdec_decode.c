/*
 * oracle/mdec_decode.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A reader for the BS v2/v3 bitstream the encoder emits, used for size-independent
 * round-trip properties (encode -> decode -> compare levels / reconstruct pixels).  The
 * reference has no decoder; the format facts come from its encoder (mdec.c:321-333 word
 * order, :441-510 block syntax, :647-651,710 end codes, :738-754 header) and the dequantiser
 * is the inverse of its quantiser (level * quant * scale / 8, DC * 2).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "bs_vlc_tables.h"
#include "mdec_oracle.h"

typedef struct {
	const uint8_t *p;
	int64_t nbits, pos;
} bitsrc_t;

static inline int get1(bitsrc_t *s) {
	if (s->pos >= s->nbits) { s->pos++; return 0; }
	const int64_t w = s->pos >> 4;
	const int b = 15 - (int)(s->pos & 15);
	const unsigned word = (unsigned)s->p[2 * w] | ((unsigned)s->p[2 * w + 1] << 8);
	s->pos++;
	return (int)((word >> b) & 1u);
}
static inline uint32_t getn(bitsrc_t *s, int n) {
	uint32_t v = 0;
	while (n--) v = (v << 1) | (uint32_t)get1(s);
	return v;
}
static int match(bitsrc_t *s, const char *bits) {
	const int64_t save = s->pos;
	for (const char *c = bits; *c; c++)
		if (get1(s) != (*c - '0')) { s->pos = save; return 0; }
	return 1;
}

static int read_dc_delta(bitsrc_t *s, int luma, int *delta) {
	const char *zero = luma ? ORC_DC_LUMA_ZERO : ORC_DC_CHROMA_ZERO;
	const char *const *pre = luma ? orc_dc_luma_prefix : orc_dc_chroma_prefix;
	/* the codes are prefix-free, try longest table first is unnecessary: try all */
	if (match(s, zero)) { *delta = 0; return 1; }
	for (int m = 0; m < 8; m++)
		if (match(s, pre[m])) {
			const int positive = get1(s);
			const int j = (int)getn(s, m);
			*delta = positive ? j + (1 << m) : j - ((2 << m) - 1);
			return 1;
		}
	return 0;
}

/* returns 1 = coefficient read, 2 = end of block, 0 = syntax error */
static int read_ac(bitsrc_t *s, int *run, int *level) {
	if (match(s, "10")) return 2;
	if (match(s, "000001")) {
		*run = (int)getn(s, 6);
		const int l = (int)getn(s, 10);
		*level = l >= 512 ? l - 1024 : l;
		return 1;
	}
	for (int i = 0; i < ORC_AC_CODE_COUNT; i++)
		if (match(s, orc_ac_codes[i].bits)) {
			const int neg = get1(s);
			*run = orc_ac_codes[i].run;
			*level = neg ? -(int)orc_ac_codes[i].level : (int)orc_ac_codes[i].level;
			return 1;
		}
	return 0;
}

int orc_mdec_decode_frame(int w, int h, const uint8_t *bs, int bs_size, int16_t *levels,
                          int *quant_scale, int *version, int *bits_consumed, int v3dc_wrap) {
	const int nblk = (w / 16) * (h / 16) * 6;
	if (bs_size < 8 || bs[2] != 0x00 || bs[3] != 0x38) return -1;
	*quant_scale = bs[4] | (bs[5] << 8);
	*version = bs[6] | (bs[7] << 8);
	if (*version != 2 && *version != 3) return -2;
	bitsrc_t s = {bs + 8, (int64_t)(bs_size - 8) * 8, 0};
	int last_dc[3] = {0, 0, 0};
	memset(levels, 0, (size_t)nblk * 64 * sizeof(int16_t));

	for (int b = 0; b < nblk; b++) {
		int16_t *out = levels + (size_t)b * 64;
		const int comp = (b % 6) < 2 ? (b % 6) : 2;
		if (*version == 2) {
			const int v = (int)getn(&s, 10);
			if (v == 0x1FF) return -3;               /* premature end of frame */
			out[0] = (int16_t)(v >= 512 ? v - 1024 : v);
		} else {
			int delta;
			if (!read_dc_delta(&s, comp == 2, &delta)) return -4;
			last_dc[comp] += delta * 4;
			if (v3dc_wrap) last_dc[comp] = ((last_dc[comp] + 512) & 0x3FF) - 512;   /* decoder-side 10-bit wrap, mdec.c:463-468 */
			out[0] = (int16_t)last_dc[comp];
		}
		int k = 0;
		for (;;) {
			int run, level;
			const int r = read_ac(&s, &run, &level);
			if (r == 0) return -5;
			if (r == 2) break;
			k += run + 1;
			if (k > 63) return -6;
			out[k] = (int16_t)level;
		}
	}
	const uint32_t eof = getn(&s, 10);
	if (eof != (*version == 2 ? 0x1FFu : 0x3FFu)) return -7;
	if (s.pos > s.nbits) return -8;
	*bits_consumed = (int)s.pos;
	return 0;
}

void orc_mdec_reconstruct(int w, int h, const int16_t *levels, int quant_scale, uint8_t *nv21) {
	const int nx = w / 16, ny = h / 16;
	static double cs[8][8];
	static int init;
	if (!init) {
		for (int u = 0; u < 8; u++)
			for (int x = 0; x < 8; x++)
				cs[u][x] = (u == 0 ? sqrt(0.125) : 0.5) * cos((2 * x + 1) * u * M_PI / 16.0);
		init = 1;
	}
	int b = 0;
	for (int fx = 0; fx < nx; fx++)
		for (int fy = 0; fy < ny; fy++)
			for (int i = 0; i < 6; i++, b++) {
				const int16_t *lv = levels + (size_t)b * 64;
				double F[64], px[64];
				for (int z = 0; z < 64; z++) {
					const int ri = orc_zagzig[z];
					F[ri] = z == 0 ? lv[0] * 2.0 : lv[z] * (double)orc_quant_matrix[ri] * quant_scale / 8.0;
				}
				for (int y = 0; y < 8; y++)
					for (int x = 0; x < 8; x++) {
						double acc = 0;
						for (int v = 0; v < 8; v++)
							for (int u = 0; u < 8; u++) acc += F[v * 8 + u] * cs[v][y] * cs[u][x];
						px[y * 8 + x] = acc + 128.0;
					}
				for (int y = 0; y < 8; y++)
					for (int x = 0; x < 8; x++) {
						int v = (int)lrint(px[y * 8 + x]);
						v = v < 0 ? 0 : (v > 255 ? 255 : v);
						uint8_t *dst;
						if (i == 0) dst = nv21 + w * h + w * (fy * 8 + y) + 2 * (fx * 8 + x);
						else if (i == 1) dst = nv21 + w * h + w * (fy * 8 + y) + 2 * (fx * 8 + x) + 1;
						else dst = nv21 + w * (fy * 16 + ((i - 2) >> 1) * 8 + y) + fx * 16 + ((i - 2) & 1) * 8 + x;
						*dst = (uint8_t)v;
					}
			}
}
