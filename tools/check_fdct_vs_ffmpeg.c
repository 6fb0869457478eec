/*
 * tools/check_fdct_vs_ffmpeg.c -- OFF-BOX pin kit for the one piece of the MDEC path this repository could not pin.
 *
 * psxavenc takes its 8x8 forward DCT from FFmpeg: avcodec_dct_alloc() / avcodec_dct_init() / AVDCT.fdct
 * (psxavenc/mdec.c:524,548,640; libavcodec 8.0.1 configured --disable-mmx in the release CI,
 * .github/scripts/build.sh:4,36-56).  libavcodec is neither part of the reference tree nor installed in the build
 * image, so oracle/mdec_oracle.c RESTATES the routine that configuration selects (ff_jpeg_fdct_islow_8) and every MDEC
 * parity claim of this repository is "bit-exact to psxavenc/mdec.c with that FDCT".  Anyone with FFmpeg can close the
 * gap with this program: it runs the real AVDCT.fdct on seeded blocks and compares
 *     (1) orc_fdct_islow8()        -- the oracle's restatement          (always)
 *     (2) psxhip_mdec_fdct_host()  -- the device arithmetic             (when built with -DWITH_DEVICE on an MI355X)
 * and reports which libavcodec it ran against.  A stock x86-64 FFmpeg (MMX/SSE2 enabled) selects ff_fdct_sse2 for
 * AVDCT's default dct_algo and WILL differ in the low bits -- that is the reference's own platform dependence
 * (SURVEY H1), not a defect here; pass "islow" as first argument to force dct_algo = FF_DCT_INT... see below.
 *
 * Build (not possible in the build image: no FFmpeg):
 *   gcc -O2 -I oracle tools/check_fdct_vs_ffmpeg.c oracle/mdec_oracle.c oracle/mdec_decode.c -lavcodec -lavutil -lm -o check_fdct
 *   gcc -O2 -DWITH_DEVICE -I oracle -I include tools/check_fdct_vs_ffmpeg.c oracle/mdec_oracle.c oracle/mdec_decode.c \
 *       -L psxavenc_amd -lpsxav_hip -Wl,-rpath,$PWD/psxavenc_amd -lavcodec -lavutil -lm -o check_fdct
 * Run:   ./check_fdct [auto|islow] [n_blocks] [seed] [file to write AVDCT's output vector to]      (tools/pin_fdct.py does both)
 * Exit status 0 = every block identical in every comparison.
 */
#include <libavcodec/avcodec.h>
#include <libavcodec/avdct.h>
#include <libavcodec/version.h>
#include <libavutil/mem.h>
#include <libavutil/opt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mdec_oracle.h"
#ifdef WITH_DEVICE
#include "psxav_hip.h"
#endif

static uint32_t rng_state;
static uint32_t rng(void) {
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 17;
	rng_state ^= rng_state << 5;
	return rng_state;
}

/* block classes: flat, ramps, checkerboards, full-range noise, small noise, extremes -- all in -128..127 */
static void make_block(int16_t *b, int kind) {
	const int base = (int)(rng() % 256) - 128;
	for (int y = 0; y < 8; y++)
		for (int x = 0; x < 8; x++) {
			int v;
			switch (kind % 7) {
			case 0: v = base; break;
			case 1: v = base + (x - 4) * (int)(rng() % 9) + (y - 4) * 3; break;
			case 2: v = ((x ^ y) & 1) ? 127 : -128; break;
			case 3: v = (int)(rng() % 256) - 128; break;
			case 4: v = base + (int)(rng() % 9) - 4; break;
			case 5: v = (rng() & 1) ? 127 : -128; break;
			default: v = (x < 4) == (y < 4) ? -128 : 127; break;
			}
			b[y * 8 + x] = (int16_t)(v < -128 ? -128 : (v > 127 ? 127 : v));
		}
}

int main(int argc, char **argv) {
	const char *algo = argc > 1 ? argv[1] : "auto";
	const int n = argc > 2 ? atoi(argv[2]) : 200000;
	rng_state = argc > 3 ? (uint32_t)strtoul(argv[3], NULL, 0) : 0x9E3779B9u;
	if (!rng_state) rng_state = 1;

	AVDCT *dct = avcodec_dct_alloc();                     /* mdec.c:524 */
	if (!dct) return 2;
	if (!strcmp(algo, "islow")) av_opt_set_int(dct, "dct", FF_DCT_INT, 0);   /* the C "accurate integer" routine */
	if (avcodec_dct_init(dct) < 0) return 2;              /* mdec.c:548: all defaults */
	printf("libavcodec %s (%u.%u.%u), configuration: %s\n", LIBAVCODEC_IDENT, LIBAVCODEC_VERSION_MAJOR,
	       LIBAVCODEC_VERSION_MINOR, LIBAVCODEC_VERSION_MICRO, avcodec_configuration());
	printf("dct_algo requested: %s, bits_per_sample %d\n", algo, dct->bits_per_sample);

	int16_t *in = malloc((size_t)n * 64 * sizeof(int16_t));
	int16_t *ff = malloc((size_t)n * 64 * sizeof(int16_t));
	int16_t *orc = malloc((size_t)n * 64 * sizeof(int16_t));
	for (int i = 0; i < n; i++) make_block(in + (size_t)i * 64, i);
	memcpy(ff, in, (size_t)n * 64 * sizeof(int16_t));
	memcpy(orc, in, (size_t)n * 64 * sizeof(int16_t));
	for (int i = 0; i < n; i++) {
		/* AVDCT wants 16-byte aligned blocks */
		int16_t tmp[64] __attribute__((aligned(32)));
		memcpy(tmp, ff + (size_t)i * 64, sizeof tmp);
		dct->fdct(tmp);                                   /* mdec.c:640 */
		memcpy(ff + (size_t)i * 64, tmp, sizeof tmp);
		orc_fdct_islow8(orc + (size_t)i * 64);
	}
	long bad_orc = 0, max_orc = 0;
	for (size_t k = 0; k < (size_t)n * 64; k++) {
		const long d = labs((long)ff[k] - (long)orc[k]);
		if (d) bad_orc++;
		if (d > max_orc) max_orc = d;
	}
	printf("AVDCT.fdct vs oracle restatement (orc_fdct_islow8): %ld of %zu coefficients differ, max |diff| %ld  -> %s\n",
	       bad_orc, (size_t)n * 64, max_orc, bad_orc ? "NOT PINNED (is this build's fdct ff_jpeg_fdct_islow_8? try `islow`)" : "PINNED");
	int rc = bad_orc ? 1 : 0;
	if (argc > 4) {           /* the vector itself, for a hash of what was compared (tools/pin_fdct.py prints its SHA-256) */
		FILE *fh = fopen(argv[4], "wb");
		if (fh) {
			fwrite(ff, sizeof(int16_t), (size_t)n * 64, fh);
			fclose(fh);
		}
	}
	if (bad_orc) {
		/* which instance of the IJG butterfly is it then?  (2 extra bits after the row pass = the IJG original, what libjpeg
		 * ships; 4 = libavcodec's jfdctint_template.c for 8-bit samples, what the oracle assumes) */
		for (int p1 = 1; p1 <= 5; p1++) {
			long bad = 0;
			for (int i = 0; i < n; i++) {
				int16_t t[64];
				memcpy(t, in + (size_t)i * 64, sizeof t);
				orc_fdct_islow8_pass1(t, p1);
				for (int k = 0; k < 64; k++) bad += t[k] != ff[(size_t)i * 64 + k];
			}
			printf("  ... against the same butterfly with %d extra bits after the row pass: %ld coefficients differ%s\n", p1, bad,
			       bad ? "" : "  <- this build's fdct");
		}
	}
#ifdef WITH_DEVICE
	int16_t *dev = malloc((size_t)n * 64 * sizeof(int16_t));
	if (psxhip_mdec_fdct_host(0, in, n, dev) != 0) {
		printf("device FDCT failed: %s\n", psxhip_last_error());
		return 2;
	}
	long bad_dev = 0;
	for (size_t k = 0; k < (size_t)n * 64; k++) bad_dev += ff[k] != dev[k];
	printf("AVDCT.fdct vs device FDCT (psxhip_mdec_fdct_host, %s): %ld coefficients differ -> %s\n", psxhip_version(), bad_dev,
	       bad_dev ? "NOT PINNED" : "PINNED");
	rc |= bad_dev ? 1 : 0;
#endif
	av_free(dct);                                        /* mdec.c:557 */
	return rc;
}
