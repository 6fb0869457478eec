/*
 * examples/percall_bench.c -- what the reference's OWN call pattern costs per call on the drop-in surface:
 *   psx_audio_spu_encode once per 28 samples   (psxavenc/filefmt.c:243)
 *   psx_audio_xa_encode once per sector        (filefmt.c:184)
 *   encode_frame_bs once per frame             (filefmt.c:643)
 * Plain C against libpsxav_hip.so; synthetic input (integer generators); prints one JSON object with the microseconds per call
 * (median and mean over the timed calls).  The bytes themselves are checked by tests/test_gpu_dropin.py and tests/test_gpu_adpcm.py.
 *
 *   ./percall_bench [spu_calls] [xa_calls] [frame_calls] [sweep]
 * sweep != 0: also psx_audio_spu_encode at 28 .. 114 688 samples per call (state carried), microseconds per call -- where the per-call
 * drop-in starts to win over the CPU it replaces (the CPU side: oracle/cpu_bench spucall).
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "psxav_audio.h"
#include "psxav_hip.h"
#include "psxav_mdec.h"

static double now_us(void) {
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
static int cmp_d(const void *a, const void *b) { return *(const double *)a < *(const double *)b ? -1 : *(const double *)a > *(const double *)b; }
static void stats(double *v, int n, double *median, double *mean) {
	double s = 0;
	for (int i = 0; i < n; i++) s += v[i];
	qsort(v, (size_t)n, sizeof *v, cmp_d);
	*median = v[n / 2];
	*mean = s / n;
}

static void make_frame(uint8_t *nv21, int w, int h, int index) {
	uint32_t lcg = 12345u + 977u * (uint32_t)index;
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) {
			lcg = lcg * 1664525u + 1013904223u;
			int v = ((x + 2 * index) % w) * 255 / w / 2 + y * 255 / h / 2 + (int)((lcg >> 24) % 9) - 4;
			nv21[y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
		}
	uint8_t *c = nv21 + w * h;
	for (int y = 0; y < h / 2; y++)
		for (int x = 0; x < w / 2; x++) {
			c[y * w + 2 * x + 0] = (uint8_t)(96 + x * 64 / (w / 2));
			c[y * w + 2 * x + 1] = (uint8_t)(160 - y * 64 / (h / 2));
		}
}

int main(int argc, char **argv) {
	const int n_spu = argc > 1 ? atoi(argv[1]) : 4000, n_xa = argc > 2 ? atoi(argv[2]) : 400, n_fr = argc > 3 ? atoi(argv[3]) : 400;
	const int warm = 50;
	double med, mean;
	printf("{");

	/* ---- SPU: 28 samples per call, state carried */
	{
		const int total = (n_spu + warm) * 28;
		int16_t *pcm = malloc(sizeof(int16_t) * (size_t)total);
		uint32_t lcg = 1;
		for (int i = 0; i < total; i++) {
			lcg = lcg * 1664525u + 1013904223u;
			pcm[i] = (int16_t)(9000.0 * sin(i * 0.031) + 4000.0 * sin(i * 0.173) + (int)((lcg >> 20) % 801) - 400);
		}
		double *t = malloc(sizeof(double) * (size_t)n_spu);
		psx_audio_encoder_channel_state_t st;
		memset(&st, 0, sizeof st);
		uint8_t blk[16];
		unsigned sum = 0;
		for (int k = 0; k < n_spu + warm; k++) {
			const double a = now_us();
			psx_audio_spu_encode(&st, pcm + k * 28, 28, 1, blk);
			if (k >= warm) t[k - warm] = now_us() - a;
			sum += blk[0];
		}
		stats(t, n_spu, &med, &mean);
		printf("\"psx_audio_spu_encode_28_samples\": {\"us_per_call_median\": %.2f, \"us_per_call_mean\": %.2f, \"calls\": %d, \"header_sum\": %u}", med, mean, n_spu, sum);
		free(t);
		free(pcm);
	}
	/* ---- XA: one 37800 Hz 4-bit stereo XACD sector (2016 sample frames) per call */
	{
		const int sps = 2016, total = (n_xa + warm) * sps + 4032;
		int16_t *pcm = calloc((size_t)total * 2, sizeof(int16_t));
		uint32_t lcg = 7;
		for (int i = 0; i < (n_xa + warm) * sps; i++) {
			lcg = lcg * 1664525u + 1013904223u;
			pcm[2 * i] = (int16_t)(9000.0 * sin(i * 0.021) + (int)((lcg >> 20) % 601) - 300);
			pcm[2 * i + 1] = (int16_t)(7000.0 * sin(i * 0.047) + (int)((lcg >> 8) % 601) - 300);
		}
		psx_audio_xa_settings_t s = {PSX_AUDIO_XA_FORMAT_XACD, true, PSX_AUDIO_XA_FREQ_DOUBLE, 4, 1, 0};
		psx_audio_encoder_state_t st;
		memset(&st, 0, sizeof st);
		uint8_t sec[2352];
		double *t = malloc(sizeof(double) * (size_t)n_xa);
		unsigned sum = 0;
		for (int k = 0; k < n_xa + warm; k++) {
			memset(sec, 0, sizeof sec);
			const double a = now_us();
			psx_audio_xa_encode(s, &st, pcm + (size_t)k * sps * 2, sps, k, sec);
			if (k >= warm) t[k - warm] = now_us() - a;
			sum += sec[0x18];
		}
		stats(t, n_xa, &med, &mean);
		printf(", \"psx_audio_xa_encode_sector\": {\"us_per_call_median\": %.2f, \"us_per_call_mean\": %.2f, \"calls\": %d, \"header_sum\": %u}", med, mean, n_xa, sum);
		free(t);
		free(pcm);
	}
	/* ---- MDEC: one 320x240 v2 frame per call, 8192-byte budget */
	{
		const int w = 320, h = 240, budget = 8192, distinct = 16;
		uint8_t *fr = malloc((size_t)distinct * w * h * 3 / 2);
		for (int i = 0; i < distinct; i++) make_frame(fr + (size_t)i * w * h * 3 / 2, w, h, i);
		mdec_encoder_t enc;
		memset(&enc, 0, sizeof enc);
		if (!init_mdec_encoder(&enc, BS_CODEC_V2, w, h)) {
			printf(", \"error\": \"init_mdec_encoder failed: %s\"}\n", psxhip_last_error());
			return 1;
		}
		enc.state.frame_output = malloc((size_t)budget);
		enc.state.frame_max_size = budget;
		double *t = malloc(sizeof(double) * (size_t)n_fr);
		for (int k = 0; k < n_fr + warm; k++) {
			const double a = now_us();
			encode_frame_bs(&enc, fr + (size_t)(k % distinct) * w * h * 3 / 2);
			if (k >= warm) t[k - warm] = now_us() - a;
		}
		stats(t, n_fr, &med, &mean);
		printf(", \"encode_frame_bs_320x240_v2\": {\"us_per_call_median\": %.2f, \"us_per_call_mean\": %.2f, \"calls\": %d, \"frames_per_sec\": %.1f, \"quant_scale_sum\": %d}",
		       med, mean, n_fr, 1e6 / mean, enc.state.quant_scale_sum);
		free(t);
		free(enc.state.frame_output);
		destroy_mdec_encoder(&enc);
		free(fr);
	}
	if (argc > 4 && atoi(argv[4])) {
		static const int sizes[] = {28, 56, 112, 224, 448, 896, 1792, 3584, 7168, 14336, 22064, 28672, 57344, 114688};
		const int ns = (int)(sizeof sizes / sizeof sizes[0]);
		int16_t *pcm = malloc(sizeof(int16_t) * (size_t)(114688 + 28));
		uint32_t lcg = 3;
		for (int i = 0; i < 114688 + 28; i++) {
			lcg = lcg * 1664525u + 1013904223u;
			pcm[i] = (int16_t)(9000.0 * sin(i * 0.031) + 4000.0 * sin(i * 0.173) + (int)((lcg >> 20) % 801) - 400);
		}
		uint8_t *out = malloc((size_t)(114688 / 28 + 2) * 16);
		printf(", \"psx_audio_spu_encode_by_samples_per_call\": {");
		for (int k = 0; k < ns; k++) {
			const int n = sizes[k], reps = n <= 3584 ? 300 : 60;
			double *t = malloc(sizeof(double) * (size_t)reps);
			psx_audio_encoder_channel_state_t st;
			memset(&st, 0, sizeof st);
			for (int r = 0; r < reps + 10; r++) {
				const double a = now_us();
				psx_audio_spu_encode(&st, pcm, n, 1, out);
				if (r >= 10) t[r - 10] = now_us() - a;
			}
			stats(t, reps, &med, &mean);
			printf("%s\"%d\": %.2f", k ? ", " : "", n, med);
			free(t);
		}
		printf("}");
		free(out);
		free(pcm);
	}
	printf(", \"library\": \"%s\"}\n", psxhip_version());
	return 0;
}
