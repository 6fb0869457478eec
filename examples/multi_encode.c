/*
 * examples/multi_encode.c -- a C host program that encodes one batch of frames over a LIST of GPUs with one call
 * (psxhip_mdec_multi_*, include/psxav_hip.h): what the reference's single loop over frames (psxavenc/filefmt.c:633-662)
 * becomes when several devices are there.  Host code stays C; no ranks, no launcher.
 *
 *   gcc -std=c11 -O2 -I../include multi_encode.c -L../psxavenc_amd -lpsxav_hip -Wl,-rpath,$PWD/../psxavenc_amd -o multi_encode
 *   ./multi_encode 0,1,2,3 [frames] [width] [height] [budget] [codec] [static|tickets]
 *
 * Encodes the batch on the first listed device alone, then over the whole list, and compares bytes and results.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "psxav_hip.h"

static double now(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void make_frame(uint8_t *nv21, int w, int h, int index) {
	uint32_t lcg = 777u + 131u * (uint32_t)index;
	const int amp = (index & 64) ? 8 : 4;          /* runs of cheaper and of more expensive frames */
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) {
			lcg = lcg * 1664525u + 1013904223u;
			int v = ((x + index) % w) * 255 / w / 2 + y * 255 / h / 2 + (int)((lcg >> 24) % (unsigned)(2 * amp + 1)) - amp;
			nv21[y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
		}
	uint8_t *c = nv21 + w * h;
	for (int y = 0; y < h / 2; y++)
		for (int x = 0; x < w / 2; x++) {
			c[y * w + 2 * x + 0] = (uint8_t)(100 + x * 56 / (w / 2));
			c[y * w + 2 * x + 1] = (uint8_t)(156 - y * 56 / (h / 2));
		}
}

int main(int argc, char **argv) {
	int devices[PSXHIP_MULTI_MAX_REPORT], nd = 0;
	char list[256];
	snprintf(list, sizeof list, "%s", argc > 1 ? argv[1] : "0");
	for (char *t = strtok(list, ","); t && nd < PSXHIP_MULTI_MAX_REPORT; t = strtok(NULL, ",")) devices[nd++] = atoi(t);
	const int n = argc > 2 ? atoi(argv[2]) : 2000, w = argc > 3 ? atoi(argv[3]) : 320, h = argc > 4 ? atoi(argv[4]) : 240;
	const int budget = argc > 5 ? atoi(argv[5]) : 8192, codec = argc > 6 ? atoi(argv[6]) : 0;
	const int schedule = (argc > 7 && !strcmp(argv[7], "tickets")) ? PSXHIP_SCHED_TICKETS : PSXHIP_SCHED_STATIC;
	const size_t fsz = (size_t)w * h * 3 / 2;
	uint8_t *frames = malloc(fsz * (size_t)n), *one = calloc((size_t)n, (size_t)budget), *all = calloc((size_t)n, (size_t)budget);
	psxhip_mdec_result_t *r1 = calloc((size_t)n, sizeof *r1), *rn = calloc((size_t)n, sizeof *rn);
	if (!frames || !one || !all || !r1 || !rn) return 2;
	for (int i = 0; i < n; i++) make_frame(frames + fsz * (size_t)i, w, h, i);

	psxhip_mdec_multi_t *single = NULL, *multi = NULL;
	if (psxhip_mdec_multi_create(&single, devices, 1, codec, w, h, budget) ||
	    psxhip_mdec_multi_create(&multi, devices, nd, codec, w, h, budget)) {
		fprintf(stderr, "create: %s\n", psxhip_last_error());
		return 1;
	}
	psxhip_multi_report_t rep[PSXHIP_MULTI_MAX_REPORT];
	for (int pass = 0; pass < 2; pass++) {           /* pass 0 warms the pinned staging buffers up */
		double t0 = now();
		if (psxhip_mdec_multi_encode_frames_host(single, frames, n, NULL, budget, one, (size_t)budget, r1, PSXHIP_SCHED_STATIC, 0, NULL)) {
			fprintf(stderr, "single: %s\n", psxhip_last_error());
			return 1;
		}
		double t1 = now();
		if (psxhip_mdec_multi_encode_frames_host(multi, frames, n, NULL, budget, all, (size_t)budget, rn, schedule, 0, rep)) {
			fprintf(stderr, "multi: %s\n", psxhip_last_error());
			return 1;
		}
		double t2 = now();
		if (pass) {
			printf("%d frames %dx%d budget %d: one device %.0f frames/s, %d devices (%s) %.0f frames/s\n", n, w, h, budget,
			       n / (t1 - t0), nd, schedule == PSXHIP_SCHED_TICKETS ? "tickets" : "static", n / (t2 - t1));
			for (int d = 0; d < nd; d++)
				printf("  worker %d: device %d, %lld frames in %d range(s), %.2f ms\n", d, rep[d].device, (long long)rep[d].units,
				       rep[d].tickets, rep[d].seconds * 1e3);
		}
	}
	const int same = !memcmp(one, all, (size_t)n * (size_t)budget) && !memcmp(r1, rn, (size_t)n * sizeof *r1);
	printf("identical to the single-device call: %s\n", same ? "yes" : "NO");
	psxhip_mdec_multi_destroy(single);
	psxhip_mdec_multi_destroy(multi);
	free(frames); free(one); free(all); free(r1); free(rn);
	return same ? 0 : 1;
}
