/* oracle/cpu_bench.c -- TEST INFRASTRUCTURE ONLY: the CPU baseline of bench.py, timed on N host cores.
 *
 * One encoder per thread over disjoint inputs, which is legal for the reference too: neither psxavenc/mdec.c nor
 * libpsxav/adpcm.c has globals.  A plain C pthread harness, because the same loop driven from Python
 * threads scaled 10x on 256 cores (GIL + per-call marshalling) and understated the CPU by an order of magnitude.
 *
 *   cpu_bench mdec <threads> <seconds> <codec> <w> <h> <budget> <amp> <seed>
 *       oracle/mdec_oracle.c (this repository's restatement of psxavenc/mdec.c:580-755; the FFmpeg-linked reference
 *       cannot be built here).  Every thread generates its own 16 frames (orc_synth_frame, same generator as the GPU
 *       bench) and encodes them round-robin until the time is up.
 *   cpu_bench xa <threads> <seconds> <seed> [ref.so]
 *       37800 Hz 4-bit stereo XACD sectors: the reference's own psx_audio_xa_encode when the path of oracle/_ref's
 *       library is given (libpsxav/adpcm.c compiled unchanged), else oracle/adpcm_oracle.c.  Every thread encodes its own
 *       40 sectors of kind-0 PCM, state carried, round-robin.
 * Prints one JSON object: units (frames / sectors) per second over all threads, and per thread min / max.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "adpcm_oracle.h"
#include "mdec_oracle.h"

static double now(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* the reference's structs (libpsxav/libpsxav.h:44-62), for calling oracle/_ref through dlsym */
typedef struct { int format; bool stereo; int frequency, bits_per_sample, file_number, channel_number; } ref_xa_settings_t;
typedef struct { int qerr; uint64_t mse; int prev1, prev2; } ref_chan_t;
typedef struct { ref_chan_t left, right; } ref_state_t;
typedef int (*ref_xa_encode_fn)(ref_xa_settings_t, ref_state_t *, const int16_t *, int, int, uint8_t *);

typedef struct {
	int mode, index;
	int codec, w, h, budget, amp;
	uint32_t seed;
	double seconds;
	ref_xa_encode_fn ref_xa;
	pthread_barrier_t *start;
	long units;
	double elapsed;
	int failed;
} job_t;

enum { FRAMES_PER_THREAD = 16, SECTORS_PER_THREAD = 40 };

static void *work(void *p) {
	job_t *j = (job_t *)p;
	if (j->mode == 0) {
		const size_t fsz = (size_t)j->w * j->h * 3 / 2;
		uint8_t *frames = malloc(fsz * FRAMES_PER_THREAD), *out = malloc((size_t)j->budget);
		for (int i = 0; i < FRAMES_PER_THREAD; i++)
			orc_synth_frame(j->w, j->h, j->seed, (uint32_t)(j->index * FRAMES_PER_THREAD + i), j->amp, frames + fsz * i);
		pthread_barrier_wait(j->start);
		const double t0 = now();
		long n = 0;
		do {
			orc_mdec_result_t r;
			if (orc_mdec_encode_frame(j->codec, j->w, j->h, frames + fsz * (n % FRAMES_PER_THREAD), j->budget, out, &r)) j->failed = 1;
			n++;
		} while (now() - t0 < j->seconds);
		j->elapsed = now() - t0;
		j->units = n;
		free(frames);
		free(out);
	} else {
		const int sps = 2016, total = sps * SECTORS_PER_THREAD;
		int16_t *pcm = calloc((size_t)(total + 4032) * 2, sizeof(int16_t));
		int16_t *tmp = malloc((size_t)total * sizeof(int16_t));
		for (int c = 0; c < 2; c++) {
			orc_synth_pcm(j->seed, (uint32_t)(2 * j->index + c), 0, total, 0, tmp);
			for (int i = 0; i < total; i++) pcm[2 * i + c] = tmp[i];
		}
		free(tmp);
		uint8_t sector[2352];
		orc_xa_settings_t os = {1, 1, 37800, 4, 1, 0};
		ref_xa_settings_t rs = {1, true, 37800, 4, 1, 0};
		orc_adpcm_state_t ost;
		ref_state_t rst;
		memset(&ost, 0, sizeof ost);
		memset(&rst, 0, sizeof rst);
		pthread_barrier_wait(j->start);
		const double t0 = now();
		long n = 0;
		do {
			const int16_t *src = pcm + (size_t)(n % SECTORS_PER_THREAD) * sps * 2;
			const int len = j->ref_xa ? j->ref_xa(rs, &rst, src, sps, (int)n, sector) : orc_xa_encode(os, &ost, src, sps, (int)n, sector);
			if (len != 2352) j->failed = 1;
			n++;
		} while (now() - t0 < j->seconds);
		j->elapsed = now() - t0;
		j->units = n;
		free(pcm);
	}
	return NULL;
}

int main(int argc, char **argv) {
	if (argc < 5) {
		fprintf(stderr, "usage: cpu_bench mdec <threads> <seconds> <codec> <w> <h> <budget> <amp> <seed>\n"
		                "       cpu_bench xa <threads> <seconds> <seed> [path to oracle/_ref/libpsxav_ref.so]\n");
		return 2;
	}
	const int mode = strcmp(argv[1], "xa") == 0;
	const int threads = atoi(argv[2]);
	const double seconds = atof(argv[3]);
	if (threads < 1 || threads > 4096 || seconds <= 0) return 2;
	job_t proto;
	memset(&proto, 0, sizeof proto);
	proto.mode = mode;
	proto.seconds = seconds;
	const char *kind = "port";
	if (!mode) {
		if (argc < 10) return 2;
		proto.codec = atoi(argv[4]); proto.w = atoi(argv[5]); proto.h = atoi(argv[6]);
		proto.budget = atoi(argv[7]); proto.amp = atoi(argv[8]); proto.seed = (uint32_t)strtoul(argv[9], NULL, 0);
	} else {
		proto.seed = (uint32_t)strtoul(argv[4], NULL, 0);
		if (argc > 5) {
			void *h = dlopen(argv[5], RTLD_NOW | RTLD_LOCAL);
			if (h) proto.ref_xa = (ref_xa_encode_fn)dlsym(h, "psx_audio_xa_encode");
			if (proto.ref_xa) kind = "reference";
		}
	}
	(void)orc_mdec_ac_code(0, 1);        /* build the oracle's lazily built VLC tables before the threads start */
	pthread_barrier_t start;
	pthread_barrier_init(&start, NULL, (unsigned)threads);
	job_t *jobs = calloc((size_t)threads, sizeof(job_t));
	pthread_t *th = calloc((size_t)threads, sizeof(pthread_t));
	for (int i = 0; i < threads; i++) {
		jobs[i] = proto;
		jobs[i].index = i;
		jobs[i].start = &start;
		if (pthread_create(&th[i], NULL, work, &jobs[i])) return 3;
	}
	long total = 0;
	double rate = 0, lo = 1e30, hi = 0, longest = 0;
	int failed = 0;
	for (int i = 0; i < threads; i++) {
		pthread_join(th[i], NULL);
		const double r = (double)jobs[i].units / jobs[i].elapsed;
		total += jobs[i].units;
		rate += r;
		if (r < lo) lo = r;
		if (r > hi) hi = r;
		if (jobs[i].elapsed > longest) longest = jobs[i].elapsed;
		failed |= jobs[i].failed;
	}
	printf("{\"mode\": \"%s\", \"kind\": \"%s\", \"threads\": %d, \"units\": %ld, \"seconds\": %.3f, \"units_per_sec\": %.2f, "
	       "\"per_thread_min\": %.2f, \"per_thread_max\": %.2f, \"failed\": %d}\n",
	       mode ? "xa" : "mdec", kind, threads, total, longest, rate, lo, hi, failed);
	return failed ? 1 : 0;
}
