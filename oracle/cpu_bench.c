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
 *   cpu_bench str <threads> <seconds> <seed> [ref.so]
 *       config 3 (`strcd v2`): the sector loop of encode_file_str (psxavenc/filefmt.c:450-503) in C -- 320x240 @15 fps BS v2 through
 *       orc_mdec_encode_sector_str (the restatement of mdec.c:757-836), one 37800 Hz 4-bit stereo XA sector in eight (the
 *       reference's own psx_audio_xa_encode when ref.so is given), mode-2 form-1 headers and EDC around the video sectors.  Every
 *       thread muxes its own 24 frames + their audio over and over.  (bench.py timed this loop in Python until round 5; the
 *       interpreter was in the number.)
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
	} else if (j->mode == 2) {
		enum { W = 320, H = 240, NF = 24, SPS = 2016, INTERLEAVE = 8 };
		const size_t fsz = (size_t)W * H * 3 / 2;
		const int n_audio = NF * 10 / INTERLEAVE + 2, total = SPS * n_audio;
		uint8_t *frames = malloc(fsz * NF);
		for (int i = 0; i < NF; i++) orc_synth_frame(W, H, j->seed, (uint32_t)(j->index * NF + i), 4, frames + fsz * i);
		int16_t *pcm = calloc((size_t)(total + 4032) * 2, sizeof(int16_t));
		int16_t *tmp = malloc((size_t)total * sizeof(int16_t));
		for (int c = 0; c < 2; c++) {
			orc_synth_pcm(j->seed, (uint32_t)(2 * j->index + c), 0, total, 0, tmp);
			for (int i = 0; i < total; i++) pcm[2 * i + c] = tmp[i];
		}
		free(tmp);
		uint8_t frame_output[2016 * 10];
		orc_xa_settings_t os = {1, 1, 37800, 4, 1, 0};
		ref_xa_settings_t rs = {1, true, 37800, 4, 1, 0};
		pthread_barrier_wait(j->start);
		const double t0 = now();
		long n = 0;
		do {
			/* one little stream: filefmt.c:422-443 (state), then the sector loop until the frames are used up */
			orc_adpcm_state_t ost;
			ref_state_t rst;
			memset(&ost, 0, sizeof ost);
			memset(&rst, 0, sizeof rst);
			orc_str_state_t st;
			memset(&st, 0, sizeof st);
			st.base_overflow = 75 * 2 * (INTERLEAVE - 1) * 1;
			st.overflow_den = INTERLEAVE * 15;
			st.frame_output = frame_output;
			int frame = 0, audio = 0;
			for (int sector_count = 0; frame < NF - 2 || st.frame_data_offset < st.frame_max_size; sector_count++) {
				uint8_t sector[2352];
				memset(sector, 0, sizeof sector);
				if (sector_count % INTERLEAVE) {
					orc_cdrom_init_sector(sector, sector_count, 1);
					sector[16] = 1; sector[17] = 0; sector[18] = 0x08 | 0x40; sector[19] = 0;
					memcpy(sector + 20, sector + 16, 4);
					frame += orc_mdec_encode_sector_str(&st, 0, W, H, ORC_FMT_STRCD, 0x8001, frames + fsz * (frame < NF ? frame : NF - 1), sector);
					orc_cdrom_calculate_checksums(sector, 1);
				} else {
					const int16_t *src = pcm + (size_t)(audio < n_audio ? audio : n_audio - 1) * SPS * 2;
					const int len = j->ref_xa ? j->ref_xa(rs, &rst, src, SPS, sector_count, sector) : orc_xa_encode(os, &ost, src, SPS, sector_count, sector);
					if (len != 2352) j->failed = 1;
					audio++;
				}
				n++;
			}
		} while (now() - t0 < j->seconds);
		j->elapsed = now() - t0;
		j->units = n;
		free(frames);
		free(pcm);
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

/* cpu_bench converge <kind> <warmup units> <seed: 0 silence, 1 raw history> <starts> <cap units> <filters> <range>
 * How far a wrong start state travels (what bounds the GPU's speculate-and-verify): the serial encode of a
 * long chain of synthetic PCM gives the true state after every unit; from `starts` evenly spaced units an encoder is started
 * `warmup` units early -- from silence, or from the two RAW samples in front of it -- and the units from the start on are counted
 * until its state equals the true one (cap = never within that many).  Prints the distribution as JSON. */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
static int converge_main(int argc, char **argv) {
	if (argc < 9) return 2;
	const int kind = atoi(argv[2]), warm = atoi(argv[3]), seed_raw = atoi(argv[4]), starts = atoi(argv[5]), cap = atoi(argv[6]);
	const int filters = atoi(argv[7]), range = atoi(argv[8]);
	const int spacing = 2000, n_units = (starts + 2) * spacing + cap + warm + 8;
	int16_t *pcm = malloc((size_t)n_units * 28 * sizeof(int16_t));
	orc_synth_pcm(5, 3, 0, n_units * 28, kind, pcm);
	orc_adpcm_chan_t *truth = malloc((size_t)n_units * sizeof(orc_adpcm_chan_t));
	orc_adpcm_chan_t st = {0, 0};
	uint8_t codes[28];
	for (int u = 0; u < n_units; u++) {
		(void)orc_adpcm_encode_unit(&st, pcm + (size_t)u * 28, 28, 1, filters, range, codes);
		truth[u] = st;
	}
	int *dist = malloc((size_t)starts * sizeof(int));
	long sum = 0;
	int never = 0, at_start = 0;
	for (int k = 0; k < starts; k++) {
		const int first = (k + 1) * spacing;
		orc_adpcm_chan_t g = {0, 0};
		const int w0 = first - warm;
		if (seed_raw) { g.prev1 = pcm[(size_t)w0 * 28 - 1]; g.prev2 = pcm[(size_t)w0 * 28 - 2]; }
		for (int u = w0; u < first; u++) (void)orc_adpcm_encode_unit(&g, pcm + (size_t)u * 28, 28, 1, filters, range, codes);
		int d = 0;
		if (g.prev1 == truth[first - 1].prev1 && g.prev2 == truth[first - 1].prev2) at_start++;
		else {
			for (d = 1; d <= cap; d++) {
				(void)orc_adpcm_encode_unit(&g, pcm + (size_t)(first + d - 1) * 28, 28, 1, filters, range, codes);
				if (g.prev1 == truth[first + d - 1].prev1 && g.prev2 == truth[first + d - 1].prev2) break;
			}
			if (d > cap) never++;
		}
		dist[k] = d;
		sum += d > cap ? cap : d;
	}
	qsort(dist, (size_t)starts, sizeof(int), cmp_int);
	printf("{\"mode\": \"converge\", \"kind\": %d, \"warmup_units\": %d, \"seed\": \"%s\", \"starts\": %d, \"cap\": %d, \"guess_was_the_truth\": %d, "
	       "\"mean_units\": %.2f, \"median\": %d, \"p90\": %d, \"max\": %d, \"never_within_cap\": %d}\n",
	       kind, warm, seed_raw ? "raw history" : "silence", starts, cap, at_start, (double)sum / starts, dist[starts / 2], dist[starts * 9 / 10], dist[starts - 1], never);
	free(pcm); free(truth); free(dist);
	return 0;
}

/* cpu_bench spucall <ref.so or -> : microseconds per psx_audio_spu_encode call on ONE core at 28 .. 114 688 samples per call (state
 * carried): the reference's own function when the path of oracle/_ref's library is given, else orc_spu_encode.  The CPU side of the
 * per-call break-even; the GPU side is examples/percall_bench's sweep. */
typedef int (*ref_spu_encode_fn)(ref_chan_t *, const int16_t *, int, int, uint8_t *);
static int spucall_main(int argc, char **argv) {
	static const int sizes[] = {28, 56, 112, 224, 448, 896, 1792, 3584, 7168, 14336, 22064, 28672, 57344, 114688};
	const int ns = (int)(sizeof sizes / sizeof sizes[0]);
	ref_spu_encode_fn ref = NULL;
	if (argc > 2 && strcmp(argv[2], "-") != 0) {
		void *h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
		if (h) ref = (ref_spu_encode_fn)dlsym(h, "psx_audio_spu_encode");
	}
	int16_t *pcm = malloc(sizeof(int16_t) * (size_t)(114688 + 28));
	orc_synth_pcm(3, 0, 0, 114688 + 28, 0, pcm);
	uint8_t *out = malloc((size_t)(114688 / 28 + 2) * 16);
	printf("{\"mode\": \"spucall\", \"kind\": \"%s\", \"us_per_call_by_samples\": {", ref ? "reference" : "port");
	for (int k = 0; k < ns; k++) {
		const int n = sizes[k];
		ref_chan_t rst;
		orc_adpcm_chan_t ost = {0, 0};
		memset(&rst, 0, sizeof rst);
		long calls = 0;
		const double t0 = now();
		do {
			if (ref) (void)ref(&rst, pcm, n, 1, out);
			else (void)orc_spu_encode(&ost, pcm, n, 1, out);
			calls++;
		} while (now() - t0 < 0.25);
		printf("%s\"%d\": %.3f", k ? ", " : "", n, (now() - t0) * 1e6 / (double)calls);
	}
	printf("}}\n");
	free(pcm); free(out);
	return 0;
}

int main(int argc, char **argv) {
	if (argc >= 2 && strcmp(argv[1], "converge") == 0) return converge_main(argc, argv);
	if (argc >= 2 && strcmp(argv[1], "spucall") == 0) return spucall_main(argc, argv);
	if (argc < 5) {
		fprintf(stderr, "usage: cpu_bench mdec <threads> <seconds> <codec> <w> <h> <budget> <amp> <seed>\n"
		                "       cpu_bench xa <threads> <seconds> <seed> [path to oracle/_ref/libpsxav_ref.so]\n"
		                "       cpu_bench str <threads> <seconds> <seed> [path to oracle/_ref/libpsxav_ref.so]\n");
		return 2;
	}
	const int mode = strcmp(argv[1], "xa") == 0 ? 1 : (strcmp(argv[1], "str") == 0 ? 2 : 0);
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
			if (proto.ref_xa && mode == 1) kind = "reference";      /* (str: the video leg is the port whatever encodes the audio) */
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
	       mode == 1 ? "xa" : (mode == 2 ? "str" : "mdec"), kind, threads, total, longest, rate, lo, hi, failed);
	return failed ? 1 : 0;
}
