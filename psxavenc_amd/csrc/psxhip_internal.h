// psxhip_internal.h -- glue between the C-ABI layer (psxhip_api.cpp) and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/psxav_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	const uint8_t *d_frames;
	size_t frame_stride;
	int width, height, codec;
	int n_frames;
	const int32_t *d_max_sizes;
	int uniform_max_size;
	uint8_t *d_out;
	size_t out_stride;
	psxhip_mdec_result_t *d_results;
	int16_t *d_coef_slab;
	int out_words;
	int grid;
	int large;   /* 1: 16-wavefront groups, one per CU (large frames / budgets) */
	void *stream;
	unsigned long long *d_timing;
} psxhip_mdec_launch_t;

size_t psxhip_mdec_lds_bytes(int nmb, int out_words, int large);
size_t psxhip_mdec_slab_bytes_per_group(int nmb);
int psxhip_mdec_threads_per_group(int large);
hipError_t psxhip_mdec_upload_tables(void);
hipError_t psxhip_mdec_set_max_lds(int codec, size_t bytes);
hipError_t psxhip_mdec_launch(const psxhip_mdec_launch_t *a);

void psxhip_set_error(const char *fmt, ...);

#ifdef __cplusplus
}
#endif
