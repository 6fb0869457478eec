// psxhip_internal.h -- glue between the C-ABI layer (psxhip_api.cpp) and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/psxav_hip.h"

/* bumped with every change to the MDEC kernel: bench.py keys the committed PMC summaries on it (profiles/pmc_index.json) */
#define PSXHIP_MDEC_KERNEL_REV "mdec-k3.7"
/* ... and with every change to the ADPCM kernels (round 4's kernels count as adpcm-k4.0) */
#define PSXHIP_ADPCM_KERNEL_REV "adpcm-k5.3"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	const psxhip_mdec_batch_t *batches;   /* [n_batches], 1 .. PSXHIP_MDEC_MAX_BATCHES (host memory; copied into the kernel arguments) */
	int n_batches;
	size_t frame_stride;
	int width, height, codec;
	int n_frames;                         /* over all batches */
	int n_tickets, t4, t2;                /* frame tickets: runs of 4, runs of 2, single frames (psxhip_mdec_ticket_plan) */
	int uniform_max_size;
	size_t out_stride;
	int out_words;      /* LDS dwords of the frame image tile: out_tile + 2 */
	int out_tile;       /* image dwords assembled in LDS at a time: min((max_frame_size + 3) / 4, 2048) */
	int max_frame_size; /* the context's largest budget */
	int stg_words;      /* LDS dwords of the macroblock staging area: (max_frame_size + 3) / 4 + nmb + 2 */
	int grid;
	int large;          /* 1: 16-wavefront groups, one per CU (large frames / budgets) */
	void *stream;
	unsigned int *d_ticket;         /* [128] of the launch's lane: frame hand-out counters, retry queue words: zero between launches */
	unsigned int *d_hint;           /* one word shared by the context's lanes: the answer the next launch's groups start from */
	unsigned int *d_retry;          /* [retry_cap] retry queue slots, all 0xFFFFFFFF between launches; NULL = frames are never handed on */
	int retry_cap;
	int retry_patience;             /* looks a group without work waits for a handed-on frame before it leaves */
	unsigned long long *d_stats;    /* optional [PSXHIP_MDEC_STATS] diagnostics */
	unsigned prio_pattern;          /* see FrameJob */
	int trust_mode;                 /* 0 = the trust policy; experiments: 1 = foreign hints always trusted, 2 = never */
	int ck_margin;                  /* checkpoint margin in thousandths of the projection's standard error (0 = default) */
	const uint32_t *d_order;        /* psxhip_mdec_pass_table() for this geometry, in device memory */
} psxhip_mdec_launch_t;

/* one frame across many workgroups (mdec_split.inc): launches of a few frames -- the drop-in's one frame per call most of all */
typedef struct {
	int seg_mbs, segs;                  /* macroblocks per workgroup (1, 2, 4, 8, 16), workgroups per frame */
	int img_words;
	size_t ws_stride, ws_slots, ws_dcq, ws_img, ws_done;   /* workspace bytes per frame and the offsets of its parts */
	size_t lds_bytes;
} psxhip_mdec_split_geo_t;
typedef struct {
	const uint8_t *d_frames;
	uint8_t *d_out;                     /* written by plain stores only: may be page-locked host memory the device can see */
	psxhip_mdec_result_t *d_results;
	const int32_t *d_frame_max_sizes;
	size_t frame_stride, out_stride;
	int width, height, codec, n_frames, uniform_max_size, max_frame_size;
	psxhip_mdec_split_geo_t geo;
	void *d_ws;                         /* geo.ws_stride x n_frames bytes, all zero between launches (the kernel leaves it so) */
	unsigned int *d_lost;               /* frames given up by the rendezvous watchdog */
	unsigned long long *d_dbg;          /* diagnostics: [groups][8] phase stamps, or NULL */
	unsigned int *d_done_flag;          /* one-frame calls: a page-locked host word (device address) that receives done_seq when row and result are written, or NULL */
	unsigned int done_seq;
	int hint;                           /* the answer of the context's previous one-frame call (0: none) */
	void *stream;
} psxhip_mdec_split_t;
/* 1: the split kernel takes launches of n_frames frames of this geometry (g filled in), 0: it does not */
int psxhip_mdec_split_geometry(int codec, int width, int height, int max_frame_size, int n_frames, int n_cu, psxhip_mdec_split_geo_t *g);
hipError_t psxhip_mdec_split_launch(const psxhip_mdec_split_t *a);

size_t psxhip_mdec_lds_bytes(int nmb, int out_words, int stg_words, int large);
int psxhip_mdec_threads_per_group(int large);
hipError_t psxhip_mdec_upload_tables(void);
hipError_t psxhip_mdec_set_max_lds(int codec, size_t bytes);
int psxhip_mdec_pass_order(int width, int height, int large, uint32_t *out, int cap);
int psxhip_mdec_pass_table(int width, int height, int large, uint32_t *out /* [2 * (n + 1)] */, int cap);
void psxhip_mdec_ticket_plan(int n_frames, int groups, int max_run, int *t4, int *t2, int *n_tickets);
hipError_t psxhip_mdec_launch(const psxhip_mdec_launch_t *a);
hipError_t psxhip_mdec_stage_in_launch(const void *src_mapped, void *d_dst, size_t bytes, void *stream);
hipError_t psxhip_mdec_fdct_launch(const int16_t *d_in, int16_t *d_out, int n_blocks, void *stream);

int psxhip_mdec_encode_frames_host_rows(psxhip_mdec_ctx_t *ctx, const uint8_t *frames, int n_frames,
                                        const int32_t *frame_max_sizes, int uniform_max_size, uint8_t *out,
                                        size_t out_stride, psxhip_mdec_result_t *results, int row_bytes);
/* psxhip_xa_encode_streams_host with one EOF flag per sector (n_streams x sectors, or NULL = `finalize` semantics) */
int psxhip_xa_encode_streams_host_flags(int device, int format, int stereo, int frequency, int bits, int file_number,
                                        int channel_number, const int16_t *samples, int n_streams, int64_t stream_stride,
                                        int samples_per_stream, const int32_t *lbas, psxhip_adpcm_state_t *states,
                                        uint8_t *out, int64_t out_stride, int finalize, const uint8_t *eof_flags);

/* one launch for the reference's per-call pattern (adpcm_call_kernel): up to four chains, descriptors and start states in the
 * kernel arguments, samples read from device-visible (page-locked host) memory */
typedef struct {
	const int16_t *samples;
	int stage_elems;                     /* > 0: elements (multiple of 8, <= psxhip_adpcm_call_stage_max()) staged in LDS first */
	psxhip_adpcm_chain_t chains[4];
	psxhip_adpcm_state_t states_in[4];
	int32_t unit_base[4];
	int n_chains, filter_count, bits;
	psxhip_adpcm_state_t *states_out;    /* [n_chains] */
	uint8_t *units;                      /* unit records (PSXHIP_ADPCM_RECORD_SIZE(bits) apart), or NULL when spu_out is given */
	uint8_t *spu_out;                    /* packed 16-byte SPU blocks */
} psxhip_adpcm_call_t;
hipError_t psxhip_adpcm_call_launch(const psxhip_adpcm_call_t *a, void *stream);
int psxhip_adpcm_call_stage_max(void);
int psxhip_xa_assemble_device_bits(int device, const uint8_t *d_units, int n_sectors, int format, int stereo, int frequency, int bits,
                                   int file_number, int channel_number, int first_lba, const uint8_t *d_eof_flags, uint32_t eof_bits,
                                   uint8_t *d_out, void *stream);

int psxhip_xa_assemble_scatter(int device, const uint8_t *d_units, int n_sectors, int format, int stereo, int frequency, int bits,
                               int file_number, int channel_number, int first_lba, const uint8_t *d_eof_flags, uint32_t eof_bits,
                               uint8_t *d_out, const int32_t *d_dst_sector, int n_streams, size_t units_stream_stride,
                               size_t out_stream_stride, void *stream);
/* video sectors of muxed STR streams, built on the device (adpcm_kernels.hip: str_video_sector_kernel) */
typedef struct {
	const uint8_t *d_bs;                 /* the frames' bitstreams */
	size_t bs_stride, bs_stream_stride;
	const psxhip_mdec_result_t *d_res;   /* [n_streams][frames_per_stream] */
	int frames_per_stream;
	const int32_t *d_tab;                /* [n_entries][4]: slot in the stream, frame (-2: zero sector), byte offset into the bitstream, the frame's budget */
	int n_entries, n_streams;
	int format, sector_size;
	int xa_file, xa_channel, video_id, width, height;
	uint8_t *d_out;
	size_t out_stream_stride;
} psxhip_str_video_job_t;
int psxhip_str_video_sectors_launch(int device, const psxhip_str_video_job_t *a, void *stream);
void psxhip_adpcm_pick_chunking(long long total_units, int rows, int device, int *chunk_units, int *warmup_units);
/* units per chain from which a chain is cut along time (speculate-and-verify) instead of run serially; the host entry points and
 * psxhip_str_encode_device ask the same function */
int psxhip_adpcm_chunked_threshold(int n_chains);

void psxhip_set_error(const char *fmt, ...);

#ifdef __cplusplus
}
#endif
