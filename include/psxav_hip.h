/*
 * psxav_hip.h -- batched, device-resident extensions of the psxavenc hot path for MI355X (gfx950).
 *
 * The reference API encodes one frame / one 28-sample block per synchronous call
 * (psxavenc/mdec.h:65-74, libpsxav/libpsxav.h:73-101).  A kernel launch + copy per call costs
 * more than the work itself, so the throughput surface is batched: N frames (or N independent
 * ADPCM chains) per call, buffers already resident in HBM, asynchronous on a caller stream.
 * The per-call drop-in functions in psxav_mdec.h / psxav_audio.h are thin wrappers over these.
 *
 * Plain C ABI: pointers, sizes, ints.  Functions return 0 on success or a negative PSXHIP_E*
 * code; psxhip_last_error() gives the text for the calling thread.  "d_" parameters are device
 * pointers (hipMalloc / torch CUDA tensors), everything else is host memory.  `stream` is a
 * hipStream_t passed as void* (NULL = the null stream).
 */
#ifndef PSXAV_HIP_H
#define PSXAV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
	PSXHIP_OK = 0,
	PSXHIP_EINVAL = -1,     /* bad argument (size not a multiple of 16, NULL pointer, ...) */
	PSXHIP_EDEVICE = -2,    /* HIP runtime error / no gfx950 device */
	PSXHIP_ENOMEM = -3,
	PSXHIP_ENOFIT = -4      /* some frame does not fit its budget at any quant scale < 64 (the
	                           reference asserts here, psxavenc/mdec.c:723) */
};

int psxhip_device_count(void);
const char *psxhip_last_error(void);
const char *psxhip_version(void);

/* ---------------------------------------------------------------- MDEC BS frame encoder ---- */

/* what encode_frame_bs leaves in mdec_encoder_state_t (psxavenc/mdec.c:719-736) */
typedef struct {
	int32_t quant_scale;         /* 1..63; 64 = no scale fits (bytes_used = 0 then) */
	int32_t bytes_used;          /* bitstream bytes incl. the 8-byte header, rounded up to 4 */
	int32_t blocks_used;         /* MDEC command word count */
	int32_t uncomp_hwords_used;  /* rounded up to 64 */
} psxhip_mdec_result_t;

typedef struct psxhip_mdec_ctx psxhip_mdec_ctx_t;

/* codec: 0 = BS v2, 1 = v3, 2 = v3dc (bs_codec_t, psxavenc/args.h:61-65).
 * width/height: multiples of 16 (psxavenc/mdec.c:601-602), at most 1024 each.
 * max_frame_size: largest per-frame byte budget that will be passed (sizes the LDS staging).
 * A frame's working set (budget + min(budget, 8 KiB) + ~30 bytes per macroblock + ~37 KiB) must fit the CU's 160 KiB
 * LDS, else PSXHIP_EINVAL: e.g. 640x512 (the reference CLI's maximum, args.c:410-421) works up to 84 668-byte budgets;
 * psxhip_mdec_query_geometry() tells before creating a context. */
int psxhip_mdec_create(psxhip_mdec_ctx_t **ctx, int device, int codec, int width, int height,
                       int max_frame_size);
void psxhip_mdec_destroy(psxhip_mdec_ctx_t *ctx);

/* Encode n_frames NV21 frames (frame i at d_frames + i*frame_stride, w*h*3/2 bytes each) into
 * d_out + i*out_stride.  Exactly frame_max_size bytes are written per frame: header, bitstream,
 * zero fill -- what psxavenc/mdec.c:676,739-754 leave in frame_output.  d_frame_max_sizes may be
 * NULL, then every frame uses uniform_max_size.  d_frames, d_out, frame_stride and out_stride
 * must be 4-byte aligned.  Asynchronous on `stream`; results land in d_results[i].
 * Launches on ONE context must be stream-ordered (same stream, or ordered by events): the context owns the
 * frame hand-out counters the kernel uses.  Use one context per concurrent stream (or two launch lanes, psxhip_mdec_set_lanes).
 * Per-frame budgets (device memory, not vetted by the host) outside [8, min(the context's max_frame_size,
 * out_stride)] make that frame's result quant_scale 64 and write nothing.
 * Launches of at most 12 frames (PSXHIP_MDEC_SPLIT_MAX) cut every frame across many workgroups (csrc/mdec_split.inc: the
 * reference's call pattern is ONE frame per call, psxavenc/filefmt.c:641-647) -- same bytes, same results, same ordering rules.  Such
 * launches of one process are ordered one behind the other per device; a frame whose workgroups could not all become resident
 * within 0.2 s (another PROCESS holding the device's compute units) comes back with quant_scale 64 and is counted by
 * psxhip_mdec_watchdog -- the host-buffer one-frame call takes such a frame again through the one-workgroup kernel by itself. */
int psxhip_mdec_encode_frames_device(psxhip_mdec_ctx_t *ctx, const uint8_t *d_frames, size_t frame_stride,
                                     int n_frames, const int32_t *d_frame_max_sizes, int uniform_max_size,
                                     uint8_t *d_out, size_t out_stride, psxhip_mdec_result_t *d_results,
                                     void *stream);

/* Several batches, ONE launch.  The reference's caller is one in-order loop over frames (psxavenc/filefmt.c:641-647); a caller
 * that holds a few batches -- four 1000-frame chunks of a file, say -- hands them over together and gets the behaviour of one
 * large batch (frames of all batches are drawn from one ticket counter: no launch boundary between them, the tail of one batch
 * is filled by the head of the next) without concatenating its buffers.  Bytes and results are those of one
 * psxhip_mdec_encode_frames_device call per batch.  frame_stride, out_stride and uniform_max_size are common to the batches; a
 * batch's d_frame_max_sizes may be NULL (uniform_max_size applies).  More than PSXHIP_MDEC_MAX_BATCHES batches are issued as
 * several launches.  Batches with n_frames == 0 are skipped. */
#define PSXHIP_MDEC_MAX_BATCHES 8
typedef struct {
	const uint8_t *d_frames;              /* n_frames NV21 frames, frame_stride apart */
	int32_t n_frames;
	int32_t reserved;
	const int32_t *d_frame_max_sizes;     /* [n_frames] or NULL */
	uint8_t *d_out;                       /* n_frames rows, out_stride apart */
	psxhip_mdec_result_t *d_results;      /* [n_frames] */
} psxhip_mdec_batch_t;
int psxhip_mdec_encode_batches_device(psxhip_mdec_ctx_t *ctx, const psxhip_mdec_batch_t *batches, int n_batches,
                                      size_t frame_stride, int uniform_max_size, size_t out_stride, void *stream);

/* Launch lanes.  With one lane (the default) a launch is an ordinary stream operation: it starts when everything before it on
 * `stream` is done, and everything after it on `stream` sees its results -- so consecutive launches of an in-order caller
 * (filefmt.c:641-647 is one) run strictly one after the other, and the GPU idles through every launch's tail (8 % at 1000
 * frames of 320x240: a launch ends when its slowest CU does).  With two lanes the context owns two sets of frame hand-out
 * counters and two internal streams, and psxhip_mdec_encode_frames_device / _batches_device on a caller stream S become:
 *   - INPUTS are stream-ordered: launch k starts when everything enqueued on S before call k is done;
 *   - RESULTS lag one call: when call k returns, S is ordered behind launches 0 .. k-1; launch k itself is ordered into S by the
 *     next encode call on the context or by psxhip_mdec_fence(ctx, S).  A caller that reads launch k's output (or reuses its
 *     input or output buffers) must have called one of the two first -- i.e. it double-buffers, which is what lets the head of
 *     launch k+1 fill the tail of launch k.
 * Bytes never depend on the number of lanes.  lanes: 1 or 2.  Switching waits for the context's outstanding launches, and so do
 * the host-buffer entry points of the same context (psxhip_mdec_encode_frames_host*, encode_frame_bs): they use both lanes, on the
 * context's own streams.  With ONE lane a device-path launch runs on the caller's stream with lane 0's hand-out counters and nothing
 * orders a later host-buffer call (encode_frame_bs included) behind it: a caller that mixes the two on one context synchronises
 * its stream first -- the rule above (launches on one context are stream-ordered) applies to the host entry points too. */
int psxhip_mdec_set_lanes(psxhip_mdec_ctx_t *ctx, int lanes);
/* order `stream` behind every launch of the context issued so far (a no-op with one lane) */
int psxhip_mdec_fence(psxhip_mdec_ctx_t *ctx, void *stream);
/* Waits for the context's launches and returns, in *lost, how often the retry queue's watchdog gave a frame up (a workgroup that
 * reserved a queue slot never filled it within about a second: a faulted or preempted workgroup).  0 on a healthy device --
 * anything else means some launch's results are incomplete.  Never reset. */
int psxhip_mdec_watchdog(psxhip_mdec_ctx_t *ctx, unsigned *lost);

/* Same, host buffers: H2D, kernel, D2H, synchronise.  frame_max_sizes may be NULL (uniform).
 * Returns PSXHIP_ENOFIT if any frame could not be fitted (its result has quant_scale 64).
 * The batch moves in chunks over two streams (the copies of one chunk overlap the kernel of the next).  Pageable `frames`
 * / `out` go through pinned staging buffers (a multi-threaded CPU copy per chunk); buffers that are page-locked --
 * hipHostMalloc, hipHostRegister or psxhip_host_register() below -- are read and written by DMA directly. */
int psxhip_mdec_encode_frames_host(psxhip_mdec_ctx_t *ctx, const uint8_t *frames, int n_frames,
                                   const int32_t *frame_max_sizes, int uniform_max_size, uint8_t *out,
                                   size_t out_stride, psxhip_mdec_result_t *results);

/* The 8x8 forward DCT alone, exactly as the frame kernel computes it: blocks = n_blocks * 64 level-shifted samples
 * (-128..127, raster order; what psxavenc/mdec.c:619-633 hands to AVDCT.fdct), coefs = the 64 coefficients per block
 * in raster order (the in-place result of mdec.c:640).  The FDCT is the one piece of the path the reference takes from
 * FFmpeg; tools/check_fdct_vs_ffmpeg.c uses this entry point to diff the device arithmetic against a real libavcodec. */
int psxhip_mdec_fdct_host(int device, const int16_t *blocks, int n_blocks, int16_t *coefs);

/* Name and grid of the kernel the last encode call launched (for bench.py's roofline block). */
const char *psxhip_mdec_kernel_name(void);

/* What a geometry costs, before creating a context for it.  A frame's working set lives in the CU's
 * 160 KiB LDS: the macroblock staging area (max_frame_size) + the frame image (whole, or one 8 KiB tile at a time when
 * that is what fits) + ~30 bytes per macroblock + ~37 KiB (two workgroups per CU when twice that fits, else one).  fits == 0 means psxhip_mdec_create would return PSXHIP_EINVAL; max_frame_size_limit
 * is the largest budget this frame size supports (320x240: 109 644 bytes, 640x480: 86 700, 640x512: 84 668). */
typedef struct {
	int32_t fits;
	int32_t groups_per_cu;          /* frames in flight per compute unit (2 or 1) */
	int32_t wavefronts_per_group;
	int32_t frames_in_flight;       /* persistent grid size = compute units * groups_per_cu */
	int32_t max_frame_size_limit;
	int32_t image_tile_bytes;       /* bytes of the frame image assembled in LDS at a time: the whole budget, or 8 / 4 / 2 KiB */
	int64_t lds_bytes_per_group;
	int64_t lds_bytes_per_cu;
} psxhip_mdec_geometry_t;
int psxhip_mdec_query_geometry(int device, int codec, int width, int height, int max_frame_size,
                               psxhip_mdec_geometry_t *out);

/* Diagnostics: when the context was created with PSXHIP_MDEC_STATS=1 in the environment, the kernel counts
 * [0] frames encoded, [1] passes over frames (1 per frame when the pilot's prediction held), [2..7] histogram of
 * passes per frame (0, 1, 2, 3, 4, >= 5).  Copies min(n, PSXHIP_MDEC_STATS_TOTAL) entries; zeros otherwise. */
#define PSXHIP_MDEC_STATS 8
/* after those, 4 entries per workgroup (the first PSXHIP_MDEC_TRACE_GROUPS groups of the last launch): start and end
 * time (100 MHz wall clock), frames encoded, reserved */
#define PSXHIP_MDEC_TRACE_GROUPS 1024
/* and then 16 time sums over all groups (100 MHz ticks).  Of each group's first thread, per phase: 0 ticket/idle, 1 reset +
 * DC pre-pass, 2 pilot, 3 passes over the frame, 4 offset scan + merge, 5 header + write-out.  Over all wavefronts: 6 time
 * spent waiting at group barriers, 7 residency; 8..13 the barrier time by barrier (frame start, DC pre-pass + pilot,
 * checkpoint, end of pass, search step, merge + write-out).  Word 2 of a group's trace record: frames | ticks from group
 * entry to the end of its prologue << 8 | ticks to the end of its first frame << 32. */
#define PSXHIP_MDEC_STATS_PHASE0 (PSXHIP_MDEC_STATS + 4 * PSXHIP_MDEC_TRACE_GROUPS)
/* and one word per frame of the last launch (the first PSXHIP_MDEC_TRACE_FRAMES frames): first guess | first checkpoint
 * verdict << 8 | answer << 16 | passes << 24 */
#define PSXHIP_MDEC_TRACE_FRAMES 2048
#define PSXHIP_MDEC_STATS_FRAME0 (PSXHIP_MDEC_STATS_PHASE0 + 16)
#define PSXHIP_MDEC_STATS_TOTAL (PSXHIP_MDEC_STATS_FRAME0 + PSXHIP_MDEC_TRACE_FRAMES)
int psxhip_mdec_read_stats(psxhip_mdec_ctx_t *ctx, unsigned long long *out, int n, int reset);

/* ---------------------------------------------------------------- SPU / XA ADPCM ----------- */

/* carried state of one channel: the last two DECODED samples (libpsxav/adpcm.c:135-136).  The
 * reference's qerr is never updated and mse is per-trial scratch (adpcm.c:107,131-132). */
typedef struct {
	int32_t prev1, prev2;
} psxhip_adpcm_state_t;

/* One independent encoder chain: `n_units` consecutive 28-sample sound units read from
 * d_samples + sample_offset with stride `pitch` (int16 elements); chain-local samples at index
 * >= sample_limit read as zero without touching memory (what the reference gets from its zero-padded
 * input, adpcm.c:65,110 and decoding.c:497-503).  Unit u of the chain is written to record
 * unit_base[c] + u * unit_stride, so the L/R chains of a stereo XA stream interleave into encode order. */
typedef struct {
	int64_t sample_offset;   /* element offset into d_samples */
	int32_t pitch;           /* 1 = mono / planar, 2 = interleaved stereo */
	int32_t sample_limit;    /* valid samples of this chain from sample_offset on */
	int32_t n_units;         /* sound units to encode */
	int32_t unit_stride;     /* record stride, normally 1 (2 for the halves of a stereo pair) */
} psxhip_adpcm_chain_t;

#define PSXHIP_ADPCM_RECORD_BYTES 32   /* 8-bit codes: byte 0: (shift & 15) | filter << 4; bytes 4..31: the 28 codes.  Also the upper bound of a record's size */
#define PSXHIP_ADPCM_RECORD_BYTES_4BIT 16   /* 4-bit codes (SPU, 4-bit XA): the layout of an SPU block (adpcm.c:367-372): header, 0, 14 bytes of two codes each (even sample low) */
#define PSXHIP_ADPCM_RECORD_SIZE(bits) ((bits) == 4 ? PSXHIP_ADPCM_RECORD_BYTES_4BIT : PSXHIP_ADPCM_RECORD_BYTES)   /* bytes between the records of consecutive unit indices */

/* Encode n_chains independent chains.  filter_count 5 (SPU) or 4 (XA); bits 4 or 8 (shift range
 * 12 / 8, adpcm.c:29-34).  Output: one record per unit (PSXHIP_ADPCM_RECORD_SIZE(bits) bytes apart, see above) in d_units; d_states[c]
 * is read and updated.  Result per unit == libpsxav/adpcm.c:142-191 encode(). */
int psxhip_adpcm_encode_chains_device(int device, const int16_t *d_samples, const psxhip_adpcm_chain_t *d_chains,
                                      const int32_t *d_unit_base, int n_chains, int filter_count, int bits,
                                      psxhip_adpcm_state_t *d_states, uint8_t *d_units, void *stream);

/* Same result as psxhip_adpcm_encode_chains_device, but parallel ALONG each chain ("speculate and verify"):
 * chains are cut into chunks of `chunk_units` sound units, every chunk is encoded concurrently from a guessed
 * start state (obtained by running `warmup_units` units before the chunk from a zero state), then verify
 * passes re-encode any chunk whose guess differs from its predecessor's actual end state, until a pass changes
 * nothing.  The fixpoint equals the serial encode bit for bit; the guesses only affect speed.  `chains` and
 * `unit_base` are HOST arrays here (the chunk tables are built on the host); the call synchronises.
 * max_passes <= 0: no limit (the worst case is one chunk per pass).  Returns the number of verify passes
 * (>= 1) or a negative error. */
int psxhip_adpcm_encode_chains_chunked(int device, const int16_t *d_samples, const psxhip_adpcm_chain_t *chains,
                                       const int32_t *unit_base, int n_chains, int filter_count, int bits,
                                       psxhip_adpcm_state_t *d_states, uint8_t *d_units, int chunk_units,
                                       int warmup_units, int max_passes, void *stream);

/* The same machinery as a persistent session, for sharding chains ALONG TIME across GPUs: a rank that owns units
 * [a, b) of a chain does not know the state at `a` until its predecessor has finished.  lead_units[c] > 0 makes the
 * session guess chain c's start state from up to `warmup_units` units located BEFORE the chain's sample_offset
 * (they must be readable); psxhip_adpcm_session_run() (re)verifies everything against the start states passed in --
 * call it again with corrected states after exchanging final states with the neighbour rank; it returns the
 * number of verify passes and sets *any_change when any record was (re)written.  The fixpoint over all ranks
 * (no rank changed) equals the serial encode.  See psxavenc_amd/parallel.py: encode_chains_time_sharded(). */
typedef struct psxhip_adpcm_session psxhip_adpcm_session_t;
int psxhip_adpcm_session_create(psxhip_adpcm_session_t **session, int device, const int16_t *d_samples,
                                const psxhip_adpcm_chain_t *chains, const int32_t *unit_base,
                                const int32_t *lead_units, int n_chains, int filter_count, int bits, uint8_t *d_units,
                                int chunk_units, int warmup_units, void *stream);
/* start_known (optional, [n_chains]): 0 = start_states[c] is not known yet -> keep the warm-up guess for chain c. */
int psxhip_adpcm_session_run(psxhip_adpcm_session_t *session, const psxhip_adpcm_state_t *start_states,
                             const uint8_t *start_known, int max_passes, psxhip_adpcm_state_t *final_states,
                             int *any_change);
/* Measurement: HIP events around the speculate launch and around the verify passes of every run that speculates (the first run
 * after create / reset); _last_timing returns the last such run's two durations in milliseconds.  Off by default. */
int psxhip_adpcm_session_set_timing(psxhip_adpcm_session_t *session, int on);
int psxhip_adpcm_session_last_timing(const psxhip_adpcm_session_t *session, float *speculate_ms, float *verify_ms);
/* revision of the ADPCM kernels (profiles/pmc_index.json is keyed by it, like the frame kernel's in psxhip_version()) */
const char *psxhip_adpcm_kernel_rev(void);
/* forget the speculative encode: the next psxhip_adpcm_session_run starts over (the samples may have changed) */
void psxhip_adpcm_session_reset(psxhip_adpcm_session_t *session);
void psxhip_adpcm_session_destroy(psxhip_adpcm_session_t *session);

/* Pack n_blocks unit records into 16-byte SPU blocks (adpcm.c:367-372).  d_out 16-byte aligned. */
int psxhip_spu_pack_device(int device, const uint8_t *d_units, int n_blocks, uint8_t *d_out, void *stream);

/* Assemble XA sectors from unit records in encode order (adpcm.c:193-233,266-332): sector s takes
 * records [s * 18 * U, (s+1) * 18 * U), U = 8 (4-bit) or 4 (8-bit) units per sound group.  Writes
 * 2336 (.xa, format 0) or 2352 (XACD, format 1) bytes per sector incl. sync, BCD time code,
 * subheaders and the form-2 EDC (libpsxav/cdrom.c:28-41,55-74,102-110); bytes the reference leaves
 * unwritten are zero.  d_eof_flags (optional): non-zero entries set the EOF submode bit
 * (psx_audio_xa_encode_finalize, adpcm.c:334-340). */
int psxhip_xa_assemble_device(int device, const uint8_t *d_units, int n_sectors, int format, int stereo,
                              int frequency, int bits, int file_number, int channel_number, int first_lba,
                              const uint8_t *d_eof_flags, uint8_t *d_out, void *stream);

/* Host-buffer batches.  n_streams independent streams, stream i at samples + i*stream_stride
 * (elements), samples_per_stream samples each read with `pitch`; states[i] is carried in and out.
 * SPU: stream i's 16 * ceil(n/28) bytes go to out + i*out_stride.  Returns bytes per stream or < 0. */
int psxhip_spu_encode_streams_host(int device, const int16_t *samples, int n_streams, int64_t stream_stride,
                                   int pitch, int samples_per_stream, psxhip_adpcm_state_t *states,
                                   uint8_t *out, int64_t out_stride);
/* XA: stereo streams are interleaved L,R (samples_per_stream counts per channel); states[2*i] /
 * states[2*i+1] are the left / right channel states; lbas[i] is the first sector's LBA.  finalize != 0
 * sets EOF on each stream's last sector.  Returns bytes per stream (whole sectors) or < 0. */
int psxhip_xa_encode_streams_host(int device, int format, int stereo, int frequency, int bits, int file_number,
                                  int channel_number, const int16_t *samples, int n_streams, int64_t stream_stride,
                                  int samples_per_stream, const int32_t *lbas, psxhip_adpcm_state_t *states,
                                  uint8_t *out, int64_t out_stride, int finalize);

/* The host-buffer ADPCM entry points keep their device scratch buffers per calling thread between calls (the reference
 * calls them once per 28 samples / once per sector); this releases the calling thread's. */
void psxhip_release_scratch(void);

/* Page-lock a caller-owned host buffer (hipHostRegister) so that the *_host entry points move it by DMA without a
 * staging copy; worth it for buffers that live across many calls (registration costs about as much as copying the
 * buffer once).  Unregister before freeing the memory. */
int psxhip_host_register(void *p, size_t bytes);
int psxhip_host_unregister(void *p);

/* ---------------------------------------------------------------- several devices behind one call ---- */

/* The reference's host side is one C loop (psxavenc/filefmt.c:633-662 over frames, :450-503 over sectors): a C caller has
 * no ranks to shard over, so these entry points take a device LIST and shard inside the call -- one host thread, one
 * encoder context and one pair of pinned staging buffers per list entry.  Frames (mdec.c:678-686: every attempt resets all
 * bit / DC state) and XA streams (adpcm.c:202-209: a state per channel) are independent units, so the bytes are those of
 * the single-device call whatever the schedule.  A device may be listed more than once.  The host-buffer path is bound by
 * the PCIe link of each device (about 440 k 320x240 frames/s), which is what several devices multiply. */

/* contiguous block partition of n_units over `world` workers: the first n_units % world workers get one unit more
 * (the same partition psxavenc_amd/parallel.py:shard_range uses across ranks) */
void psxhip_shard_range(int64_t n_units, int rank, int world, int64_t *first, int64_t *count);

/* host-side ticket queue: [0, n_units) handed out in ranges of ticket_units, in order, each exactly once, to any number
 * of threads (one atomic counter in host memory; no device collective).  _next returns 0 when the queue is empty. */
typedef struct psxhip_ticket_queue psxhip_ticket_queue_t;
psxhip_ticket_queue_t *psxhip_ticket_queue_create(int64_t n_units, int64_t ticket_units);
int psxhip_ticket_queue_next(psxhip_ticket_queue_t *q, int64_t *first, int64_t *count);
void psxhip_ticket_queue_destroy(psxhip_ticket_queue_t *q);

enum {
	PSXHIP_SCHED_STATIC = 0,    /* worker d takes psxhip_shard_range(n, d, n_devices) */
	PSXHIP_SCHED_TICKETS = 1    /* workers draw ranges of ticket_frames frames from a ticket queue until it is empty: a device
	                               that drew expensive frames (content at a scale boundary costs up to 1.5x) draws fewer */
};

/* what each worker did (optional out-parameter, one entry per listed device, at most PSXHIP_MULTI_MAX_REPORT) */
#define PSXHIP_MULTI_MAX_REPORT 64
typedef struct {
	int32_t device;
	int64_t units;       /* frames / streams this worker encoded */
	int32_t tickets;     /* ranges it drew */
	double seconds;      /* wall time from the start of the call to this worker's end */
} psxhip_multi_report_t;

typedef struct psxhip_mdec_multi psxhip_mdec_multi_t;
int psxhip_mdec_multi_create(psxhip_mdec_multi_t **m, const int *devices, int n_devices, int codec, int width, int height,
                             int max_frame_size);
void psxhip_mdec_multi_destroy(psxhip_mdec_multi_t *m);
int psxhip_mdec_multi_device_count(const psxhip_mdec_multi_t *m);
/* psxhip_mdec_encode_frames_host over all listed devices; same arguments, same bytes, same results, same return value.
 * ticket_frames <= 0: a default (1536 frames, halved until every device gets about 8 tickets). */
int psxhip_mdec_multi_encode_frames_host(psxhip_mdec_multi_t *m, const uint8_t *frames, int n_frames,
                                         const int32_t *frame_max_sizes, int uniform_max_size, uint8_t *out,
                                         size_t out_stride, psxhip_mdec_result_t *results, int schedule, int ticket_frames,
                                         psxhip_multi_report_t *report);

/* psxhip_xa_encode_streams_host with the streams sharded over the listed devices (contiguous stream ranges, e.g. the 8
 * XA channels of config `xacd` on 8 GPUs); same bytes and states. */
int psxhip_xa_encode_streams_host_multi(const int *devices, int n_devices, int format, int stereo, int frequency, int bits,
                                        int file_number, int channel_number, const int16_t *samples, int n_streams,
                                        int64_t stream_stride, int samples_per_stream, const int32_t *lbas,
                                        psxhip_adpcm_state_t *states, uint8_t *out, int64_t out_stride, int finalize,
                                        psxhip_multi_report_t *report);

/* ---------------------------------------------------------------- STR / STRCD / STRV muxer -- */

/* The reference's encode_file_str (psxavenc/filefmt.c:391-520) for inputs that are all there up front: every frame goes
 * through ONE batched MDEC call (the per-frame budgets are a closed-form function of the frame index, mdec.c:768-775),
 * sharded over the handle's devices; the audio is one XA stream encoded concurrently by the ADPCM kernels, and the host
 * interleaves 2016-byte slices of the finished frames with the finished audio sectors on the reference's sector schedule
 * ((sector % interleave) > 0 = video, filefmt.c:454-461).  Fields mirror args_t (psxavenc/args.h). */
enum {
	/* How the stream ends.  REFERENCE (default, 0) = the CLI's sector loop fed by its decoder (decoding.c:510-560) when the
	 * whole input is there: end_of_input is raised as soon as no more than `frames_needed` (>= 2, filefmt.c:443-446) frames or
	 * no more than one sector's worth of audio are left; the loop then runs until the frame in progress is written out
	 * (filefmt.c:450) -- the last frames_needed frames of the input are NOT encoded (the FIXME at filefmt.c:442), a stream
	 * whose audio is shorter than its video ends with the audio -- every audio sector from that point on carries EOF
	 * (:492-493), and an audio slot with no samples left is an all-zero sector that also widens the video share of the
	 * trailing-audio schedule (:483-484).
	 * COMPLETE = every frame is encoded, the stream ends with the last frame's last sector, short audio is padded with
	 * silence and only the last audio sector carries EOF (what a caller who wants all of its frames in the file asks for). */
	PSXHIP_STR_TAIL_REFERENCE = 0,
	PSXHIP_STR_TAIL_COMPLETE = 1
};

typedef struct {
	int32_t format;             /* format_t: 6 = STR (2336-byte sectors), 7 = STRCD (2352), 9 = STRV */
	int32_t video_codec;        /* bs_codec_t */
	int32_t video_width, video_height;
	int32_t str_fps_num, str_fps_den;
	int32_t str_cd_speed;       /* 1 or 2 */
	int32_t str_video_id;       /* 0x8001 */
	int32_t trailing_audio;     /* FLAG_STR_TRAILING_AUDIO: audio sector last in each block instead of first */
	int32_t audio_channels;     /* 0 = no audio stream (all sectors are video), 1, 2 */
	int32_t audio_frequency;    /* 18900 / 37800 */
	int32_t audio_bit_depth;    /* 4 / 8 */
	int32_t audio_xa_file, audio_xa_channel;
	int32_t tail_mode;          /* PSXHIP_STR_TAIL_* */
	int32_t reserved;
} psxhip_str_settings_t;

typedef struct {
	int32_t n_sectors;
	int32_t n_video_sectors, n_audio_sectors;   /* audio slots, incl. those with no samples left */
	int32_t sector_size;        /* 2336 or 2352 */
	int32_t interleave;         /* sectors per block: 1 audio + (interleave - 1) video */
	int32_t audio_samples_per_sector;   /* per channel */
	int32_t max_frame_size;     /* largest per-frame budget */
	int32_t n_frames_encoded;   /* frames that are part of the stream (REFERENCE tail: n_frames - frames_needed, or fewer when
	                               the audio ends first) */
	int64_t quant_scale_sum;    /* filled by psxhip_str_encode_host (mdec_encoder_state_t.quant_scale_sum) */
} psxhip_str_plan_t;

/* sector counts and sizes for n_frames frames and pcm_samples_per_channel samples of audio (what a caller needs to size the
 * output; the amount of audio matters because the reference's stream ends with whichever input ends first) */
int psxhip_str_plan(const psxhip_str_settings_t *settings, int n_frames, int64_t pcm_samples_per_channel,
                    psxhip_str_plan_t *plan);
/* what each sector of the stream holds, in stream order: returns the sector count (= plan.n_sectors) and fills
 * min(cap, count) entries */
enum { PSXHIP_STR_SECTOR_VIDEO = 0, PSXHIP_STR_SECTOR_AUDIO = 1, PSXHIP_STR_SECTOR_EMPTY = 2 /* audio slot, no samples left */ };
typedef struct {
	int32_t kind;
	int32_t frame;      /* video: 0-based frame; else -1 */
	int32_t index;      /* video: chunk index inside the frame (mdec.c:794); audio: index of the XA sector; empty: -1 */
	int32_t eof;        /* audio: the sector carries the EOF submode bit */
} psxhip_str_sector_t;
int psxhip_str_plan_sectors(const psxhip_str_settings_t *settings, int n_frames, int64_t pcm_samples_per_channel,
                            psxhip_str_sector_t *sectors, int cap);
/* frame_max_size of frames first_frame .. first_frame + n_frames - 1 (so that any rank can budget its own frame range) */
int psxhip_str_frame_budgets(const psxhip_str_settings_t *settings, int first_frame, int n_frames, int32_t *budgets);

/* A muxer handle owns what is kept between calls: the encoder contexts (device buffers, pinned staging) of the listed
 * devices and the page-locked buffer the bitstreams land in.  Handles are independent (no process-global state); calls
 * on ONE handle are serialised. */
typedef struct psxhip_str_ctx psxhip_str_ctx_t;
int psxhip_str_create(psxhip_str_ctx_t **ctx, const int *devices, int n_devices);
void psxhip_str_destroy(psxhip_str_ctx_t *ctx);
/* frames: n_frames NV21 frames back to back (w*h*3/2 bytes each); pcm: int16, interleaved L,R when stereo,
 * pcm_samples_per_channel of them.  out: plan.n_sectors * sector_size bytes.  Sector bytes the reference leaves unwritten
 * (it muxes into an uninitialised stack buffer) are zero. */
int psxhip_str_encode_host(psxhip_str_ctx_t *ctx, const psxhip_str_settings_t *settings, const uint8_t *frames, int n_frames,
                           const int16_t *pcm, int64_t pcm_samples_per_channel, uint8_t *out, size_t out_size,
                           psxhip_str_plan_t *plan);

/* The same, device-resident: frames and PCM in HBM, muxed sectors in HBM -- no PCIe, no host interleave -- for n_streams independent
 * streams of the same settings and lengths (stream i: frames at d_frames + i * frames_stream_stride bytes, PCM at d_pcm + i *
 * pcm_stream_stride int16 elements, sectors at d_out + i * out_stream_stride bytes; strides and pointers 4-byte aligned).  The frames
 * of all streams are one batched MDEC launch, the XA tracks chains of one speculate-and-verify session (S x 2 chains share the
 * verify passes' latency, which is what bounds ONE stream), the video sectors are built by a kernel (sector header, subheaders,
 * chunk header mdec.c:782-820, 2016-byte slice :832, form-1 EDC cdrom.c:92-100) and the audio sectors assembled into their slots
 * (filefmt.c:454-461).  Runs on devices[0] of the handle; `stream` orders the inputs; the call returns when the sectors are complete
 * (the host drives the verify passes).  Bytes = psxhip_str_encode_host per stream; plan->quant_scale_sum is summed over all streams.
 * Tables and buffers are kept in the handle for the next call of the same shape. */
int psxhip_str_encode_device(psxhip_str_ctx_t *ctx, const psxhip_str_settings_t *settings, int n_streams, const uint8_t *d_frames,
                             size_t frames_stream_stride, int n_frames, const int16_t *d_pcm, int64_t pcm_stream_stride,
                             int64_t pcm_samples_per_channel, uint8_t *d_out, size_t out_stream_stride, psxhip_str_plan_t *plan,
                             void *stream);

/* ---------------------------------------------------------------- SPU / VAG / SPUI / VAGI files ---- */

/* The reference's encode_file_spu / encode_file_spui (psxavenc/filefmt.c:212-389, .vag header :95-162) for PCM that is
 * all there up front: all channels' blocks come from one batched GPU call, the host places them -- leading dummy
 * block, loop flags, trailing trap block, alignment padding, big-endian .vag header.  Fields mirror args_t. */
typedef struct {
	int32_t format;             /* format_t: 2 = SPU, 3 = VAG (mono), 4 = SPUI, 5 = VAGI (interleaved channels) */
	int32_t audio_frequency;    /* default 44100 */
	int32_t audio_channels;     /* SPU / VAG: 1 */
	int32_t audio_interleave;   /* SPUI / VAGI: bytes per channel per chunk, multiple of 16 (default 2048) */
	int32_t alignment;          /* default 64 (SPU / VAG), 2048 (SPUI / VAGI) */
	int32_t audio_loop_point;   /* milliseconds, < 0 = none */
	int32_t enable_loop;        /* FLAG_SPU_ENABLE_LOOP */
	int32_t no_leading_dummy;   /* FLAG_SPU_NO_LEADING_DUMMY */
	char name[16];              /* .vag name field: the output file's base name (filefmt.c:150-161) */
} psxhip_spu_file_settings_t;

/* bytes the file takes for samples_per_channel samples, or < 0 */
int64_t psxhip_spu_file_size(const psxhip_spu_file_settings_t *settings, int64_t samples_per_channel);
/* pcm: int16, channels interleaved.  Returns the bytes written (= psxhip_spu_file_size) or < 0. */
int64_t psxhip_spu_file_encode_host(int device, const psxhip_spu_file_settings_t *settings, const int16_t *pcm,
                                    int64_t samples_per_channel, uint8_t *out, size_t out_size);

/* ---------------------------------------------------------------- colour conversion + scaling front-end ---- */

/* What the reference leaves to FFmpeg's libswscale (psxavenc/decoding.c:287-311: sws_getContext(... AV_PIX_FMT_NV21,
 * SWS_BICUBIC ...), destination colourspace ITU-R BT.601 full range; :463-475: sws_scale into the frame buffer): decoded
 * pictures of any size -> NV21 frames of the encoder's size, on the device, written where the MDEC kernel reads them.
 * libswscale is absent from the reference tree and from the build image, so this step's arithmetic is this library's own
 * ("psxhip front-end v1": bicubic B = 0 / C = 0.6 like SWS_BICUBIC, 14-bit taps, 15-bit intermediates; DESIGN.md section 9,
 * restated in oracle/frontend_oracle.c) -- PARITY WITH THE REFERENCE'S SCALER IS UNPINNED.  Everything downstream of the NV21
 * frames is unaffected. */
enum {
	PSXHIP_PIX_RGB24 = 0,      /* rows of 3 * width bytes, R G B */
	PSXHIP_PIX_YUV420P = 1     /* Y plane (w * h), U plane, V plane ((w/2) * (h/2) each); w, h even */
};
typedef struct psxhip_scaler psxhip_scaler_t;
/* src_full_range: YUV input only -- 0 = limited ("MPEG") range, expanded to the full range the encoder expects
 * (decoding.c:301-311 passes the stream's own range as the source range); RGB input is full range by definition.
 * dst_width / dst_height: multiples of 16 (mdec.c:601-602), at most 1024; shrinking by more than 16x is refused. */
int psxhip_scaler_create(psxhip_scaler_t **s, int device, int src_format, int src_width, int src_height, int src_full_range,
                         int dst_width, int dst_height);
void psxhip_scaler_destroy(psxhip_scaler_t *s);
size_t psxhip_scaler_source_bytes(const psxhip_scaler_t *s);      /* bytes of one source picture */
/* n_frames pictures at d_src + i * src_stride -> NV21 frames at d_frames + i * frame_stride (4-byte aligned); asynchronous
 * on `stream`.  d_frames / frame_stride are what psxhip_mdec_encode_frames_device takes. */
int psxhip_scaler_convert_device(psxhip_scaler_t *s, const uint8_t *d_src, size_t src_stride, int n_frames, uint8_t *d_frames,
                                 size_t frame_stride, void *stream);
/* host buffers (pictures and frames back to back): H2D, kernel, D2H, synchronise */
int psxhip_scaler_convert_host(psxhip_scaler_t *s, const uint8_t *src, int n_frames, uint8_t *frames);
/* the filter banks in use (which: 0 luma horizontal, 1 luma vertical, 2 chroma horizontal, 3 chroma vertical): returns
 * the number of output positions, *taps per position; fills left[n] / coef[n * taps] when both are given (cap elements) */
int psxhip_scaler_filter(const psxhip_scaler_t *s, int which, int *taps, int32_t *left, int16_t *coef, int cap);

/* ---------------------------------------------------------------- synthetic inputs --------- */

/* Integer-only generators (same function as oracle/synth.c) so benchmarks can fill HBM directly. */
int psxhip_synth_frames_device(int device, uint8_t *d_frames, size_t frame_stride, int width, int height,
                               uint32_t seed, uint32_t first_frame, int n_frames, int noise_amp, void *stream);
int psxhip_synth_pcm_device(int device, int16_t *d_pcm, uint32_t seed, uint32_t chain, int64_t first_sample,
                            int64_t n, int kind, int pitch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PSXAV_HIP_H */
