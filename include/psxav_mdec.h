/*
 * psxav_mdec.h -- drop-in surface of the reference's MDEC BS encoder (psxavenc/mdec.h:32-74), served by
 * the MI355X library libpsxav_hip.so.
 *
 * Same type names, field names/order/types and function signatures as the reference, so its callers
 * (psxavenc/filefmt.c:425-518,547-631,635-662) compile and link unchanged against this header:
 *   - the caller still allocates state.frame_output and sets state.frame_max_size / frame_data_offset /
 *     frame_index / frame_block_* / quant_scale_sum itself (filefmt.c:428-440,637-640);
 *   - after encode_frame_bs() returns, state.frame_output[0 .. frame_max_size) and state.quant_scale,
 *     bytes_used, blocks_used, uncomp_hwords_used, quant_scale_sum hold what psxavenc/mdec.c:719-754 leaves.
 * Differences (all invisible to the reference's callers):
 *   - `dct_context` is an opaque handle owned by the library (the reference stores an FFmpeg AVDCT* there,
 *     mdec.h:50); ac_huffman_map, dc_huffman_map, coeff_clamp_map and dct_block_lists stay NULL -- the
 *     tables and each frame's coefficients live in the GPU's LDS;
 *   - a frame that fits no quant scale aborts with a message (the reference asserts, mdec.c:723);
 *   - nothing is written past frame_output[frame_max_size - 1] (the reference writes one byte past it on
 *     rejected attempts, mdec.c:323-325).
 * Each call is synchronous (H2D, kernel, D2H).  For throughput use the batched API in psxav_hip.h.
 *
 * If the including program already has the reference's args.h (format_t, bs_codec_t), define
 * PSXAV_MDEC_NO_ENUMS before including this header.
 *
 * Interface compatibility notice.  The type names, field layouts, enumerator values and function signatures declared
 * here are those of psxavenc / libpsxav (https://github.com/WonderfulToolchain/psxavenc),
 *     Copyright (c) 2019 Ben "GreaseMonkey" Russell
 *     Copyright (c) 2019, 2020, 2023 Adrian "asie" Siekierka
 * which is distributed under the zlib licence ("This software is provided 'as-is', without any express or implied
 * warranty ... 1. The origin of this software must not be misrepresented ... 2. Altered source versions must be plainly
 * marked as such ... 3. This notice may not be removed or altered from any source distribution.").  THIS FILE IS AN
 * ALTERED VERSION of that interface, re-declared for the MI355X implementation in this repository; it is not the
 * original software, and none of the reference's implementation code is part of this repository's product library.
 */
#ifndef PSXAV_MDEC_H
#define PSXAV_MDEC_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef PSXAV_MDEC_NO_ENUMS
/* psxavenc/args.h:45-58 */
typedef enum {
	FORMAT_INVALID = -1,
	FORMAT_XA,
	FORMAT_XACD,
	FORMAT_SPU,
	FORMAT_VAG,
	FORMAT_SPUI,
	FORMAT_VAGI,
	FORMAT_STR,
	FORMAT_STRCD,
	FORMAT_STRSPU,
	FORMAT_STRV,
	FORMAT_SBS
} format_t;

/* psxavenc/args.h:60-65 */
typedef enum {
	BS_CODEC_INVALID = -1,
	BS_CODEC_V2,
	BS_CODEC_V3,
	BS_CODEC_V3DC
} bs_codec_t;
#endif

/* psxavenc/mdec.h:32-55 -- layout-compatible (152 bytes on LP64) */
typedef struct {
	int frame_index;
	int frame_data_offset;
	int frame_max_size;
	int frame_block_base_overflow;
	int frame_block_overflow_num;
	int frame_block_overflow_den;
	int block_type;
	int16_t last_dc_values[3];
	uint16_t bits_value;
	int bits_left;
	uint8_t *frame_output;
	int bytes_used;
	int blocks_used;
	int uncomp_hwords_used;
	int quant_scale;
	int quant_scale_sum;

	void *dct_context;            /* reference: AVDCT*; here: the library's device context */
	uint32_t *ac_huffman_map;     /* unused, NULL */
	uint32_t *dc_huffman_map;     /* unused, NULL */
	int16_t *coeff_clamp_map;     /* unused, NULL */
	int16_t *dct_block_lists[6];  /* unused, NULL */
} mdec_encoder_state_t;

/* psxavenc/mdec.h:57-63 */
typedef struct {
	bs_codec_t video_codec;
	int video_width;
	int video_height;

	mdec_encoder_state_t state;
} mdec_encoder_t;

/* psxavenc/mdec.h:65 (mdec.c:512).  false on allocation / device failure.  The GPU is chosen by the
 * environment variable PSXAV_HIP_DEVICE (default 0). */
bool init_mdec_encoder(mdec_encoder_t *encoder, bs_codec_t video_codec, int video_width, int video_height);
/* psxavenc/mdec.h:66 (mdec.c:553); idempotent */
void destroy_mdec_encoder(mdec_encoder_t *encoder);
/* psxavenc/mdec.h:67 (mdec.c:580): one NV21 frame (host memory, w*h*3/2 bytes) -> state.frame_output */
void encode_frame_bs(mdec_encoder_t *encoder, const uint8_t *video_frame);
/* psxavenc/mdec.h:68-74 (mdec.c:757): STR video sector packetiser; returns frames consumed */
int encode_sector_str(
	mdec_encoder_t *encoder,
	format_t format,
	uint16_t str_video_id,
	const uint8_t *video_frames,
	uint8_t *output
);

#ifdef __cplusplus
}
#endif
#endif /* PSXAV_MDEC_H */
