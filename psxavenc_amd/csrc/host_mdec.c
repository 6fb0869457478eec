/*
 * host_mdec.c -- the reference's MDEC call surface (include/psxav_mdec.h) in host C over the HIP library.
 *
 * init_mdec_encoder / destroy_mdec_encoder / encode_frame_bs replace psxavenc/mdec.c:512-755 as thin
 * wrappers around psxhip_mdec_* (one frame per call); encode_sector_str restates the STR packetiser
 * (mdec.c:757-836), which is host-side byte shuffling around encode_frame_bs.  No encoding is done here.
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "psxav_hip.h"
#include "psxav_mdec.h"

/* what lives behind mdec_encoder_state_t.dct_context */
typedef struct {
	psxhip_mdec_ctx_t *ctx;
	int capacity;    /* largest frame_max_size the device context was sized for */
	int device;
} host_mdec_handle_t;

static int env_device(void) {
	const char *e = getenv("PSXAV_HIP_DEVICE");
	return e ? atoi(e) : 0;
}

static int ensure_capacity(mdec_encoder_t *enc, host_mdec_handle_t *h, int frame_max_size) {
	if (h->ctx && frame_max_size <= h->capacity) return 0;
	if (h->ctx) {
		psxhip_mdec_destroy(h->ctx);
		h->ctx = NULL;
	}
	/* round the LDS staging up so that slowly varying STR budgets do not re-create the context */
	int cap = frame_max_size < 8192 ? 8192 : ((frame_max_size + 4095) / 4096) * 4096;
	int rc = psxhip_mdec_create(&h->ctx, h->device, (int)enc->video_codec, enc->video_width, enc->video_height, cap);
	if (rc != PSXHIP_OK) {
		h->ctx = NULL;
		return rc;
	}
	h->capacity = cap;
	return 0;
}

bool init_mdec_encoder(mdec_encoder_t *encoder, bs_codec_t video_codec, int video_width, int video_height) {
	encoder->video_codec = video_codec;
	encoder->video_width = video_width;
	encoder->video_height = video_height;

	mdec_encoder_state_t *st = &encoder->state;
	st->ac_huffman_map = NULL;
	st->dc_huffman_map = NULL;
	st->coeff_clamp_map = NULL;
	for (int i = 0; i < 6; i++) st->dct_block_lists[i] = NULL;

	host_mdec_handle_t *h = calloc(1, sizeof(*h));
	st->dct_context = h;
	if (!h) return false;
	h->device = env_device();
	/* create the device context now so that a missing GPU / bad geometry is reported by init, like the
	 * reference reports allocation failure (mdec.c:529-535) */
	if (ensure_capacity(encoder, h, 8192) != 0) {
		fprintf(stderr, "init_mdec_encoder: %s\n", psxhip_last_error());
		free(h);
		st->dct_context = NULL;
		return false;
	}
	return true;
}

void destroy_mdec_encoder(mdec_encoder_t *encoder) {
	mdec_encoder_state_t *st = &encoder->state;
	host_mdec_handle_t *h = st->dct_context;
	if (h) {
		if (h->ctx) psxhip_mdec_destroy(h->ctx);
		free(h);
		st->dct_context = NULL;
	}
}

void encode_frame_bs(mdec_encoder_t *encoder, const uint8_t *video_frame) {
	mdec_encoder_state_t *st = &encoder->state;
	host_mdec_handle_t *h = st->dct_context;

	assert(h);                                      /* mdec.c:583 */
	assert((encoder->video_width % 16) == 0);       /* mdec.c:601-602 */
	assert((encoder->video_height % 16) == 0);

	if (ensure_capacity(encoder, h, st->frame_max_size) != 0) {
		fprintf(stderr, "encode_frame_bs: %s\n", psxhip_last_error());
		abort();
	}
	psxhip_mdec_result_t r;
	int rc = psxhip_mdec_encode_frames_host(h->ctx, video_frame, 1, NULL, st->frame_max_size, st->frame_output,
	                                        (size_t)st->frame_max_size, &r);
	if (rc != PSXHIP_OK) {
		/* PSXHIP_ENOFIT is the reference's assert(state->quant_scale < 64), mdec.c:723 */
		fprintf(stderr, "encode_frame_bs: %s\n", psxhip_last_error());
		abort();
	}
	/* what one successful pass of mdec.c:663-736 leaves behind */
	st->quant_scale = r.quant_scale;
	st->quant_scale_sum += r.quant_scale;
	st->bytes_used = r.bytes_used;
	st->blocks_used = r.blocks_used;
	st->uncomp_hwords_used = r.uncomp_hwords_used;
	st->block_type = 0;
	st->bits_value = 0;
	st->bits_left = 16;
}

static void put_le16(uint8_t *p, unsigned v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put_le32(uint8_t *p, unsigned v) { put_le16(p, v & 0xFFFF); put_le16(p + 2, v >> 16); }

int encode_sector_str(mdec_encoder_t *encoder, format_t format, uint16_t str_video_id, const uint8_t *video_frames,
                      uint8_t *output) {
	mdec_encoder_state_t *st = &encoder->state;
	int consumed = 0;
	/* the reference steps by w*h*2 per frame here (mdec.c:765,778) although NV21 frames are w*h*3/2 apart;
	 * kept, since its only caller hands over one frame at a time and the loop runs at most once */
	const size_t step = (size_t)encoder->video_width * (size_t)encoder->video_height * 2;

	while (st->frame_data_offset >= st->frame_max_size) {
		st->frame_index++;
		st->frame_block_overflow_num += st->frame_block_base_overflow;
		st->frame_max_size = st->frame_block_overflow_num / st->frame_block_overflow_den * 2016;
		st->frame_block_overflow_num %= st->frame_block_overflow_den;
		st->frame_data_offset = 0;

		encode_frame_bs(encoder, video_frames);
		video_frames += step;
		consumed++;
	}

	/* 32-byte chunk header (mdec.c:782-820) */
	uint8_t hd[32] = {0};
	put_le16(hd + 0x00, 0x0160);
	put_le16(hd + 0x02, str_video_id);
	put_le16(hd + 0x04, (unsigned)(st->frame_data_offset / 2016));
	put_le16(hd + 0x06, (unsigned)(st->frame_max_size / 2016));
	put_le32(hd + 0x08, (unsigned)st->frame_index);
	put_le32(hd + 0x0C, (unsigned)st->bytes_used);
	put_le16(hd + 0x10, (unsigned)encoder->video_width);
	put_le16(hd + 0x12, (unsigned)encoder->video_height);
	memcpy(hd + 0x14, st->frame_output, 8);

	const int at = format == FORMAT_STR ? 0x08 : (format == FORMAT_STRCD ? 0x18 : 0x00);
	memcpy(output + at, hd, sizeof hd);
	memcpy(output + at + 0x20, st->frame_output + st->frame_data_offset, 2016);
	st->frame_data_offset += 2016;
	return consumed;
}
