/* oracle/mdec_oracle.h -- TEST INFRASTRUCTURE ONLY (see mdec_oracle.c). */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BS_V2 = 0, ORC_BS_V3 = 1, ORC_BS_V3DC = 2 };          /* bs_codec_t, args.h:61-65 */
enum { ORC_FMT_STR = 6, ORC_FMT_STRCD = 7, ORC_FMT_STRV = 9 };   /* format_t, args.h:45-58 */
enum { ORC_MDEC_EINVAL = -1, ORC_MDEC_ENOFIT = -2, ORC_MDEC_EDCRANGE = -3 };

typedef struct {
	int quant_scale;
	int bytes_used;
	int blocks_used;
	int uncomp_hwords_used;
} orc_mdec_result_t;

/* the STR packetiser's carried fields (subset of mdec_encoder_state_t, mdec.h:32-55) */
typedef struct {
	int frame_index, frame_data_offset, frame_max_size;
	int base_overflow, overflow_num, overflow_den;
	int bytes_used, quant_scale_sum;
	uint8_t *frame_output;
} orc_str_state_t;

void orc_fdct_islow8(int16_t *blk);
void orc_fdct_islow8_pass1(int16_t *blk, int pass1);   /* 4 = orc_fdct_islow8; 2 = the IJG original (checked against libjpeg-turbo) */
uint32_t orc_mdec_ac_code(int run, int level);
uint32_t orc_mdec_dc_code(int comp, int delta);
void orc_mdec_frame_to_coefs(int w, int h, const uint8_t *nv21, int16_t *coefs);
int orc_mdec_encode_frame(int codec, int w, int h, const uint8_t *nv21, int frame_max_size,
                          uint8_t *out, orc_mdec_result_t *res);
int orc_mdec_encode_frames(int codec, int w, int h, const uint8_t *frames, int n_frames,
                           const int *frame_max_sizes, int out_stride, uint8_t *out,
                           orc_mdec_result_t *res);
int orc_mdec_encode_sector_str(orc_str_state_t *st, int codec, int w, int h, int format,
                               uint16_t str_video_id, const uint8_t *video_frames, uint8_t *output);

/* oracle/mdec_decode.c -- BS bitstream reader used by the round-trip tests */
int orc_mdec_decode_frame(int w, int h, const uint8_t *bs, int bs_size, int16_t *levels /* [6*nmb*64], encode order */,
                          int *quant_scale, int *version, int *bits_consumed, int v3dc_wrap);
void orc_mdec_reconstruct(int w, int h, const int16_t *levels, int quant_scale, uint8_t *nv21);

/* oracle/synth.c -- integer-only synthetic NV21 / PCM generators (CPU twin of csrc/synth.hip) */
void orc_synth_frame(int w, int h, uint32_t seed, uint32_t frame_index, int noise_amp, uint8_t *nv21);
void orc_synth_pcm(uint32_t seed, uint32_t chain, int64_t first_sample, int n, int kind, int16_t *pcm);

#ifdef __cplusplus
}
#endif
