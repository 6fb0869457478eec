/* oracle/adpcm_oracle.h -- TEST INFRASTRUCTURE ONLY (see adpcm_oracle.c). */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int prev1, prev2; } orc_adpcm_chan_t;        /* the live part of libpsxav.h:53-57 */
typedef struct { orc_adpcm_chan_t left, right; } orc_adpcm_state_t;

typedef struct {
	int format;           /* 0 = .xa (2336-byte sectors), 1 = XACD (2352) -- libpsxav.h:39-42 */
	int stereo;
	int frequency;        /* 18900 / 37800 */
	int bits_per_sample;  /* 4 / 8 */
	int file_number, channel_number;
} orc_xa_settings_t;

int orc_adpcm_find_min_shift(const orc_adpcm_chan_t *st, const int16_t *samples, int limit, int pitch, int filter, int range);
uint64_t orc_adpcm_trial(orc_adpcm_chan_t *st, const int16_t *samples, int limit, int pitch,
                         int filter, int shift, int range, uint8_t codes[28]);
uint8_t orc_adpcm_encode_unit(orc_adpcm_chan_t *st, const int16_t *samples, int limit, int pitch,
                              int filter_count, int range, uint8_t codes[28]);

int orc_spu_encode(orc_adpcm_chan_t *st, const int16_t *samples, int sample_count, int pitch, uint8_t *out);
int orc_spu_encode_simple(const int16_t *samples, int sample_count, uint8_t *out, int loop_start);

int orc_xa_samples_per_sector(orc_xa_settings_t s);
int orc_xa_sector_size(orc_xa_settings_t s);
int orc_xa_sector_interleave(orc_xa_settings_t s);
int orc_xa_encode(orc_xa_settings_t s, orc_adpcm_state_t *st, const int16_t *samples, int sample_count, int lba, uint8_t *out);
void orc_xa_encode_finalize(orc_xa_settings_t s, uint8_t *out, int out_len);

uint32_t orc_edc_crc32(const uint8_t *data, int len);
void orc_cdrom_init_sector(uint8_t *sector, int lba, int type);
void orc_cdrom_calculate_checksums(uint8_t *sector, int type);

#ifdef __cplusplus
}
#endif
