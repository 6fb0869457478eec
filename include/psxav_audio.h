/*
 * psxav_audio.h -- drop-in surface of libpsxav's ADPCM encoder and CD-ROM sector helpers
 * (libpsxav/libpsxav.h:29-176), served by the MI355X library libpsxav_hip.so.
 *
 * Same type names, layouts and signatures as the reference.  The filter x shift search and the XA sector
 * assembly run on the GPU (psxav_hip.h); state structs are caller-owned and keep the reference's
 * prev1 / prev2 meaning, so calls can be chained exactly like the reference's (28 samples per call for
 * -t spu, one sector per call for xa / str: filefmt.c:184,243,487).
 * Differences:
 *   - psx_audio_xa_encode() writes every byte of each sector; the reference leaves some bytes of the
 *     caller's buffer untouched (the 20 bytes before the EDC, 8-bit group bytes 8..15, and ORs `coding`
 *     onto whatever was there for .xa output; adpcm.c:277-288,321-322).  Output equals the reference's
 *     when the caller zero-fills the sector first.
 *   - input is never read past sample_count; the reference's stereo path reads up to ~2x past it and
 *     relies on >= 4032 zero samples of padding (adpcm.c:307-308 vs :65,110).  Output equals the
 *     reference's on zero-padded input.
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
#ifndef PSXAV_AUDIO_H
#define PSXAV_AUDIO_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSX_AUDIO_SPU_BLOCK_SIZE        16
#define PSX_AUDIO_SPU_SAMPLES_PER_BLOCK 28

enum {
	PSX_AUDIO_XA_FREQ_SINGLE = 18900,
	PSX_AUDIO_XA_FREQ_DOUBLE = 37800
};

typedef enum {
	PSX_AUDIO_XA_FORMAT_XA,   /* 2336-byte sectors (.xa file) */
	PSX_AUDIO_XA_FORMAT_XACD  /* 2352-byte raw sectors */
} psx_audio_xa_format_t;

/* libpsxav.h:44-51 */
typedef struct {
	psx_audio_xa_format_t format;
	bool stereo;
	int frequency;        /* 18900 or 37800 */
	int bits_per_sample;  /* 4 or 8 */
	int file_number;      /* 00-FF */
	int channel_number;   /* 00-1F */
} psx_audio_xa_settings_t;

/* libpsxav.h:53-57; qerr is never updated and mse is scratch in the reference (adpcm.c:107,131-132) */
typedef struct {
	int qerr;
	uint64_t mse;
	int prev1, prev2;
} psx_audio_encoder_channel_state_t;

/* libpsxav.h:59-62 */
typedef struct {
	psx_audio_encoder_channel_state_t left;
	psx_audio_encoder_channel_state_t right;
} psx_audio_encoder_state_t;

/* libpsxav.h:64-71 */
enum {
	PSX_AUDIO_SPU_LOOP_END    = 1 << 0,
	PSX_AUDIO_SPU_LOOP_REPEAT = (1 << 0) | (1 << 1),
	PSX_AUDIO_SPU_LOOP_START  = (1 << 1) | (1 << 2),
	PSX_AUDIO_SPU_LOOP_TRAP   = (1 << 0) | (1 << 2)
};

/* libpsxav.h:73-101 */
uint32_t psx_audio_xa_get_buffer_size(psx_audio_xa_settings_t settings, int sample_count);
uint32_t psx_audio_spu_get_buffer_size(int sample_count);
uint32_t psx_audio_xa_get_buffer_size_per_sector(psx_audio_xa_settings_t settings);
uint32_t psx_audio_xa_get_samples_per_sector(psx_audio_xa_settings_t settings);
uint32_t psx_audio_xa_get_sector_interleave(psx_audio_xa_settings_t settings);
int psx_audio_xa_encode(
	psx_audio_xa_settings_t settings,
	psx_audio_encoder_state_t *state,
	const int16_t *samples,
	int sample_count,
	int lba,
	uint8_t *output
);
int psx_audio_xa_encode_simple(
	psx_audio_xa_settings_t settings,
	const int16_t *samples,
	int sample_count,
	int lba,
	uint8_t *output
);
int psx_audio_spu_encode(
	psx_audio_encoder_channel_state_t *state,
	const int16_t *samples,
	int sample_count,
	int pitch,
	uint8_t *output
);
int psx_audio_spu_encode_simple(const int16_t *samples, int sample_count, uint8_t *output, int loop_start);
void psx_audio_xa_encode_finalize(psx_audio_xa_settings_t settings, uint8_t *output, int output_length);

/* ---- CD-ROM sector helpers, libpsxav.h:105-176 (host code; cdrom.c) ---- */

#define PSX_CDROM_SECTOR_SIZE 2352

typedef struct {
	uint8_t minute;
	uint8_t second;
	uint8_t sector;
	uint8_t mode;
} psx_cdrom_sector_header_t;

typedef struct {
	uint8_t file;
	uint8_t channel;
	uint8_t submode;
	uint8_t coding;
} psx_cdrom_sector_xa_subheader_t;

typedef struct {
	uint8_t sync[12];
	psx_cdrom_sector_header_t header;
	uint8_t data[0x920];
} psx_cdrom_sector_mode1_t;

typedef struct {
	uint8_t sync[12];
	psx_cdrom_sector_header_t header;
	psx_cdrom_sector_xa_subheader_t subheader[2];
	uint8_t data[0x918];
} psx_cdrom_sector_mode2_t;

typedef union {
	psx_cdrom_sector_mode1_t mode1;
	psx_cdrom_sector_mode2_t mode2;
} psx_cdrom_sector_t;

#define PSX_CDROM_SECTOR_XA_CHANNEL_MASK 0x1F

enum {
	PSX_CDROM_SECTOR_XA_SUBMODE_EOR     = 1 << 0,
	PSX_CDROM_SECTOR_XA_SUBMODE_VIDEO   = 1 << 1,
	PSX_CDROM_SECTOR_XA_SUBMODE_AUDIO   = 1 << 2,
	PSX_CDROM_SECTOR_XA_SUBMODE_DATA    = 1 << 3,
	PSX_CDROM_SECTOR_XA_SUBMODE_TRIGGER = 1 << 4,
	PSX_CDROM_SECTOR_XA_SUBMODE_FORM2   = 1 << 5,
	PSX_CDROM_SECTOR_XA_SUBMODE_RT      = 1 << 6,
	PSX_CDROM_SECTOR_XA_SUBMODE_EOF     = 1 << 7
};

enum {
	PSX_CDROM_SECTOR_XA_CODING_MONO         = 0 << 0,
	PSX_CDROM_SECTOR_XA_CODING_STEREO       = 1 << 0,
	PSX_CDROM_SECTOR_XA_CODING_CHANNEL_MASK = 3 << 0,
	PSX_CDROM_SECTOR_XA_CODING_FREQ_DOUBLE  = 0 << 2,
	PSX_CDROM_SECTOR_XA_CODING_FREQ_SINGLE  = 1 << 2,
	PSX_CDROM_SECTOR_XA_CODING_FREQ_MASK    = 3 << 2,
	PSX_CDROM_SECTOR_XA_CODING_BITS_4       = 0 << 4,
	PSX_CDROM_SECTOR_XA_CODING_BITS_8       = 1 << 4,
	PSX_CDROM_SECTOR_XA_CODING_BITS_MASK    = 3 << 4,
	PSX_CDROM_SECTOR_XA_CODING_EMPHASIS     = 1 << 6
};

typedef enum {
	PSX_CDROM_SECTOR_TYPE_MODE1,
	PSX_CDROM_SECTOR_TYPE_MODE2_FORM1,
	PSX_CDROM_SECTOR_TYPE_MODE2_FORM2
} psx_cdrom_sector_type_t;

void psx_cdrom_init_xa_subheader(psx_cdrom_sector_xa_subheader_t *subheader, psx_cdrom_sector_type_t type);
void psx_cdrom_init_sector(psx_cdrom_sector_t *sector, int lba, psx_cdrom_sector_type_t type);
void psx_cdrom_calculate_checksums(psx_cdrom_sector_t *sector, psx_cdrom_sector_type_t type);

#ifdef __cplusplus
}
#endif
#endif /* PSXAV_AUDIO_H */
