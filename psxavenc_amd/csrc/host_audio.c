/*
 * host_audio.c -- libpsxav's audio call surface (include/psxav_audio.h) in host C over the HIP library.
 * Replaces the public functions of libpsxav/adpcm.c:235-401; the search itself (adpcm.c:39-191) and the
 * sector assembly run on the GPU via psxhip_{spu,xa}_encode_streams_host.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "psxav_audio.h"
#include "psxav_hip.h"

static int env_device(void) {
	const char *e = getenv("PSXAV_HIP_DEVICE");
	return e ? atoi(e) : 0;
}

static void die(const char *who) {
	fprintf(stderr, "%s: %s\n", who, psxhip_last_error());
	abort();
}

/* ---- size helpers (adpcm.c:235-260) ---- */
uint32_t psx_audio_xa_get_buffer_size_per_sector(psx_audio_xa_settings_t settings) {
	return settings.format == PSX_AUDIO_XA_FORMAT_XA ? 2336u : 2352u;
}

uint32_t psx_audio_xa_get_samples_per_sector(psx_audio_xa_settings_t settings) {
	uint32_t per_group = settings.bits_per_sample == 8 ? 112u : 224u;
	if (settings.stereo) per_group >>= 1;
	return per_group * 18u;
}

uint32_t psx_audio_xa_get_sector_interleave(psx_audio_xa_settings_t settings) {
	uint32_t n = settings.stereo ? 2u : 4u;
	if (settings.frequency == PSX_AUDIO_XA_FREQ_SINGLE) n *= 2u;
	if (settings.bits_per_sample == 4) n *= 2u;
	return n;
}

uint32_t psx_audio_xa_get_buffer_size(psx_audio_xa_settings_t settings, int sample_count) {
	const int per_sector = (int)psx_audio_xa_get_samples_per_sector(settings);
	const int sectors = (sample_count + per_sector - 1) / per_sector;
	return (uint32_t)sectors * psx_audio_xa_get_buffer_size_per_sector(settings);
}

uint32_t psx_audio_spu_get_buffer_size(int sample_count) {
	return (uint32_t)((sample_count + PSX_AUDIO_SPU_SAMPLES_PER_BLOCK - 1) / PSX_AUDIO_SPU_SAMPLES_PER_BLOCK) * PSX_AUDIO_SPU_BLOCK_SIZE;
}

/* ---- encoders ---- */
int psx_audio_spu_encode(psx_audio_encoder_channel_state_t *state, const int16_t *samples, int sample_count, int pitch,
                         uint8_t *output) {
	if (sample_count <= 0) return 0;
	psxhip_adpcm_state_t st = {state->prev1, state->prev2};
	const int bytes = (int)psx_audio_spu_get_buffer_size(sample_count);
	const int rc = psxhip_spu_encode_streams_host(env_device(), samples, 1, (int64_t)sample_count * pitch, pitch, sample_count,
	                                              &st, output, bytes);
	if (rc < 0) die("psx_audio_spu_encode");
	state->prev1 = st.prev1;
	state->prev2 = st.prev2;
	state->mse = 0;
	return rc;
}

int psx_audio_spu_encode_simple(const int16_t *samples, int sample_count, uint8_t *output, int loop_start) {
	psx_audio_encoder_channel_state_t st;
	memset(&st, 0, sizeof st);
	int length = psx_audio_spu_encode(&st, samples, sample_count, 1, output);
	if (length >= PSX_AUDIO_SPU_BLOCK_SIZE) {
		if (loop_start < 0) {
			/* trailing block that traps the SPU in a silent loop (adpcm.c:386-391) */
			memset(output + length, 0, PSX_AUDIO_SPU_BLOCK_SIZE);
			output[length + 1] = PSX_AUDIO_SPU_LOOP_TRAP;
			length += PSX_AUDIO_SPU_BLOCK_SIZE;
		} else {
			output[length - PSX_AUDIO_SPU_BLOCK_SIZE + 1] |= PSX_AUDIO_SPU_LOOP_REPEAT;
			output[loop_start / PSX_AUDIO_SPU_SAMPLES_PER_BLOCK * PSX_AUDIO_SPU_BLOCK_SIZE + 1] |= PSX_AUDIO_SPU_LOOP_START;
		}
	}
	return length;
}

int psx_audio_xa_encode(psx_audio_xa_settings_t settings, psx_audio_encoder_state_t *state, const int16_t *samples,
                        int sample_count, int lba, uint8_t *output) {
	if (sample_count <= 0) return 0;
	psxhip_adpcm_state_t st[2] = {{state->left.prev1, state->left.prev2}, {state->right.prev1, state->right.prev2}};
	const int32_t first_lba = lba;
	const int ch = settings.stereo ? 2 : 1;
	const int rc = psxhip_xa_encode_streams_host(env_device(), (int)settings.format, settings.stereo ? 1 : 0, settings.frequency,
	                                             settings.bits_per_sample, settings.file_number, settings.channel_number,
	                                             samples, 1, (int64_t)sample_count * ch, sample_count, &first_lba, st, output,
	                                             0, 0);
	if (rc < 0) die("psx_audio_xa_encode");
	state->left.prev1 = st[0].prev1;
	state->left.prev2 = st[0].prev2;
	if (settings.stereo) {
		state->right.prev1 = st[1].prev1;
		state->right.prev2 = st[1].prev2;
	}
	return rc;
}

void psx_audio_xa_encode_finalize(psx_audio_xa_settings_t settings, uint8_t *output, int output_length) {
	(void)settings;
	if (output_length >= 2336) {
		/* last sector viewed as a raw 2352-byte sector; for .xa output that view starts 16 bytes before it,
		 * and only the subheader at +16..+23 is touched (adpcm.c:334-340) */
		uint8_t *raw = output + output_length - PSX_CDROM_SECTOR_SIZE;
		raw[18] |= PSX_CDROM_SECTOR_XA_SUBMODE_EOF;
		memcpy(raw + 20, raw + 16, 4);
	}
}

int psx_audio_xa_encode_simple(psx_audio_xa_settings_t settings, const int16_t *samples, int sample_count, int lba,
                               uint8_t *output) {
	psx_audio_encoder_state_t st;
	memset(&st, 0, sizeof st);
	const int length = psx_audio_xa_encode(settings, &st, samples, sample_count, lba, output);
	psx_audio_xa_encode_finalize(settings, output, length);
	return length;
}
