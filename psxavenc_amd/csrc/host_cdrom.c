/*
 * host_cdrom.c -- CD-ROM sector helpers of include/psxav_audio.h in host C (libpsxav/cdrom.c:45-111).
 * Used by callers that build video (mode 2 form 1) sectors around GPU-encoded frames; the audio sectors'
 * own headers and EDC are produced on the GPU (adpcm_kernels.hip).  Table CRC (slicing-by-8) instead of the
 * reference's bit-serial loop (cdrom.c:30-41): same polynomial, same result.
 */
#include <pthread.h>
#include <string.h>

#include "psxav_audio.h"

_Static_assert(sizeof(psx_cdrom_sector_mode1_t) == PSX_CDROM_SECTOR_SIZE, "mode 1 sector layout");
_Static_assert(sizeof(psx_cdrom_sector_mode2_t) == PSX_CDROM_SECTOR_SIZE, "mode 2 sector layout");

/* slicing-by-8: edc_table[k][b] = CRC of byte b followed by k zero bytes; eight table look-ups per 8 input bytes */
static uint32_t edc_table[8][256];
static pthread_once_t edc_table_once = PTHREAD_ONCE_INIT;     /* one writer, whoever calls first (the library is re-entrant per object) */

static void edc_init(void) {
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t v = i;
		for (int k = 0; k < 8; k++) v = (v & 1u) ? (v >> 1) ^ 0xD8018001u : v >> 1;
		edc_table[0][i] = v;
	}
	for (uint32_t i = 0; i < 256; i++)
		for (int k = 1; k < 8; k++) {
			const uint32_t v = edc_table[k - 1][i];
			edc_table[k][i] = (v >> 8) ^ edc_table[0][v & 0xFF];
		}
}

static uint32_t edc(const uint8_t *p, int n) {
	(void)pthread_once(&edc_table_once, edc_init);
	uint32_t v = 0;
	while (n >= 8) {
		uint32_t lo, hi;
		memcpy(&lo, p, 4);
		memcpy(&hi, p + 4, 4);
		lo ^= v;                         /* little-endian host (x86-64 / the GPU box) */
		v = edc_table[7][lo & 0xFF] ^ edc_table[6][(lo >> 8) & 0xFF] ^ edc_table[5][(lo >> 16) & 0xFF] ^ edc_table[4][lo >> 24] ^
		    edc_table[3][hi & 0xFF] ^ edc_table[2][(hi >> 8) & 0xFF] ^ edc_table[1][(hi >> 16) & 0xFF] ^ edc_table[0][hi >> 24];
		p += 8;
		n -= 8;
	}
	while (n--) v = (v >> 8) ^ edc_table[0][(v ^ *p++) & 0xFF];
	return v;
}

static uint8_t bcd(int v) { return (uint8_t)((v / 10) * 16 + v % 10); }

void psx_cdrom_init_xa_subheader(psx_cdrom_sector_xa_subheader_t *subheader, psx_cdrom_sector_type_t type) {
	memset(subheader, 0, 2 * sizeof *subheader);
	subheader[0].submode = PSX_CDROM_SECTOR_XA_SUBMODE_DATA;
	if (type == PSX_CDROM_SECTOR_TYPE_MODE2_FORM2) subheader[0].submode |= PSX_CDROM_SECTOR_XA_SUBMODE_FORM2;
	subheader[1] = subheader[0];
}

void psx_cdrom_init_sector(psx_cdrom_sector_t *sector, int lba, psx_cdrom_sector_type_t type) {
	uint8_t *sync = sector->mode1.sync;
	sync[0] = 0x00;
	memset(sync + 1, 0xFF, 10);
	sync[11] = 0x00;

	const int t = lba + 150;               /* 2-second pregap */
	sector->mode1.header.minute = bcd(t / 4500);
	sector->mode1.header.second = bcd((t / 75) % 60);
	sector->mode1.header.sector = bcd(t % 75);

	if (type == PSX_CDROM_SECTOR_TYPE_MODE1) {
		sector->mode1.header.mode = 0x01;
	} else {
		sector->mode2.header.mode = 0x02;
		psx_cdrom_init_xa_subheader(sector->mode2.subheader, type);
	}
}

void psx_cdrom_calculate_checksums(psx_cdrom_sector_t *sector, psx_cdrom_sector_type_t type) {
	uint8_t *raw = (uint8_t *)sector;
	int from, len, at;
	switch (type) {
	case PSX_CDROM_SECTOR_TYPE_MODE1:       from = 0x00; len = 0x810; at = 0x810; break;
	case PSX_CDROM_SECTOR_TYPE_MODE2_FORM1: from = 0x10; len = 0x808; at = 0x818; break;
	default:                                from = 0x10; len = 0x91C; at = 0x92C; break;
	}
	const uint32_t v = edc(raw + from, len);
	raw[at + 0] = (uint8_t)v;
	raw[at + 1] = (uint8_t)(v >> 8);
	raw[at + 2] = (uint8_t)(v >> 16);
	raw[at + 3] = (uint8_t)(v >> 24);
	/* mode 1: the reference then clears 8 bytes at a mis-scaled address (cdrom.c:88, `sector + 0x814` on a
	 * psx_cdrom_sector_t*); the 8 reserved bytes after the EDC are cleared here instead.  ECC: not computed
	 * (cdrom.c:90,99 "TODO"). */
	if (type == PSX_CDROM_SECTOR_TYPE_MODE1) memset(raw + 0x814, 0, 8);
}
