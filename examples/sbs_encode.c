/*
 * examples/sbs_encode.c -- a C host program using the drop-in surface exactly the way the reference's
 * encode_file_sbs does (psxavenc/filefmt.c:633-662): init_mdec_encoder, the caller owns frame_output and
 * frame_max_size, one encode_frame_bs per frame, fwrite of `alignment` bytes per frame.  Input frames
 * are synthetic NV21 produced by an integer ramp + LCG noise (the reference would get them from FFmpeg).
 *
 *   gcc -std=c11 -O2 -I../include sbs_encode.c -L../psxavenc_amd -lpsxav_hip -Wl,-rpath,$PWD/../psxavenc_amd -o sbs_encode
 *   ./sbs_encode out.sbs [frames] [width] [height] [alignment] [codec 0|1|2]
 *
 * Also shows the batched entry point (psxav_hip.h) producing the same bytes in one call.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "psxav_hip.h"
#include "psxav_mdec.h"

static void make_frame(uint8_t *nv21, int w, int h, int index) {
	uint32_t lcg = 12345u + 977u * (uint32_t)index;
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) {
			lcg = lcg * 1664525u + 1013904223u;
			int v = ((x + 2 * index) % w) * 255 / w / 2 + y * 255 / h / 2 + (int)((lcg >> 24) % 9) - 4;
			nv21[y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
		}
	uint8_t *c = nv21 + w * h;
	for (int y = 0; y < h / 2; y++)
		for (int x = 0; x < w / 2; x++) {
			c[y * w + 2 * x + 0] = (uint8_t)(96 + x * 64 / (w / 2));
			c[y * w + 2 * x + 1] = (uint8_t)(160 - y * 64 / (h / 2));
		}
}

int main(int argc, char **argv) {
	const char *path = argc > 1 ? argv[1] : "out.sbs";
	const int frames = argc > 2 ? atoi(argv[2]) : 8;
	const int w = argc > 3 ? atoi(argv[3]) : 320, h = argc > 4 ? atoi(argv[4]) : 240;
	const int alignment = argc > 5 ? atoi(argv[5]) : 8192;
	const bs_codec_t codec = argc > 6 ? (bs_codec_t)atoi(argv[6]) : BS_CODEC_V2;
	const size_t fsz = (size_t)w * h * 3 / 2;

	uint8_t *all = malloc(fsz * (size_t)frames);
	for (int i = 0; i < frames; i++) make_frame(all + fsz * i, w, h, i);

	/* ---- the reference's call pattern */
	mdec_encoder_t encoder;
	if (!init_mdec_encoder(&encoder, codec, w, h)) {
		fprintf(stderr, "init_mdec_encoder failed\n");
		return 1;
	}
	encoder.state.frame_output = malloc((size_t)alignment);
	encoder.state.frame_data_offset = 0;
	encoder.state.frame_max_size = alignment;
	encoder.state.quant_scale_sum = 0;

	FILE *out = fopen(path, "wb");
	if (!out) return 1;
	uint8_t *serial = malloc((size_t)alignment * (size_t)frames);
	for (int j = 0; j < frames; j++) {
		encode_frame_bs(&encoder, all + fsz * j);
		fwrite(encoder.state.frame_output, (size_t)alignment, 1, out);
		memcpy(serial + (size_t)alignment * j, encoder.state.frame_output, (size_t)alignment);
	}
	fclose(out);
	printf("wrote %d frames to %s, average quant scale %.2f\n", frames, path,
	       (double)encoder.state.quant_scale_sum / (double)frames);
	free(encoder.state.frame_output);
	destroy_mdec_encoder(&encoder);

	/* ---- the batched extension: same bytes, one call */
	psxhip_mdec_ctx_t *ctx = NULL;
	if (psxhip_mdec_create(&ctx, 0, (int)codec, w, h, alignment) != PSXHIP_OK) {
		fprintf(stderr, "%s\n", psxhip_last_error());
		return 1;
	}
	uint8_t *batched = malloc((size_t)alignment * (size_t)frames);
	psxhip_mdec_result_t *res = malloc(sizeof(*res) * (size_t)frames);
	if (psxhip_mdec_encode_frames_host(ctx, all, frames, NULL, alignment, batched, (size_t)alignment, res) != PSXHIP_OK) {
		fprintf(stderr, "%s\n", psxhip_last_error());
		return 1;
	}
	psxhip_mdec_destroy(ctx);
	const int same = memcmp(serial, batched, (size_t)alignment * (size_t)frames) == 0;
	printf("batched path identical to per-frame path: %s\n", same ? "yes" : "NO");
	free(all); free(serial); free(batched); free(res);
	return same ? 0 : 2;
}
