/*
 * oracle/synth.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Integer-only, counter-based synthetic inputs (no reference counterpart: the reference
 * takes decoded video/audio from FFmpeg, psxavenc/decoding.c, which is out of scope).
 * Every sample is a pure function of (seed, frame-or-chain index, sample index), so any
 * rank can generate any frame independently; psxavenc_amd/csrc/synth.hip computes the
 * same function on the device and tests/test_synth.py compares the two.
 *
 * Video: NV21 (Y plane, then interleaved Cr,Cb at half resolution -- the layout
 * encode_frame_bs consumes, mdec.c:585-594,627-628).  A diagonal luma ramp that drifts
 * with the frame index plus uniform noise of amplitude +-A; smooth chroma ramps with
 * noise +-A/2.  A=4 / A=8 are the two classes BASELINE.md names.
 */
#include <stdint.h>
#include "mdec_oracle.h"

static inline uint32_t mix32(uint32_t x) {
	x ^= x >> 16; x *= 0x7FEB352Du;
	x ^= x >> 15; x *= 0x846CA68Bu;
	x ^= x >> 16;
	return x;
}

static inline int clamp_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

static inline int noise(uint32_t key, uint32_t idx, int amp) {
	if (amp <= 0) return 0;
	const uint32_t r = mix32(key ^ (idx * 0x85EBCA77u + 0x165667B1u));
	return (int)(r % (uint32_t)(2 * amp + 1)) - amp;
}

void orc_synth_frame(int w, int h, uint32_t seed, uint32_t frame_index, int noise_amp, uint8_t *nv21) {
	const uint32_t key = mix32(seed + frame_index * 0x9E3779B1u);
	const int cw = w / 2, ch = h / 2;
	uint32_t idx = 0;
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++, idx++) {
			const int xs = (int)(((uint32_t)x + 3u * frame_index) % (uint32_t)w);
			const int base = (xs * 255 / w + y * 255 / h) / 2;
			nv21[idx] = (uint8_t)clamp_u8(base + noise(key, idx, noise_amp));
		}
	for (int y = 0; y < ch; y++)
		for (int x = 0; x < cw; x++, idx += 2) {
			const int xs = (int)(((uint32_t)x + frame_index) % (uint32_t)cw);
			const int cr = 128 + xs * 64 / cw - 32;
			const int cb = 128 + 32 - y * 64 / ch;
			nv21[idx + 0] = (uint8_t)clamp_u8(cr + noise(key, idx + 0, noise_amp / 2));
			nv21[idx + 1] = (uint8_t)clamp_u8(cb + noise(key, idx + 1, noise_amp / 2));
		}
}

/* parabolic "sine": 16-bit phase -> [-32768, 32768] */
static inline int par_sin(uint32_t phase) {
	const int t = (int)(phase & 0x7FFFu);
	const int v = (t * (32768 - t)) >> 13;
	return (phase & 0x8000u) ? -v : v;
}

/*
 * kind 0: loud two-tone + noise   kind 1: quiet tone + tiny noise   kind 2: full-scale noise
 * kind 3: silence                 kind 4: pure loud tone, no noise  kind 5: half silent (1 s on / 1 s off @ 32768)
 */
void orc_synth_pcm(uint32_t seed, uint32_t chain, int64_t first_sample, int n, int kind, int16_t *pcm) {
	const uint32_t key = mix32(seed ^ (chain * 0xC2B2AE35u + 0x27D4EB2Fu));
	const uint32_t step1 = 700u + 37u * (chain % 16u), step2 = 2311u + 101u * (chain % 7u);
	for (int i = 0; i < n; i++) {
		const uint64_t s = (uint64_t)(first_sample + i);
		const uint32_t s32 = (uint32_t)s;
		int v;
		switch (kind) {
		case 0: v = ((12000 * par_sin(s32 * step1)) >> 15) + ((6000 * par_sin(s32 * step2)) >> 15) + noise(key, s32, 300); break;
		case 1: v = ((200 * par_sin(s32 * step1)) >> 15) + noise(key, s32, 3); break;
		case 2: v = noise(key, s32, 32767); break;
		case 3: v = 0; break;
		case 4: v = (16384 * par_sin(s32 * step1)) >> 15; break;
		default: v = ((s >> 15) & 1) ? 0 : ((12000 * par_sin(s32 * step1)) >> 15) + noise(key, s32, 300); break;
		}
		pcm[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
	}
}
