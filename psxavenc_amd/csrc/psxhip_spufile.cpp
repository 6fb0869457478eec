// psxhip_spufile.cpp -- SPU / VAG / SPUI / VAGI container framing (include/psxav_hip.h, psxhip_spu_file_*).
//
// What psxavenc's encode_file_spu / encode_file_spui (psxavenc/filefmt.c:212-389, .vag header :95-162) wrap around
// psx_audio_spu_encode, for PCM that is all there up front.  Every channel is one serial ADPCM chain whose state is
// carried from call to call (filefmt.c:214,312-314); the reference feeds it 28 samples (spu) or one chunk (spui) at a
// time, always a whole number of blocks except at the very end -- so a channel's blocks are exactly those of ONE encode
// over all its samples, and all channels go through one batched GPU call (psxhip_spu_encode_streams_host).  The host
// then only places the 16-byte blocks: leading dummy block, loop flags, trailing trap block, alignment padding, the
// big-endian .vag header.  "End of input" follows the reference's decoder: it becomes true in the iteration that
// consumes the last samples (decoding.c:510-534).
#include <cstring>
#include <vector>

#include "../../include/psxav_audio.h"
#include "../../include/psxav_hip.h"
#include "../../include/psxav_mdec.h"
#include "psxhip_internal.h"

namespace {

constexpr int kBlock = PSX_AUDIO_SPU_BLOCK_SIZE;                 // 16
constexpr int kSamples = PSX_AUDIO_SPU_SAMPLES_PER_BLOCK;        // 28
constexpr int kVagHeader = 0x30;                                 // filefmt.c:93

void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

bool interleaved(const psxhip_spu_file_settings_t* s) { return s->format == FORMAT_SPUI || s->format == FORMAT_VAGI; }

bool settings_ok(const psxhip_spu_file_settings_t* s) {
    if (!s) return false;
    if (s->format != FORMAT_SPU && s->format != FORMAT_VAG && s->format != FORMAT_SPUI && s->format != FORMAT_VAGI) return false;
    if (s->alignment < 1 || s->audio_frequency <= 0) return false;
    if (interleaved(s)) {
        if (s->audio_channels < 1 || s->audio_channels > 64) return false;
        if (s->audio_interleave < kBlock || (s->audio_interleave % kBlock)) return false;
    } else if (s->audio_channels != 1) {
        return false;                     // encode_file_spu reads the samples with pitch 1 (filefmt.c:243-249)
    }
    return true;
}

// write_vag_header, filefmt.c:95-162
void vag_header(const psxhip_spu_file_settings_t* s, int size_per_channel, uint8_t* h) {
    memset(h, 0, kVagHeader);
    h[0] = 'V'; h[1] = 'A'; h[2] = 'G';
    h[3] = s->format == FORMAT_VAGI ? 'i' : 'p';
    put_be32(h + 0x04, 0x20);                                     // version
    if (s->format == FORMAT_VAGI) {                               // interleave, little endian
        h[0x08] = (uint8_t)s->audio_interleave;
        h[0x09] = (uint8_t)(s->audio_interleave >> 8);
        h[0x0A] = (uint8_t)(s->audio_interleave >> 16);
        h[0x0B] = (uint8_t)(s->audio_interleave >> 24);
    }
    put_be32(h + 0x0C, (uint32_t)size_per_channel);
    put_be32(h + 0x10, (uint32_t)s->audio_frequency);
    if (s->format == FORMAT_VAGI && s->audio_loop_point >= 0) {   // loop point in bytes (non-standard)
        int loop_start_block = (s->audio_loop_point * s->audio_frequency) / (kSamples * 1000);
        if (!s->no_leading_dummy) loop_start_block++;
        put_be32(h + 0x14, (uint32_t)(loop_start_block * kBlock));
    }
    h[0x1E] = (uint8_t)s->audio_channels;
    strncpy((char*)(h + 0x20), s->name, 16);
}

// ---- spu / vag -----------------------------------------------------------------------------------------------
int64_t spu_layout(const psxhip_spu_file_settings_t* s, int64_t n, int64_t* data_blocks, int64_t* block_count) {
    const int64_t blocks = (n + kSamples - 1) / kSamples;
    int64_t count = (s->no_leading_dummy ? 0 : 1) + blocks + (s->enable_loop ? 0 : 1);
    *data_blocks = blocks;
    *block_count = count;
    int64_t bytes = count * kBlock;
    const int64_t overflow = bytes % s->alignment;
    if (overflow) bytes += s->alignment - overflow;
    // the reference seeks past the header, writes the data, then seeks back (filefmt.c:218-219,287-292)
    return (s->format == FORMAT_VAG ? kVagHeader : 0) + bytes;
}

// ---- spui / vagi ---------------------------------------------------------------------------------------------
struct SpuiPlan {
    int64_t samples_per_chunk, chunk_size, header_size, chunk_count;
};
SpuiPlan spui_plan(const psxhip_spu_file_settings_t* s, int64_t n) {
    SpuiPlan p;
    p.samples_per_chunk = (int64_t)s->audio_interleave / kBlock * kSamples;
    p.chunk_size = (int64_t)s->audio_interleave * s->audio_channels + s->alignment - 1;
    p.chunk_size -= p.chunk_size % s->alignment;
    p.header_size = kVagHeader + s->alignment - 1;
    p.header_size -= p.header_size % s->alignment;
    // chunk 0 consumes one block less when it starts with the dummy block (filefmt.c:331-335)
    int64_t left = n, chunks = 0;
    while (left > 0) {
        int64_t take = left < p.samples_per_chunk ? left : p.samples_per_chunk;
        if (chunks == 0 && !s->no_leading_dummy) take -= kSamples;
        if (take < 0) take = 0;     // fewer than 28 samples in total: the reference's samples_length goes negative and nothing is encoded
        left -= take;
        chunks++;
        if (take == 0 && chunks > 1) break;
    }
    p.chunk_count = chunks;
    return p;
}

}  // namespace

extern "C" int64_t psxhip_spu_file_size(const psxhip_spu_file_settings_t* s, int64_t samples_per_channel) {
    if (!settings_ok(s) || samples_per_channel < 0) {
        psxhip_set_error("psxhip_spu_file: bad settings");
        return PSXHIP_EINVAL;
    }
    if (!interleaved(s)) {
        int64_t a, b;
        return spu_layout(s, samples_per_channel, &a, &b);
    }
    const SpuiPlan p = spui_plan(s, samples_per_channel);
    return (s->format == FORMAT_VAGI ? p.header_size : 0) + p.chunk_count * p.chunk_size;
}

extern "C" int64_t psxhip_spu_file_encode_host(int device, const psxhip_spu_file_settings_t* s, const int16_t* pcm,
                                               int64_t samples_per_channel, uint8_t* out, size_t out_size) {
    const int64_t total = psxhip_spu_file_size(s, samples_per_channel);
    if (total < 0) return total;
    if (!out || (!pcm && samples_per_channel > 0) || (size_t)total > out_size) {
        psxhip_set_error("psxhip_spu_file_encode_host: output needs %lld bytes, %zu given", (long long)total, out_size);
        return PSXHIP_EINVAL;
    }
    if (samples_per_channel > 0x7FFFFFFF - kSamples) {
        psxhip_set_error("psxhip_spu_file_encode_host: stream too long");
        return PSXHIP_EINVAL;
    }
    memset(out, 0, (size_t)total);
    const int ch = s->audio_channels;
    const int64_t n = samples_per_channel;

    if (!interleaved(s)) {
        int64_t blocks, block_count;
        spu_layout(s, n, &blocks, &block_count);
        uint8_t* data = out + (s->format == FORMAT_VAG ? kVagHeader : 0);
        int64_t at = s->no_leading_dummy ? 0 : 1;                          // leading silent block, filefmt.c:224-230
        if (blocks) {
            psxhip_adpcm_state_t st = {0, 0};
            const int rc = psxhip_spu_encode_streams_host(device, pcm, 1, 0, 1, (int)n, &st, data + at * kBlock, blocks * kBlock);
            if (rc < 0) return rc;
        }
        if (s->audio_loop_point >= 0) {                                    // filefmt.c:234-235,251-252
            const int64_t loop_start_block = at + ((int64_t)s->audio_loop_point * s->audio_frequency) / (kSamples * 1000);
            if (loop_start_block >= at && loop_start_block < at + blocks) data[loop_start_block * kBlock + 1] |= PSX_AUDIO_SPU_LOOP_START;
        }
        if (s->enable_loop && blocks) data[(at + blocks - 1) * kBlock + 1] |= PSX_AUDIO_SPU_LOOP_REPEAT;     // filefmt.c:253-254
        if (!s->enable_loop) data[(at + blocks) * kBlock + 1] = PSX_AUDIO_SPU_LOOP_TRAP;                      // trailing looping block, :271-278
        if (s->format == FORMAT_VAG) vag_header(s, (int)(block_count * kBlock), out);
        return total;
    }

    // ---- interleaved: every channel's blocks from one batched call, then chunk placement (filefmt.c:323-371).
    //      Chunk c hands `take[c]` samples per channel to psx_audio_spu_encode (the first chunk one block less when it
    //      starts with the dummy block); a chunk's last block is zero-padded, so the chain's input is laid out planar
    //      with every chunk's segment padded to whole blocks -- normally that only pads the very end.
    const SpuiPlan p = spui_plan(s, n);
    uint8_t* data = out + (s->format == FORMAT_VAGI ? p.header_size : 0);
    const bool dummy = !s->no_leading_dummy;
    std::vector<int64_t> take((size_t)p.chunk_count), first_block((size_t)p.chunk_count);
    int64_t left = n, enc_blocks = 0;
    for (int64_t c = 0; c < p.chunk_count; c++) {
        int64_t t = left < p.samples_per_chunk ? left : p.samples_per_chunk;
        if (c == 0 && dummy) t -= kSamples;
        if (t < 0) t = 0;
        take[(size_t)c] = t;
        first_block[(size_t)c] = enc_blocks;
        enc_blocks += (t + kSamples - 1) / kSamples;
        left -= t;
    }
    std::vector<uint8_t> blocks((size_t)enc_blocks * kBlock * ch);
    if (enc_blocks) {
        const int64_t padded = enc_blocks * kSamples;
        std::vector<int16_t> planar((size_t)padded * ch, 0);
        int64_t src = 0;
        for (int64_t c = 0; c < p.chunk_count; c++) {
            int16_t* dst = planar.data() + first_block[(size_t)c] * kSamples;
            for (int64_t i = 0; i < take[(size_t)c]; i++)
                for (int k = 0; k < ch; k++) dst[(size_t)k * padded + i] = pcm[(size_t)(src + i) * ch + k];
            src += take[(size_t)c];
        }
        std::vector<psxhip_adpcm_state_t> st((size_t)ch);
        memset(st.data(), 0, st.size() * sizeof(st[0]));
        const int rc = psxhip_spu_encode_streams_host(device, planar.data(), ch, padded, 1, (int)padded, st.data(), blocks.data(),
                                                      enc_blocks * kBlock);
        if (rc < 0) return rc;
    }
    left = n;
    for (int64_t c = 0; c < p.chunk_count; c++) {
        uint8_t* ptr = data + c * p.chunk_size;
        const bool end_of_input = left <= p.samples_per_chunk;     // the decoder saw EOF while topping up its buffer (decoding.c:517-529)
        if (c == 0 && dummy) ptr += kBlock;                        // leading silent block, filefmt.c:331-335
        const int64_t nb = (take[(size_t)c] + kSamples - 1) / kSamples;
        for (int k = 0; k < ch; k++, ptr += s->audio_interleave) {
            if (nb <= 0) continue;
            memcpy(ptr, blocks.data() + ((size_t)k * enc_blocks + (size_t)first_block[(size_t)c]) * kBlock, (size_t)nb * kBlock);
            uint8_t* last = ptr + (nb - 1) * kBlock;
            if (s->enable_loop || (end_of_input && s->audio_loop_point >= 0)) {
                last[1] = PSX_AUDIO_SPU_LOOP_REPEAT;
            } else if (end_of_input) {
                memset(last, 0, kBlock);                              // the reference repurposes the last block, filefmt.c:355-361
                last[1] = PSX_AUDIO_SPU_LOOP_TRAP;
            }
        }
        left -= take[(size_t)c];
    }
    if (s->format == FORMAT_VAGI) vag_header(s, (int)(p.chunk_count * s->audio_interleave), out);
    return total;
}
