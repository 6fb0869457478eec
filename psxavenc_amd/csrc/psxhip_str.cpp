// psxhip_str.cpp -- batched STR / STRCD / STRV muxer (include/psxav_hip.h, psxhip_str_*).
//
// What psxavenc's encode_file_str does sector by sector (psxavenc/filefmt.c:391-520 with :73-91, around
// encode_sector_str, mdec.c:757-836, and psx_audio_xa_encode, adpcm.c:293-332), restated for inputs that are all
// there up front: the per-frame byte budgets are a closed-form function of the frame index (mdec.c:768-775), so
// every frame is encoded in ONE batched MDEC launch; the audio is one XA stream encoded by the ADPCM kernels
// concurrently (its own host thread and stream); the host then interleaves 2016-byte slices of the finished frames
// with the finished audio sectors following the reference's sector schedule.  No encoding happens on the host.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/psxav_audio.h"
#include "../../include/psxav_hip.h"
#include "../../include/psxav_mdec.h"
#include "psxhip_internal.h"

namespace {

// One MDEC context is kept between calls (creating one allocates pinned staging buffers, which costs more than
// encoding a thousand frames); psxhip_str_release() drops it.
std::mutex g_ctx_mu;
psxhip_mdec_ctx_t* g_ctx = nullptr;
int g_ctx_key[5] = {-1, -1, -1, -1, -1};      // device, codec, width, height, max_frame_size
// ... and so is the buffer the frames' bitstreams land in before they are cut into sectors: page-locked, so the MDEC host
// path writes it by DMA; calls are serialised on it
std::mutex g_call_mu;
uint8_t* g_bs = nullptr;
size_t g_bs_cap = 0;

void put_le16(uint8_t* p, unsigned v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
void put_le32(uint8_t* p, unsigned v) { put_le16(p, v & 0xFFFF); put_le16(p + 2, v >> 16); }

psx_audio_xa_settings_t xa_settings_of(const psxhip_str_settings_t* s) {
    psx_audio_xa_settings_t x;          // args_to_libpsxav_xa_audio, filefmt.c:55-71
    memset(&x, 0, sizeof x);
    x.bits_per_sample = s->audio_bit_depth;
    x.frequency = s->audio_frequency;
    x.stereo = s->audio_channels == 2;
    x.file_number = s->audio_xa_file;
    x.channel_number = s->audio_xa_channel;
    x.format = s->format == FORMAT_STRCD ? PSX_AUDIO_XA_FORMAT_XACD : PSX_AUDIO_XA_FORMAT_XA;
    return x;
}

bool settings_ok(const psxhip_str_settings_t* s) {
    if (!s) return false;
    if (s->format != FORMAT_STR && s->format != FORMAT_STRCD && s->format != FORMAT_STRV) return false;
    if (s->video_codec < 0 || s->video_codec > 2 || s->video_width <= 0 || s->video_height <= 0 ||
        (s->video_width % 16) || (s->video_height % 16))
        return false;
    if (s->str_fps_num <= 0 || s->str_fps_den <= 0 || (s->str_cd_speed != 1 && s->str_cd_speed != 2)) return false;
    if (s->audio_channels < 0 || s->audio_channels > 2) return false;
    if (s->audio_channels && ((s->audio_frequency != 18900 && s->audio_frequency != 37800) ||
                              (s->audio_bit_depth != 4 && s->audio_bit_depth != 8)))
        return false;
    return true;
}

// sector schedule, filefmt.c:454-461
bool is_video_sector(const psxhip_str_settings_t* s, int interleave, int video_per_block, int sector) {
    if (!s->audio_channels) return true;
    if (s->trailing_audio) return (sector % interleave) < video_per_block;
    return (sector % interleave) > 0;
}

struct Plan {
    psxhip_str_plan_t pub;
    int base, den;                // frame_block_base_overflow / frame_block_overflow_den, filefmt.c:431-432
    int video_per_block;
    std::vector<int32_t> budgets; // frame_max_size of every frame, mdec.c:768-775
};

int make_plan(const psxhip_str_settings_t* s, int n_frames, Plan* pl) {
    if (!settings_ok(s) || n_frames < 0) {
        psxhip_set_error("psxhip_str: bad settings");
        return PSXHIP_EINVAL;
    }
    memset(&pl->pub, 0, sizeof pl->pub);
    const psx_audio_xa_settings_t xa = xa_settings_of(s);
    int interleave = 1, sps = 0, vpb = 1;
    if (s->audio_channels) {              // 1/N audio, (N-1)/N video
        interleave = (int)psx_audio_xa_get_sector_interleave(xa) * s->str_cd_speed;
        sps = (int)psx_audio_xa_get_samples_per_sector(xa);
        vpb = interleave - 1;
    }
    pl->base = 75 * s->str_cd_speed * vpb * s->str_fps_den;
    pl->den = interleave * s->str_fps_num;
    pl->video_per_block = vpb;
    pl->budgets.resize((size_t)n_frames);
    int num = 0, max_budget = 0;
    long long video_sectors = 0;
    for (int i = 0; i < n_frames; i++) {
        num += pl->base;
        const int size = num / pl->den * 2016;
        num %= pl->den;
        if (size < 2016) {
            psxhip_set_error("psxhip_str: frame %d would get no sector (frame rate too high for this CD speed)", i);
            return PSXHIP_EINVAL;
        }
        pl->budgets[(size_t)i] = size;
        if (size > max_budget) max_budget = size;
        video_sectors += size / 2016;
    }
    // the stream ends with the last frame's last sector
    long long n = 0, v = 0;
    while (v < video_sectors) {
        if (is_video_sector(s, interleave, vpb, (int)n)) v++;
        n++;
        if (n > 0x7FFFFFF0ll) {
            psxhip_set_error("psxhip_str: stream too long");
            return PSXHIP_EINVAL;
        }
    }
    pl->pub.n_sectors = (int32_t)n;
    pl->pub.n_video_sectors = (int32_t)video_sectors;
    pl->pub.n_audio_sectors = (int32_t)(n - video_sectors);
    pl->pub.sector_size = (int32_t)psx_audio_xa_get_buffer_size_per_sector(xa);
    pl->pub.interleave = interleave;
    pl->pub.audio_samples_per_sector = sps;
    pl->pub.max_frame_size = max_budget;
    return PSXHIP_OK;
}

}  // namespace

extern "C" void psxhip_str_release(void) {
    std::lock_guard<std::mutex> call(g_call_mu);
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    psxhip_mdec_destroy(g_ctx);
    g_ctx = nullptr;
    g_ctx_key[0] = -1;
    if (g_bs) (void)hipHostFree(g_bs);
    g_bs = nullptr;
    g_bs_cap = 0;
}

extern "C" int psxhip_str_plan(const psxhip_str_settings_t* settings, int n_frames, psxhip_str_plan_t* plan) {
    Plan pl;
    const int rc = make_plan(settings, n_frames, &pl);
    if (plan) *plan = pl.pub;
    return rc;
}

extern "C" int psxhip_str_frame_budgets(const psxhip_str_settings_t* settings, int first_frame, int n_frames, int32_t* budgets) {
    if (!budgets || first_frame < 0 || n_frames < 0) return PSXHIP_EINVAL;
    Plan pl;
    const int rc = make_plan(settings, first_frame + n_frames, &pl);
    if (rc) return rc;
    for (int i = 0; i < n_frames; i++) budgets[i] = pl.budgets[(size_t)(first_frame + i)];
    return PSXHIP_OK;
}

extern "C" int psxhip_str_encode_host(int device, const psxhip_str_settings_t* s, const uint8_t* frames, int n_frames,
                                      const int16_t* pcm, int64_t pcm_samples_per_channel, uint8_t* out, size_t out_size,
                                      psxhip_str_plan_t* plan_out) {
    Plan pl;
    int rc = make_plan(s, n_frames, &pl);
    if (plan_out) *plan_out = pl.pub;
    if (rc) return rc;
    if (n_frames == 0) return PSXHIP_OK;
    if (!frames || !out || (s->audio_channels && !pcm && pcm_samples_per_channel > 0) || pcm_samples_per_channel < 0) {
        psxhip_set_error("psxhip_str_encode_host: NULL argument");
        return PSXHIP_EINVAL;
    }
    const size_t ssz = (size_t)pl.pub.sector_size;
    if (out_size < ssz * (size_t)pl.pub.n_sectors) {
        psxhip_set_error("psxhip_str_encode_host: output needs %zu bytes, %zu given", ssz * (size_t)pl.pub.n_sectors, out_size);
        return PSXHIP_EINVAL;
    }

    // ---- video: every frame in one batched call (its own streams inside the context)
    std::lock_guard<std::mutex> call(g_call_mu);
    const size_t ostride = (size_t)pl.pub.max_frame_size;
    if ((size_t)n_frames * ostride > g_bs_cap) {
        if (hipSetDevice(device) != hipSuccess) { psxhip_set_error("psxhip_str_encode_host: no such device %d", device); return PSXHIP_EDEVICE; }
        if (g_bs) (void)hipHostFree(g_bs);
        g_bs = nullptr;
        g_bs_cap = 0;
        if (hipHostMalloc((void**)&g_bs, (size_t)n_frames * ostride, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            psxhip_set_error("psxhip_str_encode_host: out of pinned host memory (%zu bytes)", (size_t)n_frames * ostride);
            return PSXHIP_ENOMEM;
        }
        g_bs_cap = (size_t)n_frames * ostride;
    }
    uint8_t* const bs = g_bs;
    std::vector<psxhip_mdec_result_t> res((size_t)n_frames);
    int rc_video = PSXHIP_OK;
    char err_video[256] = "";
    std::vector<uint8_t> xa_out;
    // ---- interleave (filefmt.c:450-503): which frame slice / audio sector lands in which sector is known from the plan alone
    const int at = s->format == FORMAT_STR ? 0x08 : (s->format == FORMAT_STRCD ? 0x18 : 0x00);     // mdec.c:822-829
    // which frame slice / audio sector lands in which sector: a short serial walk; the sectors themselves (2 KiB of copying
    // and a 2 KiB EDC each) are then built by a few threads, each on its own range
    const int ns = pl.pub.n_sectors;
    std::vector<int32_t> sec_frame((size_t)ns), sec_off((size_t)ns);        // video: frame, byte offset; audio: -1, audio sector
    int frames_in_stream = 0;
    {
        int frame = -1, offset = 0, budget = 0, audio_sector = 0;
        for (int n = 0; n < ns; n++) {
            if (is_video_sector(s, pl.pub.interleave, pl.video_per_block, n)) {
                if (offset >= budget) {                    // encode_sector_str moves on to the next frame, mdec.c:766-779
                    frame++;
                    budget = pl.budgets[(size_t)frame];
                    offset = 0;
                }
                sec_frame[(size_t)n] = frame;
                sec_off[(size_t)n] = offset;
                offset += 2016;
            } else {
                sec_frame[(size_t)n] = -1;
                sec_off[(size_t)n] = audio_sector++;
            }
        }
        frames_in_stream = frame + 1;
    }
    // `video`: build the video sectors of [n0, n1), else its audio sectors -- the two kinds are built by different threads at
    // different times (whichever of the frame encode and the XA encode finishes first has its sectors cut while the other runs)
    auto build = [&](int n0, int n1, bool video) {
        uint8_t sector[PSX_CDROM_SECTOR_SIZE];
        for (int n = n0; n < n1; n++) {
            uint8_t* dst = out + (size_t)n * ssz;
            const int frame = sec_frame[(size_t)n], offset = sec_off[(size_t)n];
            if ((frame >= 0) != video) continue;
            if (frame >= 0) {
                const int budget = pl.budgets[(size_t)frame];
                memset(sector, 0, sizeof sector);            // the reference's buffer is an uninitialised stack array
                // init_sector_buffer_video, filefmt.c:73-91
                psx_cdrom_sector_xa_subheader_t* sub = nullptr;
                if (s->format == FORMAT_STRCD) {
                    psx_cdrom_init_sector((psx_cdrom_sector_t*)sector, n, PSX_CDROM_SECTOR_TYPE_MODE2_FORM1);
                    sub = ((psx_cdrom_sector_t*)sector)->mode2.subheader;
                } else if (s->format == FORMAT_STR) {
                    sub = (psx_cdrom_sector_xa_subheader_t*)sector;
                }
                if (sub) {
                    sub->file = (uint8_t)s->audio_xa_file;
                    sub->channel = (uint8_t)(s->audio_xa_channel & PSX_CDROM_SECTOR_XA_CHANNEL_MASK);
                    sub->submode = PSX_CDROM_SECTOR_XA_SUBMODE_DATA | PSX_CDROM_SECTOR_XA_SUBMODE_RT;
                    sub->coding = 0;
                    sub[1] = sub[0];
                }
                // the 32-byte chunk header + 2016 payload bytes of encode_sector_str, mdec.c:782-832
                const uint8_t* fo = bs + (size_t)frame * ostride;
                uint8_t* hd = sector + at;
                put_le16(hd + 0x00, 0x0160);
                put_le16(hd + 0x02, (unsigned)s->str_video_id);
                put_le16(hd + 0x04, (unsigned)(offset / 2016));
                put_le16(hd + 0x06, (unsigned)(budget / 2016));
                put_le32(hd + 0x08, (unsigned)(frame + 1));                         // frame_index counts from 1
                put_le32(hd + 0x0C, (unsigned)res[(size_t)frame].bytes_used);
                put_le16(hd + 0x10, (unsigned)s->video_width);
                put_le16(hd + 0x12, (unsigned)s->video_height);
                memcpy(hd + 0x14, fo, 8);
                put_le32(hd + 0x1C, 0);
                memcpy(hd + 0x20, fo + offset, 2016);
                psx_cdrom_calculate_checksums((psx_cdrom_sector_t*)sector, PSX_CDROM_SECTOR_TYPE_MODE2_FORM1);
                memcpy(dst, sector, ssz);
            } else {
                memcpy(dst, xa_out.data() + (size_t)offset * ssz, ssz);
                if (s->format == FORMAT_STRCD) {
                    // the audio sectors were assembled with consecutive addresses; the header carries this sector's own LBA
                    // (psx_cdrom_init_sector, cdrom.c:55-74).  The form-2 EDC does not cover the header.
                    psx_cdrom_sector_t tmp;
                    psx_cdrom_init_sector(&tmp, n, PSX_CDROM_SECTOR_TYPE_MODE2_FORM2);
                    memcpy(dst + 12, (const uint8_t*)&tmp + 12, 3);
                }
            }
        }
    };
    auto build_all = [&](bool video) {
        const unsigned hw = std::thread::hardware_concurrency();
        int nt = ns >= 2048 ? (hw >= 64 ? 16 : (hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 1))) : 1;
        if (!video && nt > 4) nt = 4;                       // one sector in eight, and no checksum to compute
        std::vector<std::thread> th;
        const int per = (ns + nt - 1) / nt;
        for (int t = 1; t < nt; t++)
            if (t * per < ns) th.emplace_back(build, t * per, (t + 1) * per < ns ? (t + 1) * per : ns, video);
        build(0, per < ns ? per : ns, video);
        for (auto& x : th) x.join();
    };

    std::thread video([&]() {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        const int key[5] = {device, s->video_codec, s->video_width, s->video_height, pl.pub.max_frame_size};
        if (!g_ctx || memcmp(key, g_ctx_key, sizeof key) != 0) {
            psxhip_mdec_destroy(g_ctx);
            g_ctx = nullptr;
            rc_video = psxhip_mdec_create(&g_ctx, device, s->video_codec, s->video_width, s->video_height, pl.pub.max_frame_size);
            if (rc_video == PSXHIP_OK) memcpy(g_ctx_key, key, sizeof key);
        }
        if (rc_video == PSXHIP_OK)
            rc_video = psxhip_mdec_encode_frames_host(g_ctx, frames, n_frames, pl.budgets.data(), 0, bs, ostride, res.data());
        if (rc_video) snprintf(err_video, sizeof err_video, "%s", psxhip_last_error());     // thread-local text
        else build_all(true);
    });

    // ---- audio: one XA stream, concurrently
    const int na = pl.pub.n_audio_sectors, sps = pl.pub.audio_samples_per_sector, ch = s->audio_channels;
    int rc_audio = PSXHIP_OK;
    if (na > 0) {
        // the reference relies on zero padding after the end of the PCM data (decoding.c:497-503); a stream whose audio
        // is shorter than its video gets silence here (the reference writes an uninitialised sector, filefmt.c:476-490)
        const int64_t need = (int64_t)na * sps;
        std::vector<int16_t> padded;
        const int16_t* src = pcm;
        if (pcm_samples_per_channel < need) {
            padded.assign((size_t)need * ch, 0);
            if (pcm_samples_per_channel) memcpy(padded.data(), pcm, (size_t)pcm_samples_per_channel * ch * sizeof(int16_t));
            src = padded.data();
        }
        xa_out.resize((size_t)na * ssz);
        psxhip_adpcm_state_t st[2] = {{0, 0}, {0, 0}};
        const int32_t lba0 = 0;
        const int fmt = s->format == FORMAT_STRCD ? 1 : 0;
        rc_audio = psxhip_xa_encode_streams_host(device, fmt, ch == 2, s->audio_frequency, s->audio_bit_depth,
                                                 s->audio_xa_file, s->audio_xa_channel, src, 1, need * ch, (int)need, &lba0, st,
                                                 xa_out.data(), (int64_t)xa_out.size(), 1);
        if (rc_audio > 0) rc_audio = PSXHIP_OK;
    }
    if (rc_audio == PSXHIP_OK && na > 0) build_all(false);
    video.join();
    if (rc_video) {
        psxhip_set_error("psxhip_str_encode_host: video: %s", err_video);
        return rc_video;
    }
    if (rc_audio) return rc_audio;

    long long qsum = 0;
    for (int f = 0; f < frames_in_stream; f++) qsum += res[(size_t)f].quant_scale;
    pl.pub.quant_scale_sum = qsum;
    if (plan_out) *plan_out = pl.pub;
    return PSXHIP_OK;
}
