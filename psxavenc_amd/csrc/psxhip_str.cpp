// psxhip_str.cpp -- batched STR / STRCD / STRV muxer (include/psxav_hip.h, psxhip_str_*).
//
// What psxavenc's encode_file_str does sector by sector (psxavenc/filefmt.c:391-520 with :73-91, around
// encode_sector_str, mdec.c:757-836, and psx_audio_xa_encode, adpcm.c:293-332), restated for inputs that are all
// there up front: the sector loop is first run dry (make_plan: which frame slice / audio sector lands where, which
// frames are part of the stream at all, which audio sectors carry EOF -- incl. the reference's end-of-input model);
// the per-frame byte budgets are a closed-form function of the frame index (mdec.c:768-775), so every frame is encoded
// in ONE batched MDEC call sharded over the handle's devices; the audio is one XA stream encoded by the ADPCM kernels
// concurrently (its own host thread and stream); the host then cuts the finished frames into sectors and interleaves
// them with the finished audio sectors.  No encoding happens on the host.  No process-global state: everything kept
// between calls lives in a psxhip_str_ctx_t.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/psxav_audio.h"
#include "../../include/psxav_hip.h"
#include "../../include/psxav_mdec.h"
#include "psxhip_internal.h"

namespace {
struct Sector {
    int32_t frame;      // >= 0: video sector of that frame; -1: audio sector; -2: audio slot with no samples left
    int32_t at;         // video: byte offset into the frame's bitstream; audio: index of the XA sector
    uint8_t eof;        // audio: EOF submode bit (psx_audio_xa_encode_finalize)
};

struct Plan {
    psxhip_str_plan_t pub;
    int base, den;                // frame_block_base_overflow / frame_block_overflow_den, filefmt.c:431-432
    std::vector<int32_t> budgets; // frame_max_size of every frame in the stream, mdec.c:768-775
    std::vector<Sector> sectors;
    int64_t audio_samples;        // per channel, handed to the XA encoder over the whole stream
};
}  // namespace

struct psxhip_str_ctx {
    // One multi-device MDEC encoder is kept between calls (creating one allocates pinned staging buffers per device, which
    // costs more than encoding a thousand frames), re-created when the geometry changes
    std::vector<int> devices;
    psxhip_mdec_multi_t* mdec = nullptr;
    int key[4] = {-1, -1, -1, -1};          // codec, width, height, max_frame_size
    // ... and so is the buffer the frames' bitstreams land in before they are cut into sectors: page-locked, so the MDEC
    // host path writes it by DMA
    uint8_t* bs = nullptr;
    size_t bs_cap = 0;
    std::mutex mu;                          // calls on ONE handle are serialised; handles are independent of each other
    // ---- psxhip_str_encode_device: everything stays in HBM (device devices[0]).  Kept between calls with the same shape: the
    //      plan's tables on the device, the buffers the frames' bitstreams / unit records pass through, one encoder context, one
    //      ADPCM session over the caller's PCM, a stream + events for the audio leg
    struct Dev {
        psxhip_str_settings_t settings;
        int n_frames = -1, n_streams = 0;
        int64_t pcm_samples = -1;
        const int16_t* d_pcm = nullptr;
        int64_t pcm_stream_stride = 0;
        bool chunked = false;           // the XA tracks of the cached shape run as a speculate-and-verify session (else: serial chains)
        Plan plan;
        int n_vtab = 0, na = 0, nf = 0;
        void *d_vtab = nullptr, *d_budgets = nullptr, *d_adst = nullptr, *d_eof = nullptr, *d_bs = nullptr, *d_res = nullptr, *d_units = nullptr;
        void *d_chains = nullptr, *d_base = nullptr, *d_states = nullptr;      // short audio: the serial chains kernel's tables
        psxhip_mdec_result_t* h_res = nullptr;                                 // page-locked
        psxhip_mdec_ctx_t* mdec = nullptr;
        int mdec_key[4] = {-1, -1, -1, -1};
        psxhip_adpcm_session_t* session = nullptr;
        hipStream_t astream = nullptr;
        hipEvent_t ev_in = nullptr, ev_audio = nullptr;
    } dev;
};

namespace {

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
    if (s->tail_mode != PSXHIP_STR_TAIL_REFERENCE && s->tail_mode != PSXHIP_STR_TAIL_COMPLETE) return false;
    if (s->audio_channels && ((s->audio_frequency != 18900 && s->audio_frequency != 37800) ||
                              (s->audio_bit_depth != 4 && s->audio_bit_depth != 8)))
        return false;
    return true;
}

// The sector loop of encode_file_str (filefmt.c:450-503) run dry: which frame slice / audio sector lands in which sector
// follows from the frame count, the amount of audio and the settings alone.
//
// tail_mode PSXHIP_STR_TAIL_REFERENCE models the reference's decoder (decoding.c:510-560) for an input that is all there:
// ensure_av_data(needed_audio, frames_needed) raises end_of_input as soon as no more than one sector's worth of audio or no
// more than `frames_needed` frames are left to hand out (its loop polls while count <= needed, and the only way out with
// nothing left to read is end_of_input = true).  From then on the loop runs until the current frame is written out
// (filefmt.c:450) -- the last frames_needed frames are never encoded (the FIXME at :442) -- every audio sector is finalised
// (:492-493), and an audio slot with no samples left stays as the sector buffer was (zero here) and widens the video share of
// the trailing-audio schedule (:483-484).
// PSXHIP_STR_TAIL_COMPLETE: every frame is encoded, the stream ends with the last frame's last sector, short audio is padded
// with silence and only the last audio sector carries EOF.
int make_plan(const psxhip_str_settings_t* s, int n_frames, int64_t pcm_samples_per_channel, Plan* pl) {
    if (!settings_ok(s) || n_frames < 0 || pcm_samples_per_channel < 0) {
        psxhip_set_error("psxhip_str: bad settings");
        return PSXHIP_EINVAL;
    }
    memset(&pl->pub, 0, sizeof pl->pub);
    pl->budgets.clear();
    pl->sectors.clear();
    pl->audio_samples = 0;
    const psx_audio_xa_settings_t xa = xa_settings_of(s);
    const int ch = s->audio_channels;
    int interleave = 1, sps = 0, vpb = 1;
    if (ch) {                             // 1/N audio, (N-1)/N video, filefmt.c:399-403
        interleave = (int)psx_audio_xa_get_sector_interleave(xa) * s->str_cd_speed;
        sps = (int)psx_audio_xa_get_samples_per_sector(xa);
        vpb = interleave - 1;
    }
    pl->base = 75 * s->str_cd_speed * vpb * s->str_fps_den;
    pl->den = interleave * s->str_fps_num;
    pl->pub.sector_size = (int32_t)psx_audio_xa_get_buffer_size_per_sector(xa);
    pl->pub.interleave = interleave;
    pl->pub.audio_samples_per_sector = sps;
    if (pl->base / pl->den < 1) {
        psxhip_set_error("psxhip_str: a frame would get no sector (frame rate too high for this CD speed)");
        return PSXHIP_EINVAL;
    }
    // filefmt.c:443-446
    const double frame_size = (double)pl->base / (double)pl->den;
    int frames_needed = (int)ceil((double)vpb / frame_size);
    if (frames_needed < 2) frames_needed = 2;
    const bool reference = s->tail_mode == PSXHIP_STR_TAIL_REFERENCE;

    long long V = n_frames;                                    // frames the decoder still holds
    long long A = ch ? pcm_samples_per_channel * ch : 0;       // interleaved samples the decoder still holds
    bool eoi = false;
    int offset = 0, max_size = 0, num = 0, frame = -1, audio_sectors = 0, video_sectors = 0, max_budget = 0;
    // complete mode: the audio slots of the whole stream are filled (silence when the PCM runs out)
    for (long long n = 0;; n++) {
        if (reference) {
            if (eoi && offset >= max_size) break;              // loop condition, filefmt.c:450
            const long long needed_audio = (long long)sps * ch;
            if ((needed_audio && A <= needed_audio) || V <= frames_needed) eoi = true;     // ensure_av_data, decoding.c:540-553
        } else if (frame + 1 >= n_frames && offset >= max_size) {
            break;
        }
        if (n > 0x7FFFFFF0ll) {
            psxhip_set_error("psxhip_str: stream too long");
            return PSXHIP_EINVAL;
        }
        bool video;                                            // filefmt.c:454-461
        if (!sps) video = true;
        else if (s->trailing_audio) video = (n % interleave) < vpb;
        else video = (n % interleave) > 0;
        Sector sec;
        if (video) {
            // a video slot with no frame left to start (n_frames == 0: the reference asserts in its decoder's retire_av_data, there
            // is nothing to mirror): the stream ends here -- empty, or the audio sectors before this slot
            if (offset >= max_size && V <= 0) break;
            while (offset >= max_size) {                       // encode_sector_str moves on to the next frame, mdec.c:768-780
                frame++;
                num += pl->base;
                max_size = num / pl->den * 2016;
                num %= pl->den;
                offset = 0;
                pl->budgets.push_back(max_size);
                if (max_size > max_budget) max_budget = max_size;
                V--;
            }
            sec.frame = frame;
            sec.at = offset;
            sec.eof = 0;
            offset += 2016;
            video_sectors++;
        } else if (reference) {
            long long sl = A / ch;                             // filefmt.c:476-484
            if (sl > sps) sl = sps;
            if (!sl) vpb++;
            sec.frame = sl ? -1 : -2;
            sec.at = sl ? audio_sectors : 0;
            sec.eof = (eoi && sl) ? 1 : 0;                     // :492-493 (finalize does nothing to a sector of length 0)
            if (sl) {
                audio_sectors++;
                pl->audio_samples += sl;
                A -= sl * ch;
            }
        } else {
            sec.frame = -1;
            sec.at = audio_sectors++;
            sec.eof = 0;
            pl->audio_samples += sps;
        }
        pl->sectors.push_back(sec);
    }
    if (!reference && audio_sectors > 0)                       // only the last audio sector carries EOF
        for (size_t i = pl->sectors.size(); i-- > 0;)
            if (pl->sectors[i].frame == -1) { pl->sectors[i].eof = 1; break; }
    pl->pub.n_sectors = (int32_t)pl->sectors.size();
    pl->pub.n_video_sectors = video_sectors;
    pl->pub.n_audio_sectors = (int32_t)pl->sectors.size() - video_sectors;
    pl->pub.n_frames_encoded = frame + 1;
    pl->pub.max_frame_size = max_budget;
    return PSXHIP_OK;
}

}  // namespace

namespace {
void free_dev(psxhip_str_ctx* c) {
    psxhip_str_ctx::Dev& d = c->dev;
    (void)hipSetDevice(c->devices[0]);
    if (d.astream) (void)hipStreamSynchronize(d.astream);
    if (d.session) psxhip_adpcm_session_destroy(d.session);
    if (d.mdec) psxhip_mdec_destroy(d.mdec);
    void** bufs[] = {&d.d_vtab, &d.d_budgets, &d.d_adst, &d.d_eof, &d.d_bs, &d.d_res, &d.d_units, &d.d_chains, &d.d_base, &d.d_states};
    for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
    if (d.h_res) (void)hipHostFree(d.h_res);
    if (d.ev_in) (void)hipEventDestroy(d.ev_in);
    if (d.ev_audio) (void)hipEventDestroy(d.ev_audio);
    if (d.astream) (void)hipStreamDestroy(d.astream);
    d.h_res = nullptr; d.ev_in = nullptr; d.ev_audio = nullptr; d.astream = nullptr; d.session = nullptr; d.mdec = nullptr;
    d.n_frames = -1;
}
}  // namespace

extern "C" int psxhip_str_create(psxhip_str_ctx_t** out, const int* devices, int n_devices) {
    if (!out) return PSXHIP_EINVAL;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) {
        psxhip_set_error("psxhip_str_create: need 1..64 devices");
        return PSXHIP_EINVAL;
    }
    const int have = psxhip_device_count();
    if (have <= 0) {
        psxhip_set_error("no HIP device visible (libpsxav_hip has no CPU fallback)");
        return PSXHIP_EDEVICE;
    }
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= have) {
            psxhip_set_error("psxhip_str_create: device %d out of range (%d visible)", devices[i], have);
            return PSXHIP_EINVAL;
        }
    psxhip_str_ctx* c = new (std::nothrow) psxhip_str_ctx;
    if (!c) return PSXHIP_ENOMEM;
    c->devices.assign(devices, devices + n_devices);
    *out = c;
    return PSXHIP_OK;
}

extern "C" void psxhip_str_destroy(psxhip_str_ctx_t* c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        psxhip_mdec_multi_destroy(c->mdec);
        c->mdec = nullptr;
        if (c->bs) {
            (void)hipSetDevice(c->devices[0]);
            (void)hipHostFree(c->bs);
        }
        c->bs = nullptr;
        free_dev(c);
    }
    delete c;
}

extern "C" int psxhip_str_plan(const psxhip_str_settings_t* settings, int n_frames, int64_t pcm_samples_per_channel,
                               psxhip_str_plan_t* plan) {
    Plan pl;
    const int rc = make_plan(settings, n_frames, pcm_samples_per_channel, &pl);
    if (plan) *plan = pl.pub;
    return rc;
}

extern "C" int psxhip_str_plan_sectors(const psxhip_str_settings_t* settings, int n_frames, int64_t pcm_samples_per_channel,
                                       psxhip_str_sector_t* sectors, int cap) {
    Plan pl;
    const int rc = make_plan(settings, n_frames, pcm_samples_per_channel, &pl);
    if (rc) return rc;
    const int n = (int)pl.sectors.size();
    for (int i = 0; i < n && i < cap && sectors; i++) {
        const Sector& sc = pl.sectors[(size_t)i];
        sectors[i].kind = sc.frame >= 0 ? PSXHIP_STR_SECTOR_VIDEO : (sc.frame == -1 ? PSXHIP_STR_SECTOR_AUDIO : PSXHIP_STR_SECTOR_EMPTY);
        sectors[i].frame = sc.frame >= 0 ? sc.frame : -1;
        sectors[i].index = sc.frame >= 0 ? sc.at / 2016 : (sc.frame == -1 ? sc.at : -1);
        sectors[i].eof = sc.eof;
    }
    return n;
}

extern "C" int psxhip_str_frame_budgets(const psxhip_str_settings_t* settings, int first_frame, int n_frames, int32_t* budgets) {
    if (!budgets || first_frame < 0 || n_frames < 0) return PSXHIP_EINVAL;
    // the budget sequence does not depend on how the stream ends: plan all frames in the complete mode (mdec.c:768-775)
    psxhip_str_settings_t s2;
    if (!settings) return PSXHIP_EINVAL;
    s2 = *settings;
    s2.tail_mode = PSXHIP_STR_TAIL_COMPLETE;
    Plan pl;
    const int rc = make_plan(&s2, first_frame + n_frames, 0, &pl);
    if (rc) return rc;
    for (int i = 0; i < n_frames; i++) budgets[i] = pl.budgets[(size_t)(first_frame + i)];
    return PSXHIP_OK;
}

extern "C" int psxhip_str_encode_host(psxhip_str_ctx_t* c, const psxhip_str_settings_t* s, const uint8_t* frames, int n_frames,
                                      const int16_t* pcm, int64_t pcm_samples_per_channel, uint8_t* out, size_t out_size,
                                      psxhip_str_plan_t* plan_out) {
    if (!c) {
        psxhip_set_error("psxhip_str_encode_host: NULL handle");
        return PSXHIP_EINVAL;
    }
    Plan pl;
    int rc = make_plan(s, n_frames, pcm_samples_per_channel, &pl);
    if (plan_out) *plan_out = pl.pub;
    if (rc) return rc;
    const int ns = pl.pub.n_sectors;
    if (ns == 0) return PSXHIP_OK;
    const int nf = pl.pub.n_frames_encoded;
    if (nf > n_frames) {      // (a plan never takes a frame the caller did not hand over)
        psxhip_set_error("psxhip_str_encode_host: internal error: plan encodes %d of %d frames", nf, n_frames);
        return PSXHIP_EINVAL;
    }
    if ((nf && !frames) || !out || (s->audio_channels && !pcm && pcm_samples_per_channel > 0)) {
        psxhip_set_error("psxhip_str_encode_host: NULL argument");
        return PSXHIP_EINVAL;
    }
    const size_t ssz = (size_t)pl.pub.sector_size;
    if (out_size < ssz * (size_t)ns) {
        psxhip_set_error("psxhip_str_encode_host: output needs %zu bytes, %zu given", ssz * (size_t)ns, out_size);
        return PSXHIP_EINVAL;
    }

    // ---- video: every frame of the stream in one batched call, sharded over the handle's devices
    std::lock_guard<std::mutex> call(c->mu);
    const int device = c->devices[0];
    const size_t ostride = (size_t)pl.pub.max_frame_size;
    if ((size_t)nf * ostride > c->bs_cap) {
        if (hipSetDevice(device) != hipSuccess) { psxhip_set_error("psxhip_str_encode_host: no such device %d", device); return PSXHIP_EDEVICE; }
        if (c->bs) (void)hipHostFree(c->bs);
        c->bs = nullptr;
        c->bs_cap = 0;
        // page-locked for every device of the list (the default flags of a multi-GPU process map it for all of them)
        if (hipHostMalloc((void**)&c->bs, (size_t)nf * ostride, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            psxhip_set_error("psxhip_str_encode_host: out of pinned host memory (%zu bytes)", (size_t)nf * ostride);
            return PSXHIP_ENOMEM;
        }
        c->bs_cap = (size_t)nf * ostride;
    }
    uint8_t* const bs = c->bs;
    std::vector<psxhip_mdec_result_t> res((size_t)(nf ? nf : 1));
    int rc_video = PSXHIP_OK;
    char err_video[256] = "";
    std::vector<uint8_t> xa_out;
    const int at = s->format == FORMAT_STR ? 0x08 : (s->format == FORMAT_STRCD ? 0x18 : 0x00);     // mdec.c:822-829
    // Which frame slice / audio sector lands in which sector is in the plan; the sectors themselves (2 KiB of copying and a
    // 2 KiB EDC each) are built by a few threads, each on its own range.
    // `video`: build the video sectors of [n0, n1), else its audio sectors -- the two kinds are built by different threads at
    // different times (whichever of the frame encode and the XA encode finishes first has its sectors cut while the other runs)
    auto build = [&](int n0, int n1, bool video) {
        uint8_t sector[PSX_CDROM_SECTOR_SIZE];
        for (int n = n0; n < n1; n++) {
            uint8_t* dst = out + (size_t)n * ssz;
            const Sector& sc = pl.sectors[(size_t)n];
            const int frame = sc.frame, offset = sc.at;
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
            } else if (frame == -2) {
                // no samples left: psx_audio_xa_encode writes nothing (adpcm.c:310) and the reference puts out whatever its
                // stack buffer held; zero here
                memset(dst, 0, ssz);
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
        if (nf == 0) return;
        const int key[4] = {s->video_codec, s->video_width, s->video_height, pl.pub.max_frame_size};
        if (!c->mdec || memcmp(key, c->key, sizeof key) != 0) {
            psxhip_mdec_multi_destroy(c->mdec);
            c->mdec = nullptr;
            rc_video = psxhip_mdec_multi_create(&c->mdec, c->devices.data(), (int)c->devices.size(), s->video_codec, s->video_width,
                                                s->video_height, pl.pub.max_frame_size);
            if (rc_video == PSXHIP_OK) memcpy(c->key, key, sizeof key);
        }
        const auto tv0 = std::chrono::steady_clock::now();
        if (rc_video == PSXHIP_OK)
            rc_video = psxhip_mdec_multi_encode_frames_host(c->mdec, frames, nf, pl.budgets.data(), 0, bs, ostride, res.data(),
                                                            PSXHIP_SCHED_STATIC, 0, nullptr);
        if (rc_video) snprintf(err_video, sizeof err_video, "%s", psxhip_last_error());     // thread-local text
        else {
            const auto tv1 = std::chrono::steady_clock::now();
            build_all(true);
            if (getenv("PSXHIP_STR_TRACE"))
                fprintf(stderr, "psxhip_str: video encode %.3f ms, video sectors %.3f ms\n", std::chrono::duration<double, std::milli>(tv1 - tv0).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv1).count());
        }
    });

    // ---- audio: one XA stream, concurrently (on the list's first device: 3 ms of work next to the frames')
    const auto ta0 = std::chrono::steady_clock::now();
    const int ch = s->audio_channels;
    int na = 0;
    for (const Sector& sc : pl.sectors) na += sc.frame == -1;
    int rc_audio = PSXHIP_OK;
    if (na > 0) {
        // The XA encoder is handed pl.audio_samples per channel (the sum of the sector loop's samples_length, filefmt.c:476-479).
        // A short last sector is completed from zeros: what lies past the end of the PCM data is zero in the reference's
        // decoder buffer too (decoding.c:395-398,521-527), which is what its stereo tail over-reads.  The complete
        // mode fills every audio slot, with silence once the PCM has run out.
        const int64_t need = pl.audio_samples;
        std::vector<int16_t> padded;
        const int16_t* src = pcm;
        if (pcm_samples_per_channel < need) {
            padded.assign((size_t)need * ch, 0);
            if (pcm_samples_per_channel) memcpy(padded.data(), pcm, (size_t)pcm_samples_per_channel * ch * sizeof(int16_t));
            src = padded.data();
        }
        std::vector<uint8_t> eof((size_t)na, 0);
        for (const Sector& sc : pl.sectors)
            if (sc.frame == -1) eof[(size_t)sc.at] = sc.eof;
        xa_out.resize((size_t)na * ssz);
        psxhip_adpcm_state_t st[2] = {{0, 0}, {0, 0}};
        const int32_t lba0 = 0;
        const int fmt = s->format == FORMAT_STRCD ? 1 : 0;
        rc_audio = psxhip_xa_encode_streams_host_flags(device, fmt, ch == 2, s->audio_frequency, s->audio_bit_depth,
                                                       s->audio_xa_file, s->audio_xa_channel, src, 1, need * ch, (int)need,
                                                       &lba0, st, xa_out.data(), (int64_t)xa_out.size(), 0, eof.data());
        if (rc_audio > 0) rc_audio = rc_audio == (int)(na * ssz) ? PSXHIP_OK : PSXHIP_EINVAL;
    }
    const auto ta1 = std::chrono::steady_clock::now();
    if (rc_audio == PSXHIP_OK) build_all(false);
    if (getenv("PSXHIP_STR_TRACE"))
        fprintf(stderr, "psxhip_str: audio encode done at %.3f ms after entry to the audio leg, audio sectors %.3f ms\n",
                std::chrono::duration<double, std::milli>(ta1 - ta0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta1).count());
    video.join();
    if (rc_video) {
        psxhip_set_error("psxhip_str_encode_host: video: %s", err_video);
        return rc_video;
    }
    if (rc_audio) return rc_audio;

    long long qsum = 0;
    for (int f = 0; f < nf; f++) qsum += res[(size_t)f].quant_scale;
    pl.pub.quant_scale_sum = qsum;
    if (plan_out) *plan_out = pl.pub;
    return PSXHIP_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Device-resident: frames and PCM in HBM -> muxed sectors in HBM.  n_streams independent streams of the same
// settings and lengths in one call: the frames of all streams are ONE batched MDEC launch (per-frame budgets from the plan),
// their XA tracks are chains of ONE speculate-and-verify session -- S x 2 chains share the verify passes' latency, which is what
// bounds a single stream (its one tonal XA track is re-encoded serially for ~3 ms while the frames take 0.2) -- the video sectors
// are built by a scatter kernel (sector header, subheaders, chunk header, 2016-byte slice, form-1 EDC) and the audio sectors are
// assembled straight into their slots of the stream.  No PCIe, no host interleave.
#define DEV_TRY(expr, code)                                                                          \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            psxhip_set_error("psxhip_str_encode_device: %s failed: %s", #expr, hipGetErrorString(e__)); \
            return (code);                                                                           \
        }                                                                                            \
    } while (0)

extern "C" int psxhip_str_encode_device(psxhip_str_ctx_t* c, const psxhip_str_settings_t* s, int n_streams, const uint8_t* d_frames,
                                        size_t frames_stream_stride, int n_frames, const int16_t* d_pcm, int64_t pcm_stream_stride,
                                        int64_t pcm_samples_per_channel, uint8_t* d_out, size_t out_stream_stride,
                                        psxhip_str_plan_t* plan_out, void* stream) {
    if (!c || n_streams < 1 || n_streams > 4096) {
        psxhip_set_error("psxhip_str_encode_device: NULL handle or bad stream count");
        return PSXHIP_EINVAL;
    }
    if (plan_out) memset(plan_out, 0, sizeof *plan_out);
    // (what the header promises the kernels: 4-byte aligned pointers and strides; a chain's sample limit is an int)
    if (pcm_samples_per_channel < 0 || pcm_samples_per_channel > 0x7FFFFFFFll || ((uintptr_t)d_pcm & 3) || (n_streams > 1 && (pcm_stream_stride & 1)) ||
        ((uintptr_t)d_frames & 3) || ((uintptr_t)d_out & 3) || (frames_stream_stride & 3) || (out_stream_stride & 3)) {
        psxhip_set_error("psxhip_str_encode_device: pointers and strides must be 4-byte aligned, pcm_samples_per_channel within 0 .. 2^31 - 1");
        return PSXHIP_EINVAL;
    }
    std::lock_guard<std::mutex> call(c->mu);
    psxhip_str_ctx::Dev& d = c->dev;
    const int device = c->devices[0];
    DEV_TRY(hipSetDevice(device), PSXHIP_EDEVICE);
    hipStream_t S = (hipStream_t)stream;
    // ---- the plan and its device tables: rebuilt when the shape of the job changes
    const bool same = d.n_frames == n_frames && d.pcm_samples == pcm_samples_per_channel && d.n_streams == n_streams && s &&
                      memcmp(&d.settings, s, sizeof *s) == 0;
    if (!same) {
        Plan pl;
        memset(&pl.pub, 0, sizeof pl.pub);
        const int rc = make_plan(s, n_frames, pcm_samples_per_channel, &pl);
        if (rc) return rc;
        if (plan_out) *plan_out = pl.pub;
        // nothing of the old shape may still be running on the buffers that are about to go
        DEV_TRY(hipStreamSynchronize(S), PSXHIP_EDEVICE);
        if (d.astream) DEV_TRY(hipStreamSynchronize(d.astream), PSXHIP_EDEVICE);
        if (d.session) { psxhip_adpcm_session_destroy(d.session); d.session = nullptr; }
        void** bufs[] = {&d.d_vtab, &d.d_budgets, &d.d_adst, &d.d_eof, &d.d_bs, &d.d_res, &d.d_units, &d.d_chains, &d.d_base, &d.d_states};
        for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
        if (d.h_res) { (void)hipHostFree(d.h_res); d.h_res = nullptr; }
        d.n_frames = -1;
        d.plan = pl;
        d.settings = *s;
        const int nf = pl.pub.n_frames_encoded, ns = pl.pub.n_sectors;
        std::vector<int32_t> vtab, adst;
        std::vector<uint8_t> eof;
        for (int n = 0; n < ns; n++) {
            const Sector& sc = pl.sectors[(size_t)n];
            if (sc.frame == -1) {
                adst.push_back(n);
                eof.push_back(sc.eof);
            } else {
                vtab.push_back(n);
                vtab.push_back(sc.frame >= 0 ? sc.frame : -2);
                vtab.push_back(sc.frame >= 0 ? sc.at : 0);
                vtab.push_back(sc.frame >= 0 ? pl.budgets[(size_t)sc.frame] : 0);
            }
        }
        d.n_vtab = (int)(vtab.size() / 4);
        d.na = (int)adst.size();
        d.nf = nf;
        const size_t ostride = ((size_t)pl.pub.max_frame_size + 3) & ~(size_t)3;
        std::vector<int32_t> budgets((size_t)nf * n_streams);
        for (int i = 0; i < n_streams; i++)
            for (int f = 0; f < nf; f++) budgets[(size_t)i * nf + f] = pl.budgets[(size_t)f];
        const int ch = s->audio_channels, upg = s->audio_bit_depth == 4 ? 8 : 4;
        const size_t units_per_stream = (size_t)d.na * 18 * upg;
        auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
            hipError_t e = hipMalloc(dst, bytes ? bytes : 4);
            if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
            return e;
        };
        DEV_TRY(up(&d.d_vtab, vtab.data(), vtab.size() * 4), PSXHIP_ENOMEM);
        DEV_TRY(up(&d.d_budgets, budgets.data(), budgets.size() * 4), PSXHIP_ENOMEM);
        DEV_TRY(up(&d.d_adst, adst.data(), adst.size() * 4), PSXHIP_ENOMEM);
        DEV_TRY(up(&d.d_eof, eof.data(), eof.size()), PSXHIP_ENOMEM);
        const size_t bs_bytes = ostride * (size_t)nf * n_streams + 16;          // (never an empty allocation: a stream may hold no frame at all)
        DEV_TRY(hipMalloc(&d.d_bs, bs_bytes), PSXHIP_ENOMEM);
        DEV_TRY(hipMemset(d.d_bs, 0, bs_bytes), PSXHIP_EDEVICE);      // (rows wider than a frame's own budget read as zero there)
        DEV_TRY(hipMalloc(&d.d_res, sizeof(psxhip_mdec_result_t) * (size_t)(nf ? nf : 1) * n_streams), PSXHIP_ENOMEM);
        DEV_TRY(hipHostMalloc((void**)&d.h_res, sizeof(psxhip_mdec_result_t) * (size_t)(nf ? nf : 1) * n_streams, hipHostMallocDefault), PSXHIP_ENOMEM);
        DEV_TRY(hipMalloc(&d.d_units, (units_per_stream ? units_per_stream : 1) * PSXHIP_ADPCM_RECORD_BYTES * n_streams), PSXHIP_ENOMEM);
        if (!d.astream) DEV_TRY(hipStreamCreateWithFlags(&d.astream, hipStreamNonBlocking), PSXHIP_EDEVICE);
        if (!d.ev_in) DEV_TRY(hipEventCreateWithFlags(&d.ev_in, hipEventDisableTiming), PSXHIP_EDEVICE);
        if (!d.ev_audio) DEV_TRY(hipEventCreateWithFlags(&d.ev_audio, hipEventDisableTiming), PSXHIP_EDEVICE);
        (void)ch;
        d.n_frames = n_frames;
        d.pcm_samples = pcm_samples_per_channel;
        d.n_streams = n_streams;
        d.d_pcm = nullptr;           // (the ADPCM session is built below, over this call's PCM)
    }
    const Plan& pl = d.plan;
    if (plan_out) *plan_out = pl.pub;
    const int ns = pl.pub.n_sectors, nf = d.nf, na = d.na;
    if (ns == 0) return PSXHIP_OK;
    const size_t ssz = (size_t)pl.pub.sector_size;
    const size_t fsz = (size_t)s->video_width * s->video_height * 3 / 2;
    if ((nf && !d_frames) || !d_out || (na && !d_pcm) || out_stream_stride < ssz * (size_t)ns || (out_stream_stride & 3) ||
        ((uintptr_t)d_out & 3) || (nf && ((frames_stream_stride < fsz * (size_t)n_frames && n_streams > 1) || (frames_stream_stride & 3) || ((uintptr_t)d_frames & 3)))) {
        psxhip_set_error("psxhip_str_encode_device: NULL / misaligned argument, or a stream stride smaller than a stream");
        return PSXHIP_EINVAL;
    }
    const size_t ostride = ((size_t)pl.pub.max_frame_size + 3) & ~(size_t)3;

    // ---- video: one batched launch over the frames of all streams, then the sector kernel, both on the caller's stream
    if (nf) {
        const int key[4] = {s->video_codec, s->video_width, s->video_height, pl.pub.max_frame_size};
        if (!d.mdec || memcmp(key, d.mdec_key, sizeof key) != 0) {
            if (d.mdec) { psxhip_mdec_destroy(d.mdec); d.mdec = nullptr; }
            const int rc = psxhip_mdec_create(&d.mdec, device, s->video_codec, s->video_width, s->video_height, pl.pub.max_frame_size);
            if (rc) return rc;
            memcpy(d.mdec_key, key, sizeof key);
        }
        std::vector<psxhip_mdec_batch_t> batches;
        const bool contiguous = n_streams == 1 || (frames_stream_stride == fsz * (size_t)nf && nf == n_frames);
        for (int i = 0; i < (contiguous ? 1 : n_streams); i++) {
            psxhip_mdec_batch_t b;
            b.d_frames = d_frames + (size_t)i * frames_stream_stride;
            b.n_frames = contiguous ? nf * n_streams : nf;
            b.reserved = 0;
            b.d_frame_max_sizes = (const int32_t*)d.d_budgets + (size_t)i * nf;
            b.d_out = (uint8_t*)d.d_bs + (size_t)i * nf * ostride;
            b.d_results = (psxhip_mdec_result_t*)d.d_res + (size_t)i * nf;
            batches.push_back(b);
        }
        int rc = psxhip_mdec_encode_batches_device(d.mdec, batches.data(), (int)batches.size(), fsz, 0, ostride, S);
        if (rc) return rc;
        psxhip_str_video_job_t vj;
        vj.d_bs = (const uint8_t*)d.d_bs;
        vj.bs_stride = ostride;
        vj.bs_stream_stride = ostride * (size_t)nf;
        vj.d_res = (const psxhip_mdec_result_t*)d.d_res;
        vj.frames_per_stream = nf;
        vj.d_tab = (const int32_t*)d.d_vtab;
        vj.n_entries = d.n_vtab;
        vj.n_streams = n_streams;
        vj.format = s->format;
        vj.sector_size = (int)ssz;
        vj.xa_file = s->audio_xa_file;
        vj.xa_channel = s->audio_xa_channel;
        vj.video_id = s->str_video_id;
        vj.width = s->video_width;
        vj.height = s->video_height;
        vj.d_out = d_out;
        vj.out_stream_stride = out_stream_stride;
        rc = psxhip_str_video_sectors_launch(device, &vj, S);
        if (rc) return rc;
        DEV_TRY(hipMemcpyAsync(d.h_res, d.d_res, sizeof(psxhip_mdec_result_t) * (size_t)nf * n_streams, hipMemcpyDeviceToHost, S), PSXHIP_EDEVICE);
    } else if (d.n_vtab) {
        // no frame in the stream, but audio slots without samples: zero sectors (nothing reads a bitstream)
        psxhip_str_video_job_t vj;
        memset(&vj, 0, sizeof vj);
        vj.d_bs = (const uint8_t*)d.d_bs; vj.d_res = (const psxhip_mdec_result_t*)d.d_res; vj.d_tab = (const int32_t*)d.d_vtab;
        vj.n_entries = d.n_vtab; vj.n_streams = n_streams; vj.format = s->format; vj.sector_size = (int)ssz; vj.d_out = d_out;
        vj.out_stream_stride = out_stream_stride;
        const int rc = psxhip_str_video_sectors_launch(device, &vj, S);
        if (rc) return rc;
    }

    // ---- audio: the streams' XA tracks as chains of one session on the handle's own stream, behind the caller's inputs; the
    //      host drives the verify passes (the call is synchronous), the video leg above runs meanwhile
    if (na) {
        const int ch = s->audio_channels, bits = s->audio_bit_depth, upg = bits == 4 ? 8 : 4;
        const int units_per_stream = na * 18 * upg, units_per_chain = units_per_stream / ch;
        const int64_t need = pl.audio_samples;                                   // per channel, over the whole stream
        const int limit = (int)(pcm_samples_per_channel < need ? pcm_samples_per_channel : need);      // past it the encoder reads zeros (decoding.c:521-527)
        if (n_streams > 1 && pcm_stream_stride < (int64_t)limit * ch) {
            psxhip_set_error("psxhip_str_encode_device: pcm_stream_stride smaller than a stream's samples");
            return PSXHIP_EINVAL;
        }
        DEV_TRY(hipEventRecord(d.ev_in, S), PSXHIP_EDEVICE);
        DEV_TRY(hipStreamWaitEvent(d.astream, d.ev_in, 0), PSXHIP_EDEVICE);
        const int n_chains = n_streams * ch;
        const bool chunked = units_per_chain >= psxhip_adpcm_chunked_threshold(n_chains);      // (the rule of the host entry points)
        // an error from here on leaves nothing of this call in flight on the caller's buffers
        auto fail = [&](int code) { (void)hipStreamSynchronize(d.astream); (void)hipStreamSynchronize(S); return code; };
        if (d.d_pcm != d_pcm || d.pcm_stream_stride != pcm_stream_stride || d.chunked != chunked) {
            d.d_pcm = nullptr;          // (set again when the new session / tables stand: a failure below must not leave the old key on torn-down state)
            if (d.session) { psxhip_adpcm_session_destroy(d.session); d.session = nullptr; }
            std::vector<psxhip_adpcm_chain_t> chains((size_t)n_chains);
            std::vector<int32_t> base((size_t)n_chains);
            for (int i = 0; i < n_streams; i++)
                for (int k = 0; k < ch; k++) {
                    psxhip_adpcm_chain_t& cd = chains[(size_t)i * ch + k];
                    cd.sample_offset = (int64_t)i * pcm_stream_stride + k;
                    cd.pitch = ch;
                    cd.sample_limit = limit;
                    cd.n_units = units_per_chain;
                    cd.unit_stride = ch;
                    base[(size_t)i * ch + k] = i * units_per_stream + k;
                }
            if (chunked) {
                int chunk_units = 0, warmup_units = 0;
                psxhip_adpcm_pick_chunking((long long)units_per_chain * n_chains, 5, device, &chunk_units, &warmup_units);
                const int rc = psxhip_adpcm_session_create(&d.session, device, d_pcm, chains.data(), base.data(), nullptr, n_chains, 4, bits,
                                                           (uint8_t*)d.d_units, chunk_units, warmup_units, d.astream);
                if (rc) return fail(rc);
            } else {
                void** bufs[] = {&d.d_chains, &d.d_base, &d.d_states};
                for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
                if (hipMalloc(&d.d_chains, chains.size() * sizeof(chains[0])) != hipSuccess || hipMalloc(&d.d_base, base.size() * 4) != hipSuccess ||
                    hipMalloc(&d.d_states, chains.size() * sizeof(psxhip_adpcm_state_t)) != hipSuccess) {
                    (void)hipGetLastError();
                    psxhip_set_error("psxhip_str_encode_device: out of device memory (chain tables)");
                    return fail(PSXHIP_ENOMEM);
                }
                if (hipMemcpy(d.d_chains, chains.data(), chains.size() * sizeof(chains[0]), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(d.d_base, base.data(), base.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                    psxhip_set_error("psxhip_str_encode_device: chain tables: %s", hipGetErrorString(hipGetLastError()));
                    return fail(PSXHIP_EDEVICE);
                }
            }
            d.d_pcm = d_pcm;
            d.pcm_stream_stride = pcm_stream_stride;
            d.chunked = chunked;
        }
        int rc;
        if (chunked) {
            std::vector<psxhip_adpcm_state_t> zero((size_t)n_chains);
            memset(zero.data(), 0, zero.size() * sizeof(zero[0]));
            psxhip_adpcm_session_reset(d.session);
            rc = psxhip_adpcm_session_run(d.session, zero.data(), nullptr, 0, nullptr, nullptr);
            if (rc < 0) return fail(rc);
        } else {
            DEV_TRY(hipMemsetAsync(d.d_states, 0, (size_t)n_chains * sizeof(psxhip_adpcm_state_t), d.astream), PSXHIP_EDEVICE);
            rc = psxhip_adpcm_encode_chains_device(device, d_pcm, (const psxhip_adpcm_chain_t*)d.d_chains, (const int32_t*)d.d_base, n_chains, 4,
                                                   bits, (psxhip_adpcm_state_t*)d.d_states, (uint8_t*)d.d_units, d.astream);
            if (rc) return fail(rc);
        }
        rc = psxhip_xa_assemble_scatter(device, (const uint8_t*)d.d_units, na, s->format == FORMAT_STRCD ? 1 : 0, ch == 2, s->audio_frequency, bits,
                                        s->audio_xa_file, s->audio_xa_channel, 0, (const uint8_t*)d.d_eof, 0u, d_out, (const int32_t*)d.d_adst,
                                        n_streams, (size_t)units_per_stream * PSXHIP_ADPCM_RECORD_SIZE(bits), out_stream_stride, d.astream);
        if (rc) return fail(rc);
        DEV_TRY(hipEventRecord(d.ev_audio, d.astream), PSXHIP_EDEVICE);
        DEV_TRY(hipStreamWaitEvent(S, d.ev_audio, 0), PSXHIP_EDEVICE);
    }
    DEV_TRY(hipStreamSynchronize(S), PSXHIP_EDEVICE);
    long long qsum = 0;
    for (size_t i = 0; i < (size_t)nf * n_streams; i++) {
        if (d.h_res[i].quant_scale >= 64) {
            psxhip_set_error("psxhip_str_encode_device: frame %zu of stream %zu does not fit its budget at any quant scale", i % (size_t)nf, i / (size_t)nf);
            return PSXHIP_ENOFIT;
        }
        qsum += d.h_res[i].quant_scale;
    }
    if (plan_out) plan_out->quant_scale_sum = qsum;
    return PSXHIP_OK;
}
#undef DEV_TRY
