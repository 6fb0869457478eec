// psxhip_audio_api.cpp -- host-buffer batches for the ADPCM path (include/psxav_hip.h): build the chain
// descriptors, move buffers, launch the chain / pack / assemble kernels.  No encoding happens on the host.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "psxhip_internal.h"

int psxhip_ensure_device(int device);

namespace {

#define HIP_TRY(expr, code)                                                                   \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            psxhip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (code);                                                                    \
        }                                                                                     \
    } while (0)

// streams at least this long are also split along time (psxhip_adpcm_encode_chains_chunked)
const int kChunkedThreshold = []() {
    if (const char* e = getenv("PSXHIP_ADPCM_CHUNK_THRESHOLD")) { const int v = atoi(e); if (v >= 64) return v; }      // experiments
    return 4096;
}();
// ... and from 512 units when the call has only a few chains: one chain of 788 blocks (config `spu`'s second of audio) encoded serially
// by one wavefront takes 1.35 ms, cut along time 0.98 ms; 2048 blocks 3.35 ms against 0.97 (the chunked path's set-up and verify round
// trips are ~0.75 ms whatever the length, so below ~500 units serial wins; tools/gpu_r05_session_k.sh).  Many short chains keep the
// serial kernel: it runs them all side by side.
inline int chunked_threshold(int n_chains) {
    static const bool forced = getenv("PSXHIP_ADPCM_CHUNK_THRESHOLD") != nullptr;      // experiments; read once
    if (forced) return kChunkedThreshold;
    return n_chains <= 8 ? 512 : kChunkedThreshold;
}

// chunk length: long enough that verify needs few passes (a wrong guess travels one chunk per pass, and tonal material
// does not fall into the same state within a thousand units), short enough that there are >= ~8 k chunks -- 1600+
// wavefronts -- to fill 256 CUs (tools/gpu_adpcm_sweep.py: 13 M units, tonal: chunks of 1024 are 14 % faster than 512;
// noise: within 3 %); warm-up 16 .. 64 units.  Jobs that are large enough for it get the length at which every wavefront
// slot of the GPU (8 per SIMD) holds exactly ONE wavefront of `rows` chunks: the encoder is a dependent chain per unit, so a
// SIMD with four wavefronts runs its VALU at 84 % and one with eight at ~100 %, and a launch of 3.7 wavefronts per SIMD ends
// when the SIMDs that drew four do (config 5, 78 M units on one GPU: 1899 instead of 4096 units per chunk, 21.3 -> 24.4 M
// sectors/s, still two verify passes; 1266: three passes, 950: four -- tools/gpu_xacd_chunk_sweep.sh).
inline void pick_chunking(long long total_units, int rows, int device, int* chunk_units, int* warmup_units) {
    // experiments (tools/gpu_r06_chunk_sweep.sh): PSXHIP_ADPCM_CHUNK / PSXHIP_ADPCM_WARM fix both, read once
    static const int forced_chunk = [] { const char* e = getenv("PSXHIP_ADPCM_CHUNK"); return e ? atoi(e) : 0; }();
    static const int forced_warm = [] { const char* e = getenv("PSXHIP_ADPCM_WARM"); return e ? atoi(e) : 128; }();
    if (forced_chunk > 0) { *chunk_units = forced_chunk; *warmup_units = forced_warm; return; }
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n_cu < 1) n_cu = 256;
    const long long per_round = 32ll * n_cu * rows;                        // chunks in flight when every slot holds a wavefront
    const long long fill = (total_units + per_round - 1) / per_round;
    if (fill >= 1024) {
        const long long rounds = (fill + 4095) / 4096;
        *chunk_units = (int)((total_units + per_round * rounds - 1) / (per_round * rounds));
        *warmup_units = 64;
        return;
    }
    long long c = total_units / 8192;
    int p = 64;
    while (p * 2 <= c && p < 1024) p *= 2;
    *chunk_units = p;
    *warmup_units = p >= 1024 ? 32 : 16;     // noisy material converges within a few units, tonal material not within 1024 either
}

}  // namespace
extern "C" int psxhip_adpcm_chunked_threshold(int n_chains) { return chunked_threshold(n_chains); }
extern "C" void psxhip_adpcm_pick_chunking(long long total_units, int rows, int device, int* chunk_units, int* warmup_units) {
    pick_chunking(total_units, rows, device, chunk_units, warmup_units);
}
namespace {

// Device scratch of the host-buffer entry points.  The reference calls psx_audio_spu_encode once per 28 samples and
// psx_audio_xa_encode once per sector (filefmt.c:243,184): allocating half a dozen device buffers per call would cost far
// more than the encode, so every host thread keeps its scratch buffers between calls (grown on demand, released by
// psxhip_release_scratch() or with the process).
struct ScratchPool {
    static constexpr int kSlots = 8;
    void* p[kSlots] = {nullptr};
    size_t cap[kSlots] = {0};
    int device = -1;
    void release() {
        for (int i = 0; i < kSlots; i++) {
            if (p[i]) (void)hipFree(p[i]);
            p[i] = nullptr;
            cap[i] = 0;
        }
    }
    hipError_t get(int slot, int dev, size_t n, void** out) {
        if (dev != device) {
            release();
            device = dev;
        }
        if (n > cap[slot]) {
            if (p[slot]) (void)hipFree(p[slot]);
            p[slot] = nullptr;
            cap[slot] = 0;
            const size_t want = n + n / 2 + 256;
            hipError_t e = hipMalloc(&p[slot], want);
            if (e != hipSuccess) return e;
            cap[slot] = want;
        }
        *out = p[slot];
        return hipSuccess;
    }
};
thread_local ScratchPool g_pool;
thread_local int g_pool_device = 0;

struct DevBuf {
    void* p = nullptr;
    int slot;
    explicit DevBuf(int s) : slot(s) {}
    hipError_t alloc(size_t n) { return g_pool.get(slot, g_pool_device, n ? n : 4, &p); }
    template <typename T> T* as() { return (T*)p; }
};

// The per-call fast path (the reference calls psx_audio_spu_encode once per 28 samples and psx_audio_xa_encode once per sector,
// filefmt.c:243,184): one page-locked, device-visible block per host thread -- samples in, blocks / sectors and states out -- and
// one stream.  A call is: copy the samples in (CPU), ONE launch (two for XA: chains, sector assembly), wait, copy the result out.
struct CallScratch {
    static constexpr size_t kIn = 32 << 10, kOut = 16 << 10, kStates = 256;
    uint8_t* h = nullptr;       // [kIn samples | kOut blocks or sectors | kStates]
    uint8_t* d = nullptr;       // the same block as the device addresses it
    hipStream_t stream = nullptr;
    int device = -1;
    bool disabled = false;
    void release() {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (h) (void)hipHostFree(h);
        h = nullptr;
        stream = nullptr;
        device = -1;
    }
    ~CallScratch() { release(); }
    bool ready(int dev) {
        if (disabled) return false;
        if (h && device == dev) return true;
        release();
        if (getenv("PSXHIP_NO_PERCALL_PATH")) { disabled = true; return false; }      // experiments / A-B measurements
        if (hipHostMalloc((void**)&h, kIn + kOut + kStates, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            h = nullptr;
            return false;
        }
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, h, 0) != hipSuccess || !dp) {
            (void)hipGetLastError();
            (void)hipHostFree(h);
            h = nullptr;
            return false;
        }
        d = (uint8_t*)dp;
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(h);
            h = nullptr;
            stream = nullptr;
            return false;
        }
        device = dev;
        return true;
    }
    int16_t* in() { return (int16_t*)h; }
    uint8_t* out() { return h + kIn; }
    psxhip_adpcm_state_t* states() { return (psxhip_adpcm_state_t*)(h + kIn + kOut); }
    const int16_t* d_in() { return (const int16_t*)d; }
    uint8_t* d_out() { return d + kIn; }
    psxhip_adpcm_state_t* d_states() { return (psxhip_adpcm_state_t*)(d + kIn + kOut); }
};
thread_local CallScratch g_call;

}  // namespace

extern "C" void psxhip_release_scratch(void) {
    g_pool.release();
    g_call.release();
}

extern "C" int psxhip_spu_encode_streams_host(int device, const int16_t* samples, int n_streams, int64_t stream_stride,
                                              int pitch, int samples_per_stream, psxhip_adpcm_state_t* states,
                                              uint8_t* out, int64_t out_stride) {
    if (!samples || !states || !out || n_streams < 0 || samples_per_stream < 0 || pitch < 1) {
        psxhip_set_error("spu_encode_streams_host: bad argument");
        return PSXHIP_EINVAL;
    }
    const int n_units = (samples_per_stream + 27) / 28;
    const int bytes = n_units * 16;
    if (n_streams == 0 || n_units == 0) return bytes;
    if (n_streams == 1) out_stride = bytes;     /* a single stream needs no pitch */
    if (out_stride < bytes) {
        psxhip_set_error("spu_encode_streams_host: out_stride %lld < %d bytes per stream", (long long)out_stride, bytes);
        return PSXHIP_EINVAL;
    }
    int rc = psxhip_ensure_device(device);
    if (rc) return rc;

    // ---- the reference's call pattern (a few streams, a few blocks): one launch, no copies (see CallScratch)
    if (n_streams <= 4 && n_units < chunked_threshold(n_streams) && (size_t)n_streams * ((size_t)n_units * 28 + 8) <= (size_t)psxhip_adpcm_call_stage_max() &&
        (size_t)n_streams * bytes <= CallScratch::kOut && g_call.ready(device)) {
        psxhip_adpcm_call_t a;
        memset(&a, 0, sizeof a);
        const size_t row = ((size_t)n_units * 28 + 7) & ~(size_t)7;      // staged row per stream: pitch 1, zero-padded to whole units
        int16_t* in = g_call.in();
        for (int i = 0; i < n_streams; i++) {
            const int16_t* src = samples + (n_streams == 1 ? 0 : (size_t)i * (size_t)stream_stride);
            int16_t* dst = in + (size_t)i * row;
            if (pitch == 1) memcpy(dst, src, (size_t)samples_per_stream * sizeof(int16_t));
            else for (int k = 0; k < samples_per_stream; k++) dst[k] = src[(size_t)k * pitch];
            memset(dst + samples_per_stream, 0, (row - (size_t)samples_per_stream) * sizeof(int16_t));
            a.chains[i].sample_offset = (int64_t)i * (int64_t)row;
            a.chains[i].pitch = 1;
            a.chains[i].sample_limit = samples_per_stream;
            a.chains[i].n_units = n_units;
            a.chains[i].unit_stride = 1;
            a.unit_base[i] = i * n_units;
            a.states_in[i] = states[i];
        }
        a.samples = g_call.d_in();
        a.stage_elems = (int)(row * n_streams);
        a.n_chains = n_streams;
        a.filter_count = 5;
        a.bits = 4;
        a.states_out = g_call.d_states();
        a.spu_out = g_call.d_out();
        HIP_TRY(psxhip_adpcm_call_launch(&a, g_call.stream), PSXHIP_EDEVICE);
        HIP_TRY(hipStreamSynchronize(g_call.stream), PSXHIP_EDEVICE);
        for (int i = 0; i < n_streams; i++) {
            memcpy(out + (size_t)i * (size_t)out_stride, g_call.out() + (size_t)i * bytes, (size_t)bytes);
            states[i] = g_call.states()[i];
        }
        return bytes;
    }

    const size_t per = (size_t)samples_per_stream * pitch;          // device row stride (elements)
    // the reference reads samples[i * pitch] for i < n (adpcm.c:65,110): only (n - 1) * pitch + 1 elements of a
    // stream are the caller's to read -- `samples + channel` with pitch = channels ends before a full n * pitch
    const size_t readable = samples_per_stream ? (size_t)(samples_per_stream - 1) * pitch + 1 : 0;
    if (n_streams == 1) stream_stride = (int64_t)readable;        // a single stream needs no stride
    std::vector<psxhip_adpcm_chain_t> chains(n_streams);
    std::vector<int32_t> base(n_streams);
    for (int i = 0; i < n_streams; i++) {
        chains[i].sample_offset = (int64_t)i * (int64_t)per;
        chains[i].pitch = pitch;
        chains[i].sample_limit = samples_per_stream;
        chains[i].n_units = n_units;
        chains[i].unit_stride = 1;
        base[i] = i * n_units;
    }
    g_pool_device = device;
    DevBuf d_s(0), d_c(1), d_b(2), d_st(3), d_u(4), d_o(5);
    HIP_TRY(d_s.alloc(per * n_streams * sizeof(int16_t)), PSXHIP_ENOMEM);
    HIP_TRY(d_c.alloc(chains.size() * sizeof(chains[0])), PSXHIP_ENOMEM);
    HIP_TRY(d_b.alloc(base.size() * sizeof(int32_t)), PSXHIP_ENOMEM);
    HIP_TRY(d_st.alloc(n_streams * sizeof(psxhip_adpcm_state_t)), PSXHIP_ENOMEM);
    HIP_TRY(d_u.alloc((size_t)n_streams * n_units * PSXHIP_ADPCM_RECORD_BYTES), PSXHIP_ENOMEM);
    HIP_TRY(d_o.alloc((size_t)n_streams * bytes), PSXHIP_ENOMEM);
    hipStream_t st = nullptr;
    HIP_TRY(hipMemsetAsync(d_s.p, 0, per * n_streams * sizeof(int16_t), st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpy2DAsync(d_s.p, per * sizeof(int16_t), samples, (size_t)stream_stride * sizeof(int16_t),
                             readable * sizeof(int16_t), (size_t)n_streams, hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpyAsync(d_c.p, chains.data(), chains.size() * sizeof(chains[0]), hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpyAsync(d_b.p, base.data(), base.size() * sizeof(int32_t), hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpyAsync(d_st.p, states, n_streams * sizeof(psxhip_adpcm_state_t), hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
    if (n_units >= chunked_threshold(n_streams)) {
        // long streams: parallel along time as well (speculate-and-verify; same bytes as the serial chain kernel)
        HIP_TRY(hipStreamSynchronize(st), PSXHIP_EDEVICE);
        int chunk_units, warmup_units;
        pick_chunking((long long)n_units * n_streams, 4, device, &chunk_units, &warmup_units);
        rc = psxhip_adpcm_encode_chains_chunked(device, d_s.as<int16_t>(), chains.data(), base.data(), n_streams, 5, 4,
                                                d_st.as<psxhip_adpcm_state_t>(), d_u.as<uint8_t>(), chunk_units, warmup_units, 0, st);
        if (rc < 0) return rc;
    } else {
        rc = psxhip_adpcm_encode_chains_device(device, d_s.as<int16_t>(), d_c.as<psxhip_adpcm_chain_t>(), d_b.as<int32_t>(),
                                               n_streams, 5, 4, d_st.as<psxhip_adpcm_state_t>(), d_u.as<uint8_t>(), st);
        if (rc) return rc;
    }
    rc = psxhip_spu_pack_device(device, d_u.as<uint8_t>(), n_streams * n_units, d_o.as<uint8_t>(), st);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)out_stride, d_o.p, (size_t)bytes, (size_t)bytes, (size_t)n_streams,
                             hipMemcpyDeviceToHost, st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpyAsync(states, d_st.p, n_streams * sizeof(psxhip_adpcm_state_t), hipMemcpyDeviceToHost, st), PSXHIP_EDEVICE);
    HIP_TRY(hipStreamSynchronize(st), PSXHIP_EDEVICE);
    return bytes;
}

extern "C" int psxhip_xa_encode_streams_host(int device, int format, int stereo, int frequency, int bits, int file_number,
                                             int channel_number, const int16_t* samples, int n_streams,
                                             int64_t stream_stride, int samples_per_stream, const int32_t* lbas,
                                             psxhip_adpcm_state_t* states, uint8_t* out, int64_t out_stride, int finalize) {
    return psxhip_xa_encode_streams_host_flags(device, format, stereo, frequency, bits, file_number, channel_number, samples, n_streams,
                                               stream_stride, samples_per_stream, lbas, states, out, out_stride, finalize, nullptr);
}

// eof_flags (optional, n_streams x sectors): which sectors get the EOF submode bit -- the reference's encode_file_str
// finalises EVERY audio sector once its decoder has seen the end of the input (filefmt.c:492-493), not only the last one
extern "C" int psxhip_xa_encode_streams_host_flags(int device, int format, int stereo, int frequency, int bits, int file_number,
                                                   int channel_number, const int16_t* samples, int n_streams,
                                                   int64_t stream_stride, int samples_per_stream, const int32_t* lbas,
                                                   psxhip_adpcm_state_t* states, uint8_t* out, int64_t out_stride, int finalize,
                                                   const uint8_t* eof_flags) {
    if (!samples || !states || !out || n_streams < 0 || samples_per_stream < 0 || (bits != 4 && bits != 8) ||
        (format != 0 && format != 1)) {
        psxhip_set_error("xa_encode_streams_host: bad argument");
        return PSXHIP_EINVAL;
    }
    const int ch = stereo ? 2 : 1;
    const int upg = bits == 4 ? 8 : 4;                         // units per 128-byte sound group
    const int group_samples = upg * 28;                        // interleaved samples consumed per group (adpcm.c:301)
    const int64_t total = (int64_t)samples_per_stream * ch;    // adpcm.c:307-308
    const int groups = (int)((total + group_samples - 1) / group_samples);
    const int sectors = (groups + 17) / 18;                    // the loop runs until the sector is complete (adpcm.c:310)
    const int ssz = format == 0 ? 2336 : 2352;
    const int bytes = sectors * ssz;
    if (n_streams == 0 || sectors == 0) return bytes;
    if (n_streams == 1) out_stride = bytes;
    if (out_stride < bytes) {
        psxhip_set_error("xa_encode_streams_host: out_stride %lld < %d bytes per stream", (long long)out_stride, bytes);
        return PSXHIP_EINVAL;
    }
    int rc = psxhip_ensure_device(device);
    if (rc) return rc;

    const int units_per_stream = sectors * 18 * upg;
    const int units_per_chain = units_per_stream / ch;
    const size_t per = (size_t)total;
    if (n_streams == 1) stream_stride = (int64_t)per;             // a single stream needs no stride
    // ---- the reference's call pattern (one stream, a sector or two per call, filefmt.c:184,476-491): chains + assembly, two
    //      launches, no copies (see CallScratch)
    if (n_streams == 1 && sectors <= 6 && ((per + 7) & ~(size_t)7) <= (size_t)psxhip_adpcm_call_stage_max() && (size_t)bytes <= CallScratch::kOut &&
        g_call.ready(device)) {
        g_pool_device = device;
        DevBuf d_u(4);
        HIP_TRY(d_u.alloc((size_t)units_per_stream * PSXHIP_ADPCM_RECORD_BYTES), PSXHIP_ENOMEM);
        psxhip_adpcm_call_t a;
        memset(&a, 0, sizeof a);
        const size_t row = (per + 7) & ~(size_t)7;
        memcpy(g_call.in(), samples, per * sizeof(int16_t));
        memset(g_call.in() + per, 0, (row - per) * sizeof(int16_t));
        for (int c = 0; c < ch; c++) {
            a.chains[c].sample_offset = c;
            a.chains[c].pitch = ch;
            a.chains[c].sample_limit = samples_per_stream;
            a.chains[c].n_units = units_per_chain;
            a.chains[c].unit_stride = ch;
            a.unit_base[c] = c;
            a.states_in[c] = states[c];
        }
        a.samples = g_call.d_in();
        a.stage_elems = (int)row;
        a.n_chains = ch;
        a.filter_count = 4;
        a.bits = bits;
        a.states_out = g_call.d_states();
        a.units = d_u.as<uint8_t>();
        uint32_t eof_bits = 0;
        for (int k = 0; k < sectors; k++)
            if (eof_flags ? eof_flags[k] != 0 : (finalize && k == sectors - 1)) eof_bits |= 1u << k;
        HIP_TRY(psxhip_adpcm_call_launch(&a, g_call.stream), PSXHIP_EDEVICE);
        rc = psxhip_xa_assemble_device_bits(device, d_u.as<uint8_t>(), sectors, format, stereo, frequency, bits, file_number, channel_number,
                                            lbas ? lbas[0] : 0, nullptr, eof_bits, g_call.d_out(), g_call.stream);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(g_call.stream), PSXHIP_EDEVICE);
        memcpy(out, g_call.out(), (size_t)bytes);
        for (int c = 0; c < ch; c++) states[c] = g_call.states()[c];
        return bytes;
    }
    std::vector<psxhip_adpcm_chain_t> chains((size_t)n_streams * ch);
    std::vector<int32_t> base((size_t)n_streams * ch);
    for (int i = 0; i < n_streams; i++)
        for (int c = 0; c < ch; c++) {
            psxhip_adpcm_chain_t& d = chains[(size_t)i * ch + c];
            d.sample_offset = (int64_t)i * (int64_t)per + c;
            d.pitch = ch;
            d.sample_limit = samples_per_stream;
            d.n_units = units_per_chain;
            d.unit_stride = ch;
            base[(size_t)i * ch + c] = i * units_per_stream + c;
        }
    std::vector<uint8_t> eof((size_t)n_streams * sectors, 0);
    if (eof_flags)
        memcpy(eof.data(), eof_flags, eof.size());
    else if (finalize)
        for (int i = 0; i < n_streams; i++) eof[(size_t)i * sectors + sectors - 1] = 1;

    g_pool_device = device;
    DevBuf d_s(0), d_c(1), d_b(2), d_st(3), d_u(4), d_o(5), d_e(6);
    HIP_TRY(d_s.alloc(per * n_streams * sizeof(int16_t)), PSXHIP_ENOMEM);
    HIP_TRY(d_c.alloc(chains.size() * sizeof(chains[0])), PSXHIP_ENOMEM);
    HIP_TRY(d_b.alloc(base.size() * sizeof(int32_t)), PSXHIP_ENOMEM);
    HIP_TRY(d_st.alloc(chains.size() * sizeof(psxhip_adpcm_state_t)), PSXHIP_ENOMEM);
    HIP_TRY(d_u.alloc((size_t)n_streams * units_per_stream * PSXHIP_ADPCM_RECORD_BYTES), PSXHIP_ENOMEM);
    HIP_TRY(d_o.alloc((size_t)n_streams * bytes), PSXHIP_ENOMEM);
    HIP_TRY(d_e.alloc(eof.size()), PSXHIP_ENOMEM);
    hipStream_t st = nullptr;
    if (per)
        HIP_TRY(hipMemcpy2DAsync(d_s.p, per * sizeof(int16_t), samples, (size_t)stream_stride * sizeof(int16_t),
                                 per * sizeof(int16_t), (size_t)n_streams, hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpyAsync(d_c.p, chains.data(), chains.size() * sizeof(chains[0]), hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpyAsync(d_b.p, base.data(), base.size() * sizeof(int32_t), hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpyAsync(d_st.p, states, chains.size() * sizeof(psxhip_adpcm_state_t), hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpyAsync(d_e.p, eof.data(), eof.size(), hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
    if (units_per_chain >= chunked_threshold((int)chains.size())) {
        HIP_TRY(hipStreamSynchronize(st), PSXHIP_EDEVICE);
        int chunk_units, warmup_units;
        pick_chunking((long long)units_per_chain * (long long)chains.size(), 5, device, &chunk_units, &warmup_units);
        rc = psxhip_adpcm_encode_chains_chunked(device, d_s.as<int16_t>(), chains.data(), base.data(), (int)chains.size(), 4, bits,
                                                d_st.as<psxhip_adpcm_state_t>(), d_u.as<uint8_t>(), chunk_units, warmup_units, 0, st);
        if (rc < 0) return rc;
    } else {
        rc = psxhip_adpcm_encode_chains_device(device, d_s.as<int16_t>(), d_c.as<psxhip_adpcm_chain_t>(), d_b.as<int32_t>(),
                                               (int)chains.size(), 4, bits, d_st.as<psxhip_adpcm_state_t>(), d_u.as<uint8_t>(), st);
        if (rc) return rc;
    }
    for (int i = 0; i < n_streams; i++) {
        rc = psxhip_xa_assemble_device(device, d_u.as<uint8_t>() + (size_t)i * units_per_stream * PSXHIP_ADPCM_RECORD_SIZE(bits),
                                       sectors, format, stereo, frequency, bits, file_number, channel_number,
                                       lbas ? lbas[i] : 0, d_e.as<uint8_t>() + (size_t)i * sectors,
                                       d_o.as<uint8_t>() + (size_t)i * bytes, st);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)out_stride, d_o.p, (size_t)bytes, (size_t)bytes, (size_t)n_streams,
                             hipMemcpyDeviceToHost, st), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpyAsync(states, d_st.p, chains.size() * sizeof(psxhip_adpcm_state_t), hipMemcpyDeviceToHost, st), PSXHIP_EDEVICE);
    HIP_TRY(hipStreamSynchronize(st), PSXHIP_EDEVICE);
    return bytes;
}
