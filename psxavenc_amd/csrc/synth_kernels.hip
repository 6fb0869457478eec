// synth_kernels.hip -- integer-only synthetic NV21 frames and PCM, generated directly in HBM.
// Same pure function of (seed, frame/chain index, sample index) as oracle/synth.c (the CPU twin used by
// the tests); there is no reference counterpart (the reference's inputs come from FFmpeg,
// psxavenc/decoding.c, which is out of scope).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psxhip_internal.h"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ int noise(uint32_t key, uint32_t idx, int amp) {
    if (amp <= 0) return 0;
    const uint32_t r = mix32(key ^ (idx * 0x85EBCA77u + 0x165667B1u));
    return (int)(r % (uint32_t)(2 * amp + 1)) - amp;
}
__device__ __forceinline__ int clamp_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one thread per 4 output bytes
__global__ void synth_frames_kernel(uint8_t* frames, size_t frame_stride, int w, int h, uint32_t seed,
                                    uint32_t first_frame, int n_frames, int amp) {
    const int fbytes = w * h * 3 / 2;
    const int quads = fbytes / 4;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)quads * n_frames) return;
    const int f = (int)(gid / quads), q = (int)(gid - (long long)f * quads);
    const uint32_t fi = first_frame + (uint32_t)f;
    const uint32_t key = mix32(seed + fi * 0x9E3779B1u);
    const int cw = w / 2, ch = h / 2;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t idx = (uint32_t)(q * 4 + k);
        int v;
        if (idx < (uint32_t)(w * h)) {
            const int y = (int)(idx / (uint32_t)w), x = (int)(idx - (uint32_t)y * (uint32_t)w);
            const int xs = (int)(((uint32_t)x + 3u * fi) % (uint32_t)w);
            v = (xs * 255 / w + y * 255 / h) / 2 + noise(key, idx, amp);
        } else {
            const uint32_t c = idx - (uint32_t)(w * h);
            const int y = (int)(c / (uint32_t)w), xx = (int)(c - (uint32_t)y * (uint32_t)w);
            const int x = xx >> 1;
            if ((xx & 1) == 0) {
                const int xs = (int)(((uint32_t)x + fi) % (uint32_t)cw);
                v = 128 + xs * 64 / cw - 32 + noise(key, idx, amp / 2);
            } else {
                v = 128 + 32 - y * 64 / ch + noise(key, idx, amp / 2);
            }
        }
        packed |= (uint32_t)clamp_u8(v) << (8 * k);
    }
    *(uint32_t*)(frames + (size_t)f * frame_stride + (size_t)q * 4) = packed;
}

__device__ __forceinline__ int par_sin(uint32_t phase) {
    const int t = (int)(phase & 0x7FFFu);
    const int v = (t * (32768 - t)) >> 13;
    return (phase & 0x8000u) ? -v : v;
}

__global__ void synth_pcm_kernel(int16_t* pcm, uint32_t seed, uint32_t chain, long long first, long long n, int kind,
                                 int pitch) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = mix32(seed ^ (chain * 0xC2B2AE35u + 0x27D4EB2Fu));
    const uint32_t step1 = 700u + 37u * (chain % 16u), step2 = 2311u + 101u * (chain % 7u);
    const unsigned long long s = (unsigned long long)(first + i);
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
    pcm[i * pitch] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}

}  // namespace

int psxhip_ensure_device(int device);

extern "C" int psxhip_synth_frames_device(int device, uint8_t* d_frames, size_t frame_stride, int width, int height,
                                          uint32_t seed, uint32_t first_frame, int n_frames, int noise_amp, void* stream) {
    if (!d_frames || width <= 0 || height <= 0 || (width % 16) || (height % 16) || n_frames < 0 || (frame_stride & 3) ||
        ((uintptr_t)d_frames & 3)) {
        psxhip_set_error("synth_frames: bad argument");
        return PSXHIP_EINVAL;
    }
    int rc = psxhip_ensure_device(device);
    if (rc) return rc;
    if (n_frames == 0) return PSXHIP_OK;
    const long long total = (long long)(width * height * 3 / 8) * n_frames;
    const int threads = 256;
    const long long blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(synth_frames_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, d_frames,
                       frame_stride, width, height, seed, first_frame, n_frames, noise_amp);
    if (hipGetLastError() != hipSuccess) {
        psxhip_set_error("synth_frames: launch failed");
        return PSXHIP_EDEVICE;
    }
    return PSXHIP_OK;
}

extern "C" int psxhip_synth_pcm_device(int device, int16_t* d_pcm, uint32_t seed, uint32_t chain, int64_t first_sample,
                                       int64_t n, int kind, int pitch, void* stream) {
    if (!d_pcm || n < 0 || pitch < 1) {
        psxhip_set_error("synth_pcm: bad argument");
        return PSXHIP_EINVAL;
    }
    int rc = psxhip_ensure_device(device);
    if (rc) return rc;
    if (n == 0) return PSXHIP_OK;
    const int threads = 256;
    const long long blocks = (n + threads - 1) / threads;
    hipLaunchKernelGGL(synth_pcm_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, d_pcm, seed, chain,
                       (long long)first_sample, (long long)n, kind, pitch);
    if (hipGetLastError() != hipSuccess) {
        psxhip_set_error("synth_pcm: launch failed");
        return PSXHIP_EDEVICE;
    }
    return PSXHIP_OK;
}
