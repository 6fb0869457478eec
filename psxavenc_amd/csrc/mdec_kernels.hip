// mdec_kernels.hip -- MDEC "BS" frame encoder for MI355X (gfx950), hand-written HIP.
//
// Replaces, for batches of frames resident in HBM, what the reference does per frame in
// psxavenc/mdec.c:580-755 (encode_frame_bs): NV21 -> macroblocks -> 8x8 forward DCT -> quantise ->
// zig-zag/RLE -> BS v2/v3 VLC -> bit-pack, inside the "first quant scale that fits" loop.
//
// Mapping (see DESIGN.md for the reasoning):
//   * one workgroup (16 wavefronts) owns one frame at a time and loops over frames (persistent grid);
//   * one wavefront owns one macroblock at a time: 48 lanes run the row / column DCT butterflies of
//     the six 8x8 blocks (8x8 transposes staged in LDS), then lane k owns zig-zag position k;
//   * quantisation is an exact integer rounding division done with one fp32 multiply (proof in
//     quant_level()); run lengths come from a 64-bit ballot + count-leading-zeros; code lengths and
//     codes from a (run, |level|) LUT held in LDS; bit offsets from DPP prefix sums;
//   * the rate-control loop evaluates kScalesPerPass scales per pass over the frame's coefficients
//     (kept as int16 in an L2-resident scratch slab) and takes the FIRST scale that fits, exactly
//     like the reference's ascending loop (bits(s) is not provably monotone, so no bisection);
//   * the chosen scale's bitstream is assembled in LDS with ds_or and leaves the CU as coalesced
//     dword stores, header and zero tail included (the reference's memset, mdec.c:676).
//
// MFMA is deliberately not used: the DCT is the bit-exact integer "islow" butterfly (see
// fdct8()), not a dense contraction, and everything after it is integer / bit manipulation.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bs_vlc_lut.h"
#include "psxhip_internal.h"
#include "wave_ops.h"

namespace {

// Two workgroup shapes: 12 wavefronts at 6 per SIMD (two frames per CU; 80 VGPRs) when two groups' LDS fits one CU,
// otherwise 16 wavefronts at 4 per SIMD (one frame per CU; 128 VGPRs) -- large frames / large budgets.
constexpr int kWavesSmall = 12, kOccSmall = 6;
constexpr int kWavesLarge = 16, kOccLarge = 4;
constexpr int kScalesPerPass = 4;
constexpr int kTileStride = 72;   // int16 per block in the transpose tile: 6 blocks land on disjoint LDS banks
constexpr int kZStride = 66;      // int16 per block in the zig-zag tile (+1 dword: the 6 blocks' scatters hit different banks)

__constant__ uint8_t c_ac_len[BS_LUT_SIZE];
__constant__ uint32_t c_ac_code[BS_LUT_SIZE];
__constant__ uint8_t c_zagzig[64];
__constant__ uint8_t c_quant_zz[64];
__constant__ uint8_t c_dc_prefix[2][8];   // [0] chroma, [1] luma
__constant__ uint8_t c_dc_plen[2][8];

struct FrameJob {
    const uint8_t* frames;
    size_t frame_stride;
    int width, height, nx, ny, nmb;
    int n_frames;
    const int32_t* max_sizes;
    int uniform_max_size;
    uint8_t* out;
    size_t out_stride;
    psxhip_mdec_result_t* results;
    int16_t* coef_slab;      // [gridDim.x][nmb][6][64], zig-zag order
    int out_words;           // LDS dwords reserved for one frame's output
    unsigned long long* timing;   // optional [8] phase cycle counters (diagnostics), NULL in normal runs
};

// ---------------------------------------------------------------------------------------------
// 8-point forward DCT, IJG "jfdctint" (Loeffler-Ligtenberg-Moschytz) for 8-bit samples as
// libavcodec specialises it (ff_jpeg_fdct_islow_8, the routine psxavenc's AVDCT call resolves to in
// the release configuration; mdec.c:640, .github/scripts/build.sh:36-56).  13-bit constants, 4
// fractional bits kept after the row pass.  COLUMN selects the output scaling of the second pass.
// ---------------------------------------------------------------------------------------------
template <bool COLUMN>
__device__ __forceinline__ void fdct8(int (&d)[8]) {
    constexpr int K_0_298 = 2446, K_0_390 = 3196, K_0_541 = 4433, K_0_765 = 6270, K_0_899 = 7373,
                  K_1_175 = 9633, K_1_501 = 12299, K_1_847 = 15137, K_1_961 = 16069, K_2_053 = 16819,
                  K_2_562 = 20995, K_3_072 = 25172;
    constexpr int SH = COLUMN ? 13 + 4 : 13 - 4;
    constexpr int RND = 1 << (SH - 1);

    const int s07 = d[0] + d[7], s16 = d[1] + d[6], s25 = d[2] + d[5], s34 = d[3] + d[4];
    int o0 = d[3] - d[4], o1 = d[2] - d[5], o2 = d[1] - d[6], o3 = d[0] - d[7];
    const int e0 = s07 + s34, e3 = s07 - s34, e1 = s16 + s25, e2 = s16 - s25;

    if (COLUMN) {
        d[0] = (e0 + e1 + 8) >> 4;
        d[4] = (e0 - e1 + 8) >> 4;
    } else {
        d[0] = (e0 + e1) * 16;
        d[4] = (e0 - e1) * 16;
    }
    const int r = (e2 + e3) * K_0_541;
    d[2] = (r + e3 * K_0_765 + RND) >> SH;
    d[6] = (r - e2 * K_1_847 + RND) >> SH;

    int z1 = o0 + o3, z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
    const int z5 = (z3 + z4) * K_1_175;
    o0 *= K_0_298;
    o1 *= K_2_053;
    o2 *= K_3_072;
    o3 *= K_1_501;
    z1 *= -K_0_899;
    z2 *= -K_2_562;
    z3 = z3 * -K_1_961 + z5;
    z4 = z4 * -K_0_390 + z5;
    d[7] = (o0 + z1 + z3 + RND) >> SH;
    d[5] = (o1 + z2 + z4 + RND) >> SH;
    d[3] = (o2 + z2 + z3 + RND) >> SH;
    d[1] = (o3 + z1 + z4 + RND) >> SH;
}

// diagnostics: thread 0 of a workgroup accumulates s_memtime deltas per phase
struct PhaseClock {
    unsigned long long* dst;
    unsigned long long last;
    __device__ __forceinline__ void start(unsigned long long* d) {
        dst = d;
        if (dst && threadIdx.x == 0) last = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void mark(int phase) {
        if (dst && threadIdx.x == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            atomicAdd(&dst[phase], now - last);
            last = now;
        }
    }
};

// wave-level ordering point for LDS traffic between lanes of the same wavefront
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// Quantiser.  The reference computes (int)round((double)n / (double)d)  (mdec.c:438): round half
// away from zero, i.e.  sgn(n) * floor((2|n| + d) / (2d)).   With N = 2|n| + d and D = 2d,
// floor(N / D) == floor((N + 0.5) / D), and (N + 0.5) / D is at least 0.5 / D away from every
// integer.  N < 2^17, so N + 0.5 is exact in fp32; one rcp (<= 1 ulp) and one multiply put the
// product within (N + 0.5) * 1.5 * 2^-23 / D of the true quotient, which is < 0.5 / D for
// N < 2.7e6.  Truncation therefore gives the exact floor.  `two_abs` = 2|n|, `d` = quant * scale,
// `inv2d` = 1 / (2d).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int quant_level(int two_abs, int d, float inv2d) {
    return (int)(((float)(two_abs + d) + 0.5f) * inv2d);
}

// DC: divisor is always 16 (mdec.c:671), clamp to [-512, 510] (mdec.c:260-267).
__device__ __forceinline__ int quant_dc(int c0) {
    const int a = c0 < 0 ? -c0 : c0;
    const int q = (a + 8) >> 4;
    const int v = c0 < 0 ? -q : q;
    return v < -512 ? -512 : (v > 510 ? 510 : v);
}

struct Lds {
    uint32_t* out;          // [out_words]           frame output staging, dword j = output bytes 4j..4j+3 (pre-swizzle)
    uint16_t* mb_bits;      // [nmb][kScalesPerPass] AC bits of each macroblock at each scale of the pass
    uint32_t* mb_off;       // [nmb]                 bit offset of each macroblock in the chosen bitstream
    int16_t* dcv;           // [nmb*6]               per block, encode order: v2 the quantised DC; v3 the quantised DC during
                            //                       loop (A), then the DPCM delta (codes are derived where needed)
    uint8_t* ac_len;        // [BS_LUT_SIZE]
    uint32_t* ac_code;      // [BS_LUT_SIZE]      bits << 24 | code
    uint8_t* dc_plen;       // [16]
    uint8_t* dc_prefix;     // [16]
    int16_t* tiles;         // per-wave DCT staging
    float2* qtab;           // [kScalesPerPass][64]  {1/(2 quant scale), 0.5 + 0.5/(2 quant scale)} for the current pass
    int* pass_bits;         // [kScalesPerPass] AC bits of the whole frame per scale
    int* scalars;           // [8]: 0 dc_bits, 1 chosen scale, 2 chosen index in pass, 3 nnz, 4 total bits
};

constexpr int kWaveTileBytes = ((6 * kTileStride * 2 + 6 * kZStride * 2) + 15) / 16 * 16;   // transpose tile + zig-zag tile (the pixel tile aliases the latter)

__host__ __device__ inline size_t lds_bytes(int nmb, int out_words, int waves) {
    size_t b = 0;
    b += (size_t)out_words * 4;
    b += (size_t)nmb * kScalesPerPass * 2;
    b = (b + 3) & ~(size_t)3;
    b += (size_t)nmb * 4;
    b += (size_t)nmb * 6 * 2;
    b = (b + 3) & ~(size_t)3;
    b += BS_LUT_SIZE;             // ac_len
    b = (b + 3) & ~(size_t)3;
    b += BS_LUT_SIZE * 4;         // ac_code
    b += 32;                      // dc tables
    b = (b + 15) & ~(size_t)15;
    b += (size_t)waves * kWaveTileBytes;
    b += (size_t)kScalesPerPass * 64 * 8;   // qtab
    b += 64;                      // pass_bits + scalars
    return b;
}

__device__ __forceinline__ Lds carve(char* base, int nmb, int out_words, int waves) {
    Lds L;
    size_t b = 0;
    L.out = (uint32_t*)(base + b);        b += (size_t)out_words * 4;
    L.mb_bits = (uint16_t*)(base + b);    b += (size_t)nmb * kScalesPerPass * 2;  b = (b + 3) & ~(size_t)3;
    L.mb_off = (uint32_t*)(base + b);     b += (size_t)nmb * 4;
    L.dcv = (int16_t*)(base + b);         b += (size_t)nmb * 6 * 2;               b = (b + 3) & ~(size_t)3;
    L.ac_len = (uint8_t*)(base + b);      b += BS_LUT_SIZE;                        b = (b + 3) & ~(size_t)3;
    L.ac_code = (uint32_t*)(base + b);    b += BS_LUT_SIZE * 4;
    L.dc_plen = (uint8_t*)(base + b);     b += 16;
    L.dc_prefix = (uint8_t*)(base + b);   b += 16;                                 b = (b + 15) & ~(size_t)15;
    L.tiles = (int16_t*)(base + b);       b += (size_t)waves * kWaveTileBytes;
    L.qtab = (float2*)(base + b);         b += (size_t)kScalesPerPass * 64 * 8;
    L.pass_bits = (int*)(base + b);       b += kScalesPerPass * 4;
    L.scalars = (int*)(base + b);
    return L;
}

// OR `len` bits (value `v`, MSB first) into the staging buffer at bit position `pos` of the
// bitstream.  Staging dword j holds stream bits [32j, 32j+32) with bit 32j in its MSB.
__device__ __forceinline__ void put_bits(uint32_t* words, uint32_t pos, int len, uint32_t v) {
    const uint32_t w = pos >> 5, sh = pos & 31;
    const uint64_t t = (uint64_t)v << (64 - sh - len);
    const uint32_t hi = (uint32_t)(t >> 32), lo = (uint32_t)t;
    atomicOr(&words[w], hi);
    if (lo) atomicOr(&words[w + 1], lo);
}

// Per-lane constants of the AC path: lane k owns zig-zag position k.
struct LaneConst {
    int quant;          // quant matrix entry at this zig-zag position
    uint64_t below;     // mask of lanes below this one
    int lane_m64;       // lane - 64
};

// One block at one scale, branch-free.  `two_abs_f` = float(2|n|), 0 on lane 0 (the DC slot never carries an
// AC code); `inv2d` = 1/(2 quant scale), `bias` = 0.5 + 0.5 * inv2d.
//   floor((2|n| + d) / 2d) with d / 2d = 0.5 folded into the addend: trunc(fma(2|n|, 1/2d, 0.5 + 0.5/2d)).
//   Same argument as quant_level(): the exact value is >= 0.5/2d away from every integer, the computed one is
//   within (2|n| + d) * 1.5 * 2^-23 / 2d + 2^-25 of it (tests/test_mdec_oracle.py checks every operand).
__device__ __forceinline__ int quant_mag(float two_abs_f, float inv2d, float bias) {
    return (int)__builtin_fmaf(two_abs_f, inv2d, bias);
}

// number of zero coefficients between this lane and the previous non-zero one (bit 0 of the mask, the
// DC slot, is the sentinel): lane - 1 - (63 - clz(mask below me))
__device__ __forceinline__ int run_before(uint64_t nz_mask, const LaneConst& lc) {
    const uint64_t prev = (nz_mask | 1ull) & lc.below;
    return __clzll((long long)prev) + lc.lane_m64;
}

// index into the padded LUTs: row = min(|level|, MAX_LEVEL + 1), column = run (0..62).  Level 0 is row 0
// = 0 bits, so silent lanes need no predicate.  Only the LENGTH needs no level clamp at 510/512
// (mdec.c:260-267): every level > MAX_LEVEL is an escape of the same length.
__device__ __forceinline__ int lut_index(int q, int run) {
    const unsigned qc = (unsigned)q > (unsigned)(BS_LUT_MAX_LEVEL + 1) ? (unsigned)(BS_LUT_MAX_LEVEL + 1) : (unsigned)q;
    return (int)__umul24(qc, (unsigned)BS_LUT_W) + run;     // full-rate 24-bit multiply (v_mul_lo_u32 is quarter rate)
}

// AC bits of one block at the four scales of a pass; two 16-bit counters per accumulator register.
// (the DC slot of every block is stored as 0, so lane 0 never produces a level)
// `live` (wave-uniform) has bit s set while scale s of the pass can still fit: like the reference, which stops an
// attempt at the first overflow (mdec.c:323-325,689-706), a scale whose running frame total already exceeds the budget
// is not evaluated any further -- its verdict cannot change.
// DC code of one block: v2 = the 10-bit value (mdec.c:451-453); v3 = VLC of the DPCM delta: size class = magnitude
// bits, then a sign-dependent offset (mdec.c:285-318).  `tab` = {plen[2][8], prefix[2][8]} in LDS.
template <int CODEC>
__device__ __forceinline__ void dc_code(int v, int luma, const uint8_t* plen, const uint8_t* prefix, int& len, uint32_t& code) {
    if (CODEC == 0) {
        len = 10;
        code = (uint32_t)v & 0x3FFu;
        return;
    }
    len = luma ? BS_DC_LUMA_ZERO_LEN : BS_DC_CHROMA_ZERO_LEN;
    code = luma ? BS_DC_LUMA_ZERO_CODE : BS_DC_CHROMA_ZERO_CODE;
    if (v != 0) {
        const int ad = v < 0 ? -v : v;
        const int mm = 31 - __builtin_clz((unsigned)ad);
        const uint32_t j = v > 0 ? (uint32_t)(v - (1 << mm)) : (uint32_t)(v + ((2 << mm) - 1));
        len = plen[luma * 8 + mm] + 1 + mm;
        code = ((uint32_t)prefix[luma * 8 + mm] << (mm + 1)) | ((v > 0 ? 1u : 0u) << mm) | j;
    }
}

template <int LIVE>
__device__ __forceinline__ void count_block4_live(float two_abs, const float2& k0, const float2& k1, const float2& k2,
                                                  const float2& k3, const LaneConst& lc, const uint8_t* ac_len, int& acc01,
                                                  int& acc23) {
    // straight-line code for one set of live scales: the compiler interleaves the independent chains
    int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    if (LIVE & 1) { const int q = quant_mag(two_abs, k0.x, k0.y); i0 = lut_index(q, run_before(wave::ballot(q != 0), lc)); }
    if (LIVE & 2) { const int q = quant_mag(two_abs, k1.x, k1.y); i1 = lut_index(q, run_before(wave::ballot(q != 0), lc)); }
    if (LIVE & 4) { const int q = quant_mag(two_abs, k2.x, k2.y); i2 = lut_index(q, run_before(wave::ballot(q != 0), lc)); }
    if (LIVE & 8) { const int q = quant_mag(two_abs, k3.x, k3.y); i3 = lut_index(q, run_before(wave::ballot(q != 0), lc)); }
    int a01 = 0, a23 = 0;
    if (LIVE & 1) a01 = (int)ac_len[i0];
    if (LIVE & 2) a01 |= (int)ac_len[i1] << 16;
    if (LIVE & 4) a23 = (int)ac_len[i2];
    if (LIVE & 8) a23 |= (int)ac_len[i3] << 16;
    if (LIVE & 3) acc01 += a01;
    if (LIVE & 12) acc23 += a23;
}

__device__ __forceinline__ void count_block4(int c, const float2& k0, const float2& k1, const float2& k2,
                                             const float2& k3, const LaneConst& lc, const uint8_t* ac_len, int live,
                                             int& acc01, int& acc23) {
    static_assert(kScalesPerPass == 4, "count_block4 evaluates 4 scales");
    const float two_abs = (float)(2 * (c < 0 ? -c : c));
    // bits(s) falls with s on ordinary material, so scales die lowest-first: those sets get straight-line code
    switch (live) {
    case 0xF: count_block4_live<0xF>(two_abs, k0, k1, k2, k3, lc, ac_len, acc01, acc23); break;
    case 0xE: count_block4_live<0xE>(two_abs, k0, k1, k2, k3, lc, ac_len, acc01, acc23); break;
    case 0xC: count_block4_live<0xC>(two_abs, k0, k1, k2, k3, lc, ac_len, acc01, acc23); break;
    case 0x8: count_block4_live<0x8>(two_abs, k0, k1, k2, k3, lc, ac_len, acc01, acc23); break;
    case 0x0: break;
    default:
        if (live & 1) count_block4_live<1>(two_abs, k0, k1, k2, k3, lc, ac_len, acc01, acc23);
        if (live & 2) count_block4_live<2>(two_abs, k0, k1, k2, k3, lc, ac_len, acc01, acc23);
        if (live & 4) count_block4_live<4>(two_abs, k0, k1, k2, k3, lc, ac_len, acc01, acc23);
        if (live & 8) count_block4_live<8>(two_abs, k0, k1, k2, k3, lc, ac_len, acc01, acc23);
        break;
    }
}

// which scales of the pass are still below the AC-bit limit (wave-uniform bit mask)
__device__ __forceinline__ int live_scales(const int4& totals, int limit_ac) {
    static_assert(kScalesPerPass == 4, "one int4 of running totals");
    const int p0 = __builtin_amdgcn_readfirstlane(totals.x), p1 = __builtin_amdgcn_readfirstlane(totals.y);
    const int p2 = __builtin_amdgcn_readfirstlane(totals.z), p3 = __builtin_amdgcn_readfirstlane(totals.w);
    return (p0 <= limit_ac ? 1 : 0) | (p1 <= limit_ac ? 2 : 0) | (p2 <= limit_ac ? 4 : 0) | (p3 <= limit_ac ? 8 : 0);
}

// per-macroblock sums of the packed counters -> LDS (per macroblock for the offset scan, per frame for rate control)
__device__ __forceinline__ void count_finish4(int acc01, int acc23, int lane, uint16_t* mb_bits_slot, int* pass_bits) {
    const int t01 = wave::reduce_add(acc01);
    const int t23 = wave::reduce_add(acc23);
    if (lane == 0) {
        uint2 v;
        v.x = (uint32_t)t01;
        v.y = (uint32_t)t23;
        *(uint2*)mb_bits_slot = v;
        atomicAdd(&pass_bits[0], t01 & 0xFFFF);
        atomicAdd(&pass_bits[1], (int)((unsigned)t01 >> 16));
        atomicAdd(&pass_bits[2], t23 & 0xFFFF);
        atomicAdd(&pass_bits[3], (int)((unsigned)t23 >> 16));
    }
}

__device__ __forceinline__ void fill_qtab(float2* qtab, int tid, int lane, int scale0) {
    // quantiser constants of a pass, once per workgroup (IEEE division, not per macroblock)
    if (tid < kScalesPerPass * 64) {
        const float r = 1.0f / (float)(2 * (int)c_quant_zz[lane] * (scale0 + (tid >> 6)));
        qtab[tid] = make_float2(r, 0.5f + 0.5f * r);
    }
}

// ---------------------------------------------------------------------------------------------
// Register budget: the kernel is compiled for 8 wavefronts per SIMD (<= 64 VGPRs), i.e. two frames in
// flight per CU.  The hot path is latency-bound (LDS look-ups, DPP scans, ballots), so it is written
// as short per-block bodies that rely on 8-way wave interleaving rather than on wide unrolled bodies
// that would need > 64 registers.  The three per-macroblock loops are: (A) DCT -> slab, (B) bit counts
// for kScalesPerPass scales from the slab, (C) emit at the chosen scale from the slab.
// ---------------------------------------------------------------------------------------------
template <int CODEC, int WAVES, int OCC>
__global__ __launch_bounds__(WAVES * 64, OCC) void mdec_encode_frames_kernel(const FrameJob job) {
    constexpr int kWavesPerGroup = WAVES;
    constexpr int kThreads = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Lds L = carve(smem, job.nmb, job.out_words, WAVES);

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int nmb = job.nmb, nx = job.nx, ny = job.ny, W = job.width, H = job.height;
    const int nblk = nmb * 6;

    PhaseClock clk;
    clk.start(job.timing);

    // ---- once per workgroup: LUTs into LDS, per-lane constants
    for (int i = tid; i < BS_LUT_SIZE; i += kThreads) {
        L.ac_len[i] = c_ac_len[i];
        L.ac_code[i] = c_ac_code[i];
    }
    if (tid < 16) {
        L.dc_plen[tid] = c_dc_plen[tid >> 3][tid & 7];
        L.dc_prefix[tid] = c_dc_prefix[tid >> 3][tid & 7];
    }
    LaneConst lc;
    lc.quant = c_quant_zz[lane];
    lc.below = (1ull << lane) - 1ull;
    lc.lane_m64 = lane - 64;

    // inverse zig-zag for the column-pass scatter: this lane handles column (lane & 7) of block (lane >> 3);
    // the eight scan positions are packed one byte each into two registers
    uint32_t zpos_lo = 0, zpos_hi = 0;
    {
        uint8_t* inv = (uint8_t*)L.tiles;   // temporary use before the tiles are live
        if (tid < 64) inv[c_zagzig[tid]] = (uint8_t)tid;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 4; v++) {
            zpos_lo |= (uint32_t)inv[v * 8 + (lane & 7)] << (8 * v);
            zpos_hi |= (uint32_t)inv[(v + 4) * 8 + (lane & 7)] << (8 * v);
        }
        __syncthreads();
    }

    int16_t* tileT = L.tiles + (size_t)wid * (kWaveTileBytes / 2);    // [6][kTileStride] row-pass output, transposed
    int16_t* tileZ = tileT + 6 * kTileStride;                          // [6][kZStride] zig-zag ordered coefficients

    int16_t* slab = job.coef_slab + (size_t)blockIdx.x * nmb * 384;

    clk.mark(7);   // per-workgroup prologue

    for (int f = (int)blockIdx.x; f < job.n_frames; f += (int)gridDim.x) {
        const uint8_t* frame = job.frames + (size_t)f * job.frame_stride;
        int max_size = job.max_sizes ? job.max_sizes[f] : job.uniform_max_size;
        // per-frame budgets live in device memory the host cannot vet: a budget outside [8, the context's maximum]
        // is treated as "nothing fits" (result quant_scale 64, no bytes written) instead of overrunning the staging
        const bool bad_budget = max_size < 8 || max_size > (job.out_words - 2) * 4;
        if (bad_budget) max_size = 8;
        const int max_words = (max_size + 3) >> 2;
        // a scale is hopeless once its AC bits alone exceed what the budget leaves after the cheapest possible
        // DC codes (v2: 10 bits, v3: >= 2 bits), the end-of-block codes and the end-of-frame code:
        // fits <=> 8 + 2*ceil(bits/16) <= max_size <=> bits <= 16 * floor((max_size - 8) / 2)
        const int limit_ac = 16 * ((max_size - 8) >> 1) - (nblk * ((CODEC == 0 ? 10 : 2) + 2) + 10);

        // ---- reset per-frame state
        for (int i = tid; i < max_words; i += kThreads) L.out[i] = 0u;
        if (tid < kScalesPerPass) L.pass_bits[tid] = 0;
        if (tid < 8) L.scalars[tid] = 0;
        fill_qtab(L.qtab, tid, lane, 1);
        __syncthreads();
        clk.mark(0);

        // =====================================================================================
        // (A) DCT of every macroblock -> coefficient slab (zig-zag order) + quantised DC, fused with the first
        //     count pass (scales 1..kScalesPerPass) while the coefficients are still in LDS
        // =====================================================================================
        {
            // Source bytes of a macroblock (mdec.c:619-633).  Lane t < 48 = (block t>>3, row t&7) fetches its own
            // 8 pixels straight from the frame (no LDS staging): a luma row is 8 contiguous bytes, a chroma row is
            // 16 bytes of interleaved Cr,Cb (NV21: Cr at even bytes, Cb at odd).  All offsets are 32-bit (a frame is
            // < 2^31 bytes); (fx, fy) advance incrementally, no divisions.
            const int blk = lane >> 3, r8 = lane & 7;
            const bool is_chroma = blk < 2;
            uint32_t lane_off;      // offset of this lane's pixel row inside macroblock (0, 0)
            if (is_chroma) lane_off = (uint32_t)W * (uint32_t)H + (uint32_t)r8 * (uint32_t)W;
            else lane_off = ((uint32_t)(((blk - 2) >> 1) * 8 + r8)) * (uint32_t)W + (uint32_t)((blk - 2) & 1) * 8u;
            if (lane >= 48) lane_off = 0;                       // idle lanes read the frame's first bytes (unused)
            const uint32_t hi_off = is_chroma ? 8u : 0u;        // chroma rows are 16 bytes long
            const uint32_t mb_row_step = (is_chroma ? 8u : 16u) * (uint32_t)W;
            const uint32_t perm_sel = blk == 0 ? 0x06040200u : 0x07050301u;   // even (Cr) / odd (Cb) bytes of a dword pair
            int fy = wid / nx, fx = wid - fy * nx;              // once per frame per wavefront
            uint2 plo = make_uint2(0, 0), phi = make_uint2(0, 0);
            if (wid < nmb) {
                const uint8_t* p = frame + (lane_off + (uint32_t)fy * mb_row_step + (uint32_t)fx * 16u);
                plo = *(const uint2*)p;
                phi = *(const uint2*)(p + hi_off);
            }
            for (int m = wid; m < nmb; m += kWavesPerGroup) {
                const int mbe = fx * ny + fy;   // encode order: fx outer, fy inner (mdec.c:689-690)
                // which scales are still in the race: one 16-byte LDS read, issued early (a slightly stale view only
                // means a dead scale is evaluated once more)
                const int4 totals = *(const int4*)L.pass_bits;

                // -- this lane's 8 pixels as two dwords
                uint2 px;
                px.x = is_chroma ? __builtin_amdgcn_perm(plo.y, plo.x, perm_sel) : plo.x;
                px.y = is_chroma ? __builtin_amdgcn_perm(phi.y, phi.x, perm_sel) : plo.y;
                // -- prefetch the next macroblock of this wavefront while this one is transformed
                {
                    fx += kWavesPerGroup;
                    while (fx >= nx) { fx -= nx; fy++; }
                    if (m + kWavesPerGroup < nmb) {
                        const uint8_t* p = frame + (lane_off + (uint32_t)fy * mb_row_step + (uint32_t)fx * 16u);
                        plo = *(const uint2*)p;
                        phi = *(const uint2*)(p + hi_off);
                    }
                }

                int d[8];
                if (lane < 48) {
                    // -- row pass: lane = (block, row)
                    d[0] = (int)(px.x & 0xFF) - 128;
                    d[1] = (int)((px.x >> 8) & 0xFF) - 128;
                    d[2] = (int)((px.x >> 16) & 0xFF) - 128;
                    d[3] = (int)(px.x >> 24) - 128;
                    d[4] = (int)(px.y & 0xFF) - 128;
                    d[5] = (int)((px.y >> 8) & 0xFF) - 128;
                    d[6] = (int)((px.y >> 16) & 0xFF) - 128;
                    d[7] = (int)(px.y >> 24) - 128;
                    fdct8<false>(d);
#pragma unroll
                    for (int c = 0; c < 8; c++) tileT[blk * kTileStride + c * 8 + r8] = (int16_t)d[c];
                }
                wave_sync();
                if (lane < 48) {
                    // -- column pass: lane = (block, column); 8 int16 of that column are contiguous
                    const int4 q = *(const int4*)&tileT[blk * kTileStride + r8 * 8];
                    d[0] = (int)(int16_t)(q.x & 0xFFFF);
                    d[1] = q.x >> 16;
                    d[2] = (int)(int16_t)(q.y & 0xFFFF);
                    d[3] = q.y >> 16;
                    d[4] = (int)(int16_t)(q.z & 0xFFFF);
                    d[5] = q.z >> 16;
                    d[6] = (int)(int16_t)(q.w & 0xFFFF);
                    d[7] = q.w >> 16;
                    fdct8<true>(d);
                    // -- column 0 holds the block's DC term in d[0]: quantise it here, and store 0 in its place so that
                    //    the AC path (here, in later count passes and in emit) sees "no coefficient" at scan position 0
                    if (r8 == 0) {
                        const int dc = quant_dc(d[0]);
                        L.dcv[mbe * 6 + blk] = (int16_t)dc;     // v3: raw value for the DPCM chain below
                        d[0] = 0;
                    }
#pragma unroll
                    for (int v = 0; v < 8; v++) {
                        const uint32_t zp = ((v < 4 ? zpos_lo : zpos_hi) >> (8 * (v & 3))) & 0xFFu;
                        tileZ[blk * kZStride + zp] = (int16_t)d[v];
                    }
                }
                wave_sync();
                // -- lane k owns zig-zag position k of each of the 6 blocks: to the slab, and counted at scales 1..4
                int16_t* dst = slab + ((unsigned)mbe * 384u + (unsigned)lane);
                {
                    const float2 k0 = L.qtab[lane], k1 = L.qtab[64 + lane], k2 = L.qtab[128 + lane], k3 = L.qtab[192 + lane];
                    int acc01 = 0, acc23 = 0;
                    const int live = live_scales(totals, limit_ac);
#pragma unroll 1
                    for (int b = 0; b < 6; b++) {
                        const int c = tileZ[b * kZStride + lane];
                        dst[b * 64] = (int16_t)c;
                        count_block4(c, k0, k1, k2, k3, lc, L.ac_len, live, acc01, acc23);
                    }
                    count_finish4(acc01, acc23, lane, &L.mb_bits[mbe * kScalesPerPass], L.pass_bits);
                }
                wave_sync();   // the zig-zag tile is rewritten by the next iteration
            }
        }
        __syncthreads();
        clk.mark(1);

        // =====================================================================================
        // DC: v2 = 10 bits per block; v3 = DPCM chain per component in encode order (mdec.c:454-479)
        // =====================================================================================
        if (CODEC == 0) {
            if (tid == 0) L.scalars[0] = 10 * nblk;
        } else if (wid < 3) {
            // wave 0: Cr chain, wave 1: Cb chain, wave 2: the Y chain (4 blocks per macroblock).
            // Element i maps last -> new_last:
            //   dc % 4 != 2 : constant 4*round(dc/4)                (last is always a multiple of 4)
            //   dc % 4 == 2 : last < dc ? dc + 2 : dc - 2           (tie, rounds away from zero)
            // Both are step functions (thr, lo, hi); composition g(f(x)) = (thr_f, g(lo_f), g(hi_f)),
            // so the chain is an inclusive scan under composition.
            const int count = wid == 2 ? 4 * nmb : nmb;
            int carry = 0, bits = 0;
            for (int base = 0; base < count; base += 64) {
                const int i = base + lane;
                const bool live = i < count;
                const int idx = wid == 2 ? ((i >> 2) * 6 + 2 + (i & 3)) : (i * 6 + wid);
                const int dc = live ? (int)L.dcv[idx] : 0;
                int thr, lo, hi;
                if ((dc & 3) == 2) {
                    thr = dc; lo = dc + 2; hi = dc - 2;
                } else {
                    const int a = dc < 0 ? -dc : dc;
                    const int rq = ((a + 2) >> 2) << 2;
                    thr = 0; lo = hi = dc < 0 ? -rq : rq;
                }
                if (!live) { thr = -100000; lo = hi = 0; }   // dead lanes sit after all live ones, never feed them
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int pthr = __shfl_up(thr, off, 64);
                    const int plo = __shfl_up(lo, off, 64);
                    const int phi = __shfl_up(hi, off, 64);
                    if (lane >= off) {
                        // me(prev(x)): apply my current (thr, lo, hi) to the predecessor's two outputs
                        const int nlo = plo < thr ? lo : hi;
                        const int nhi = phi < thr ? lo : hi;
                        thr = pthr; lo = nlo; hi = nhi;
                    }
                }
                const int cur = carry < thr ? lo : hi;            // last value after element i
                int prev = __shfl_up(cur, 1, 64);
                if (lane == 0) prev = carry;
                int delta = (cur - prev) >> 2;                      // exact: both multiples of 4
                if (CODEC == 2) {                                   // v3dc wrap (mdec.c:469-474)
                    if (delta < -0x80) delta += 0x100;
                    else if (delta > 0x80) delta -= 0x100;
                }
                int dlen;
                uint32_t dcode;
                dc_code<CODEC>(delta, wid == 2, L.dc_plen, L.dc_prefix, dlen, dcode);
                if (live) {
                    L.dcv[idx] = (int16_t)delta;
                    bits += dlen;
                }
                carry = __shfl(cur, 63, 64);
            }
            bits = wave::reduce_add(bits);
            if (lane == 0) atomicAdd(&L.scalars[0], bits);
        }
        __syncthreads();
        clk.mark(2);

        // =====================================================================================
        // (B) Rate control: first scale s with 8 + 2*ceil(bits(s)/16) <= max_size (mdec.c:663-723),
        //     kScalesPerPass scales per pass over the slab
        // =====================================================================================
        int scale0 = 1;   // first scale of the current pass; scales 1..kScalesPerPass were counted in (A)
        for (;;) {
            if (scale0 > 1) {
                fill_qtab(L.qtab, tid, lane, scale0);
                __syncthreads();
                for (int mbe = wid; mbe < nmb; mbe += kWavesPerGroup) {
                    const int16_t* src = slab + ((unsigned)mbe * 384u + (unsigned)lane);
                    const float2 k0 = L.qtab[lane], k1 = L.qtab[64 + lane], k2 = L.qtab[128 + lane], k3 = L.qtab[192 + lane];
                    int acc01 = 0, acc23 = 0;
                    const int live = live_scales(*(const int4*)L.pass_bits, limit_ac);
                    // (kept as a rolled loop with a one-block prefetch: unrolling it costs registers that the
                    // allocator takes from loop (A))
                    int cnext = src[0];
#pragma unroll 1
                    for (int b = 0; b < 6; b++) {
                        const int c = cnext;
                        if (b < 5) cnext = src[(b + 1) * 64];
                        count_block4(c, k0, k1, k2, k3, lc, L.ac_len, live, acc01, acc23);
                    }
                    count_finish4(acc01, acc23, lane, &L.mb_bits[mbe * kScalesPerPass], L.pass_bits);
                }
                __syncthreads();
            }

            if (tid == 0) {
                const int fixed = L.scalars[0] + 2 * nblk + 10;   // DC codes + end-of-block codes + end-of-frame code
                int chosen = 0;
                for (int s = 0; s < kScalesPerPass && scale0 + s < 64; s++) {
                    const int bits = L.pass_bits[s] + fixed;
                    if (8 + 2 * ((bits + 15) >> 4) <= max_size) {
                        chosen = scale0 + s;
                        L.scalars[2] = s;
                        L.scalars[4] = bits;
                        break;
                    }
                }
                L.scalars[1] = chosen;
            }
            __syncthreads();
            if (L.scalars[1] != 0 || scale0 + kScalesPerPass >= 64) break;
            __syncthreads();
            scale0 += kScalesPerPass;
            if (tid < kScalesPerPass) L.pass_bits[tid] = 0;
            __syncthreads();
        }
        clk.mark(3);

        const int scale = L.scalars[1];
        uint8_t* outp = job.out + (size_t)f * job.out_stride;

        if (scale == 0 || bad_budget) {
            // nothing fits (the reference asserts, mdec.c:723): zero output, flag the result
            if (!bad_budget)
                for (int i = tid; i < max_size; i += kThreads) outp[i] = 0;
            if (tid == 0) {
                psxhip_mdec_result_t r;
                r.quant_scale = 64; r.bytes_used = 0; r.blocks_used = 0; r.uncomp_hwords_used = 0;
                job.results[f] = r;
            }
            __syncthreads();
            continue;
        }
        const int sidx = L.scalars[2];

        // =====================================================================================
        // Bit offsets of the macroblocks: exclusive scan in encode order (wave 0)
        // =====================================================================================
        if (wid == 0) {
            uint32_t carry = 0;
            for (int base = 0; base < nmb; base += 64) {
                const int mbe = base + lane;
                int bits = 0;
                if (mbe < nmb) {
                    bits = L.mb_bits[mbe * kScalesPerPass + sidx] + 12;   // six end-of-block codes
#pragma unroll
                    for (int b = 0; b < 6; b++) {
                        int dlen;
                        uint32_t dcode;
                        dc_code<CODEC>((int)L.dcv[mbe * 6 + b], b >= 2, L.dc_plen, L.dc_prefix, dlen, dcode);
                        bits += dlen;
                    }
                }
                const int incl = wave::inclusive_scan_add(bits);
                if (mbe < nmb) L.mb_off[mbe] = carry + (uint32_t)(incl - bits);
                carry += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
            }
        }
        __syncthreads();
        clk.mark(4);

        // =====================================================================================
        // (C) Emit at the chosen scale.  At the accepted scale only a few of a block's 64 coefficients are
        // non-zero, so the expensive part (VLC look-up, bit positions, LDS writes) runs on a COMPACTED list:
        //   1. per block: quantise, ballot, and append {level, sign, scan position} of the non-zero lanes to a
        //      per-wavefront list in LDS.  Lane 0 always appends the block's DC slot, so the list is exactly
        //      the macroblock's code sequence: DC, AC..., DC, AC..., and the run before an AC coefficient is
        //      simply (its position - its predecessor's position - 1);
        //   2. per 64 list entries: look the codes up, prefix-sum their lengths, OR them into the frame image.
        // Each block's 2-bit end-of-block code "10" (mdec.c:501-503) travels as two extra leading bits of the
        // NEXT block's DC code; the frame's last one goes out with the end-of-frame code below.
        // =====================================================================================
        {
            const float inv1 = 1.0f / (float)(2 * lc.quant * scale), bias1 = 0.5f + 0.5f * inv1;
            uint32_t* stream = L.out + 2;                  // bitstream starts at byte 8 (mdec.c:686)
            uint32_t* clist = (uint32_t*)tileT;            // the DCT tiles are idle now: 384 entries fit (kWaveTileBytes >= 1536)
            const uint32_t lane_tag = (uint32_t)lane << 13;
            int nnz = 0;
            // the six coefficients of this lane are fetched one macroblock ahead (one memory latency per macroblock,
            // hidden behind the previous macroblock's work)
            int cn[6];
            if (wid < nmb) {
                const int16_t* src = slab + ((unsigned)wid * 384u + (unsigned)lane);
#pragma unroll
                for (int b = 0; b < 6; b++) cn[b] = src[b * 64];
            }
            for (int mbe = wid; mbe < nmb; mbe += kWavesPerGroup) {
                int cc[6];
#pragma unroll
                for (int b = 0; b < 6; b++) cc[b] = cn[b];
                if (mbe + kWavesPerGroup < nmb) {
                    const int16_t* src = slab + ((unsigned)(mbe + kWavesPerGroup) * 384u + (unsigned)lane);
#pragma unroll
                    for (int b = 0; b < 6; b++) cn[b] = src[b * 64];
                }
                // ---- 1. compaction.  Entry: [11:0] unclamped |level|, [12] sign, [18:13] scan position.
                //      Lane 0 (scan position 0) is always kept: it marks the block's DC slot.
                int count = 0;                             // wave-uniform
#pragma unroll
                for (int b = 0; b < 6; b++) {
                    const int c = cc[b];                       // scan position 0 holds 0 in the slab (the DC term lives in dcv)
                    const int q = quant_mag((float)(2 * (c < 0 ? -c : c)), inv1, bias1);     // <= 2048
                    const uint64_t m = wave::ballot(q != 0) | 1ull;
                    if (q != 0 || lane == 0)
                        clist[count + wave::popc_below(m)] = (uint32_t)q | (((uint32_t)c >> 19) & 0x1000u) | lane_tag;
                    const int n = (int)__builtin_popcountll(m);
                    count += n;
                    nnz += n - 1;
                }
                wave_sync();

                // ---- 2. codes
                uint32_t pos = L.mb_off[mbe] - (mbe > 0 ? 2u : 0u);   // the previous macroblock's last end-of-block code starts here
                int kcarry = 0, bcarry = 0;
                for (int base = 0; base < count; base += 64) {
                    const int i = base + lane;
                    const bool live = i < count;
                    const uint32_t e = live ? clist[i] : 0xFFFFFFFFu;      // dead lanes: scan position 63, never a DC slot
                    const int k = (int)((e >> 13) & 63u);
                    const bool neg = (e & 0x1000u) != 0;
                    const bool is_dc = k == 0;
                    int q = (int)(e & 0xFFFu);
                    const int lim = neg ? 512 : 510;               // level clamp, mdec.c:260-267
                    q = q > lim ? lim : q;
                    const int kprev = __builtin_amdgcn_update_dpp(kcarry, k, 0x138, 0xF, 0xF, false);   // wave_shr:1, lane 0 <- carry
                    const bool is_ac = live && !is_dc;
                    const int run = is_ac ? k - kprev - 1 : 0;
                    const uint32_t entry = L.ac_code[lut_index(is_ac ? q : 0, run)];
                    const int sl = neg ? -q : q;
                    const uint32_t esc = (1u << 16) | ((uint32_t)run << 10) | ((uint32_t)sl & 0x3FFu);   // mdec.c:258
                    int len = (int)(entry >> 24);
                    uint32_t code = len == BS_ESCAPE_BITS ? esc : ((entry & 0xFFFFFFu) | (neg ? 1u : 0u));
                    // DC slots: block index = number of DC slots before this one in the macroblock's list
                    const uint64_t dcmask = wave::ballot(is_dc);
                    if (is_dc) {
                        const int blk = bcarry + wave::popc_below(dcmask);
                        dc_code<CODEC>((int)L.dcv[mbe * 6 + blk], blk >= 2, L.dc_plen, L.dc_prefix, len, code);
                        if (blk > 0 || mbe > 0) {                   // carry the previous block's end-of-block code
                            code |= 2u << len;
                            len += 2;
                        }
                    }
                    const int incl = wave::inclusive_scan_add(len);
                    if (len) put_bits(stream, pos + (uint32_t)(incl - len), len, code);
                    pos += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
                    kcarry = __builtin_amdgcn_readlane(k, 63);
                    bcarry += (int)__builtin_popcountll(dcmask);
                }
                wave_sync();   // the list is rewritten by the next macroblock
            }
            if (lane == 0) atomicAdd(&L.scalars[3], nnz);
        }
        __syncthreads();
        clk.mark(5);

        // ---- end-of-frame code, header, results (mdec.c:710-754)
        const int total_bits = L.scalars[4];
        if (tid == 0) {
            // last block's end-of-block code + end-of-frame code (mdec.c:647-651,710)
            put_bits(L.out + 2, (uint32_t)(total_bits - 12), 12, (2u << 10) | (CODEC == 0 ? 0x1FFu : 0x3FFu));
            int hwords = L.scalars[3] + 2 * nblk + 2;
            hwords = (hwords + 0x3F) & ~0x3F;
            const int blocks_used = (hwords + 1) >> 1;
            int bytes_used = 8 + 2 * ((total_bits + 15) >> 4);
            bytes_used = (bytes_used + 3) & ~3;
            // header dwords are stored pre-swizzle like the rest: final dword = rotate16(staging)
            const uint32_t h0 = ((uint32_t)blocks_used & 0xFFFFu) | (0x3800u << 16);
            const uint32_t h1 = ((uint32_t)scale & 0xFFFFu) | ((CODEC == 0 ? 2u : 3u) << 16);
            L.out[0] = (h0 >> 16) | (h0 << 16);
            L.out[1] = (h1 >> 16) | (h1 << 16);
            psxhip_mdec_result_t r;
            r.quant_scale = scale; r.bytes_used = bytes_used; r.blocks_used = blocks_used; r.uncomp_hwords_used = hwords;
            job.results[f] = r;
        }
        __syncthreads();

        // ---- write-out: staging dword holds two MSB-first 16-bit words; each word is stored low byte
        //      first (mdec.c:321-333), i.e. the output dword is the staging dword rotated by 16.
        {
            const int full = max_size >> 2;
            uint32_t* o32 = (uint32_t*)outp;
            for (int i = tid; i < full; i += kThreads) {
                const uint32_t v = L.out[i];
                o32[i] = (v >> 16) | (v << 16);
            }
            const int tail = max_size & 3;
            if (tid < tail) {
                const uint32_t v = L.out[full];
                const uint32_t o = (v >> 16) | (v << 16);
                outp[full * 4 + tid] = (uint8_t)(o >> (8 * tid));
            }
        }
        __syncthreads();
        clk.mark(6);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Host side of the kernel (called from psxhip_mdec.cpp through psxhip_internal.h)
// ---------------------------------------------------------------------------------------------
extern "C" size_t psxhip_mdec_lds_bytes(int nmb, int out_words, int large) {
    return lds_bytes(nmb, out_words, large ? kWavesLarge : kWavesSmall);
}
extern "C" size_t psxhip_mdec_slab_bytes_per_group(int nmb) { return (size_t)nmb * 384 * sizeof(int16_t); }
extern "C" int psxhip_mdec_threads_per_group(int large) { return (large ? kWavesLarge : kWavesSmall) * 64; }

extern "C" hipError_t psxhip_mdec_upload_tables(void) {
    hipError_t e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_ac_len), bs_ac_len_lut, sizeof(bs_ac_len_lut))) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_ac_code), bs_ac_code_lut, sizeof(bs_ac_code_lut))) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_zagzig), bs_zagzig, sizeof(bs_zagzig))) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_quant_zz), bs_quant_zz, sizeof(bs_quant_zz))) != hipSuccess) return e;
    uint8_t pre[2][8], pl[2][8];
    for (int i = 0; i < 8; i++) {
        pre[0][i] = bs_dc_chroma_prefix[i]; pl[0][i] = bs_dc_chroma_plen[i];
        pre[1][i] = bs_dc_luma_prefix[i];   pl[1][i] = bs_dc_luma_plen[i];
    }
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_dc_prefix), pre, sizeof(pre))) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_dc_plen), pl, sizeof(pl))) != hipSuccess) return e;
    return hipSuccess;
}

extern "C" hipError_t psxhip_mdec_launch(const psxhip_mdec_launch_t* a) {
    FrameJob job;
    job.frames = a->d_frames;
    job.frame_stride = a->frame_stride;
    job.width = a->width;
    job.height = a->height;
    job.nx = a->width / 16;
    job.ny = a->height / 16;
    job.nmb = job.nx * job.ny;
    job.n_frames = a->n_frames;
    job.max_sizes = a->d_max_sizes;
    job.uniform_max_size = a->uniform_max_size;
    job.out = a->d_out;
    job.out_stride = a->out_stride;
    job.results = a->d_results;
    job.coef_slab = a->d_coef_slab;
    job.out_words = a->out_words;
    job.timing = a->d_timing;
    const int waves = a->large ? kWavesLarge : kWavesSmall;
    const size_t lds = lds_bytes(job.nmb, job.out_words, waves);
    const dim3 grid((unsigned)a->grid), block((unsigned)waves * 64u);
    hipStream_t st = (hipStream_t)a->stream;
#define PSX_LAUNCH(CODEC)                                                                                             \
    do {                                                                                                              \
        if (a->large) hipLaunchKernelGGL((mdec_encode_frames_kernel<CODEC, kWavesLarge, kOccLarge>), grid, block, lds, st, job); \
        else hipLaunchKernelGGL((mdec_encode_frames_kernel<CODEC, kWavesSmall, kOccSmall>), grid, block, lds, st, job);          \
    } while (0)
    switch (a->codec) {
    case 0: PSX_LAUNCH(0); break;
    case 1: PSX_LAUNCH(1); break;
    default: PSX_LAUNCH(2); break;
    }
#undef PSX_LAUNCH
    return hipGetLastError();
}

extern "C" hipError_t psxhip_mdec_set_max_lds(int codec, size_t bytes) {
    hipError_t e = hipSuccess;
#define PSX_ATTR(CODEC)                                                                                                         \
    do {                                                                                                                        \
        e = hipFuncSetAttribute((const void*)mdec_encode_frames_kernel<CODEC, kWavesSmall, kOccSmall>,                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                                        \
        if (e == hipSuccess)                                                                                                    \
            e = hipFuncSetAttribute((const void*)mdec_encode_frames_kernel<CODEC, kWavesLarge, kOccLarge>,                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                                    \
    } while (0)
    switch (codec) {
    case 0: PSX_ATTR(0); break;
    case 1: PSX_ATTR(1); break;
    default: PSX_ATTR(2); break;
    }
#undef PSX_ATTR
    return e;
}
