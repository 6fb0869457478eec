// wave_ops.h -- wave64 cross-lane primitives for gfx950 (CDNA4).
//
// DPP (data-parallel primitive) forms: no LDS traffic, one VALU op per step.  gfx9-family
// DPP controls used here: row_shr:n (0x110+n), row_bcast:15 (0x142), row_bcast:31 (0x143),
// quad_perm (0x00..0xFF), row_half_mirror (0x141), row_mirror (0x140).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wave {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_or_zero(int x) {
    // lanes whose source is out of range or masked read 0: with every row and bank enabled the hardware's own zero fill
    // (bound_ctrl) does it and no register has to hold the 0; masked forms keep old = 0
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, BANK_MASK, ROW_MASK == 0xF && BANK_MASK == 0xF);
}

// Inclusive prefix sum over the 64 lanes.
__device__ __forceinline__ int inclusive_scan_add(int x) {
    int s = x;
    s += dpp_or_zero<0x111, 0xF, 0xF>(x);   // row_shr:1
    s += dpp_or_zero<0x112, 0xF, 0xF>(x);   // row_shr:2
    s += dpp_or_zero<0x113, 0xF, 0xF>(x);   // row_shr:3   -> each lane: sum of itself and up to 3 left neighbours in its row
    s += dpp_or_zero<0x114, 0xF, 0xE>(s);   // row_shr:4, banks 1..3
    s += dpp_or_zero<0x118, 0xF, 0xC>(s);   // row_shr:8, banks 2..3  -> inclusive scan inside each 16-lane row
    s += dpp_or_zero<0x142, 0xA, 0xF>(s);   // row_bcast:15 into rows 1 and 3
    s += dpp_or_zero<0x143, 0xC, 0xF>(s);   // row_bcast:31 into rows 2 and 3
    return s;
}

// Sum over the 64 lanes, returned wave-uniform.
__device__ __forceinline__ int reduce_add(int x) {
    int s = x;
    s += dpp_or_zero<0xB1, 0xF, 0xF>(s);    // quad_perm [1,0,3,2]
    s += dpp_or_zero<0x4E, 0xF, 0xF>(s);    // quad_perm [2,3,0,1]
    s += dpp_or_zero<0x141, 0xF, 0xF>(s);   // row_half_mirror
    s += dpp_or_zero<0x140, 0xF, 0xF>(s);   // row_mirror          -> every lane holds its row's sum
    s += dpp_or_zero<0x142, 0xA, 0xF>(s);   // row_bcast:15
    s += dpp_or_zero<0x143, 0xC, 0xF>(s);   // row_bcast:31        -> lane 63 holds the total
    return __builtin_amdgcn_readlane(s, 63);
}

// Number of set bits of a wave-uniform 64-bit mask strictly below this lane.
__device__ __forceinline__ int popc_below(uint64_t mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// (the intrinsic, not __ballot(): it folds into the compare that produced `p` instead of materialising 0 / 1 first)
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

template <typename T>
__device__ __forceinline__ T broadcast_first(T v) {
    static_assert(sizeof(T) == 4, "32-bit only");
    int i = __builtin_amdgcn_readfirstlane(*reinterpret_cast<int*>(&v));
    return *reinterpret_cast<T*>(&i);
}

// 64-bit lexicographic minimum over the wave (used by the ADPCM candidate search).
__device__ __forceinline__ uint64_t reduce_min_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

}  // namespace wave
