// tools/microbench/dct_rowpass_mfma.hip -- go / no-go experiment: the 8-point row pass of the frame
// kernel's FDCT evaluated on the matrix pipe (v_mfma_i32_32x32x16_i8, exact in int32) against the v_dot2_i32_i16 form the
// kernel uses (mdec_kernels.hip, fdct8_pk<false>).
//
// Both variants start from a lane's 8 raw pixels (two dwords, what the frame kernel's fetch leaves in a lane) and end with
// the row pass's eight outputs in registers; operand preparation is included (the byte permutes of the dot2 form, the bias
// xor of the MFMA form), the LDS transpose that follows in the kernel is not (it is the same 8 x ds_write_b16 per lane in
// both).  One "unit" = 64 row vectors = 1 1/3 macroblocks.
//
//   MFMA form: A (32 x 16) = the pass's 8 x 8 linear forms as two balanced int8 digits (c = 256 h + l), block-diagonal over
//   two k-groups: rows 0-7 / 8-15 = l / h digits acting on k 0-7, rows 16-23 / 24-31 on k 8-15.  B (16 x 32): lane l holds
//   the 8 pixels (biased by -128: the row sums of all forms but out0 are 0, and out0's 16 * 8 * 128 is exactly the level
//   shift) of vector l % 32 in k-group l / 32.  D: lane n < 32 gets outputs 0-3 of vectors n and 32 + n, lane 32 + n
//   outputs 4-7, both digits of an output in the same lane: out = ((h << 8) + l + rnd) >> 9 (outputs 0 and 4: no shift).
//
// Build and run on an MI355X:  hipcc --offload-arch=gfx950 -O3 -o dct_mb dct_rowpass_mfma.hip && ./dct_mb
// Prints: results identical or not, ns per unit per wavefront-slot for each form at 6 wavefronts per SIMD (the frame kernel's
// occupancy), and the same with the two forms' wavefronts mixed 1:1 on every SIMD (does the matrix pipe run beside VALU work?).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int K_0_298 = 2446, K_0_390 = 3196, K_0_541 = 4433, K_0_765 = 6270, K_0_899 = 7373, K_1_175 = 9633, K_1_501 = 12299,
              K_1_847 = 15137, K_1_961 = 16069, K_2_053 = 16819, K_2_562 = 20995, K_3_072 = 25172;
constexpr int A_ = K_0_541 + K_0_765, B_ = K_0_541, C_ = K_0_541 - K_1_847;
constexpr int C7_0 = K_0_298 - K_0_899 - K_1_961 + K_1_175, C7_1 = K_1_175, C7_2 = K_1_175 - K_1_961, C7_3 = K_1_175 - K_0_899;
constexpr int C5_0 = K_1_175, C5_1 = K_2_053 - K_2_562 - K_0_390 + K_1_175, C5_2 = K_1_175 - K_2_562, C5_3 = K_1_175 - K_0_390;
constexpr int C3_0 = K_1_175 - K_1_961, C3_1 = K_1_175 - K_2_562, C3_2 = K_3_072 - K_2_562 - K_1_961 + K_1_175, C3_3 = K_1_175;
constexpr int C1_0 = K_1_175 - K_0_899, C1_1 = K_1_175 - K_0_390, C1_2 = K_1_175, C1_3 = K_1_501 - K_0_899 - K_0_390 + K_1_175;

// the row pass as an 8 x 8 integer matrix over the pixels d0..d7 (o0 = d3-d4, o1 = d2-d5, o2 = d1-d6, o3 = d0-d7)
static void row_matrix(int M[8][8]) {
    const int e0[8] = {16, 16, 16, 16, 16, 16, 16, 16};
    const int e4[8] = {16, -16, -16, 16, 16, -16, -16, 16};
    const int e2[8] = {A_, B_, -B_, -A_, -A_, -B_, B_, A_};
    const int e6[8] = {B_, C_, -C_, -B_, -B_, -C_, C_, B_};
    auto odd = [](int c3, int c2, int c1, int c0, int* r) {
        r[0] = c3; r[7] = -c3; r[1] = c2; r[6] = -c2; r[2] = c1; r[5] = -c1; r[3] = c0; r[4] = -c0;
    };
    for (int j = 0; j < 8; j++) { M[0][j] = e0[j]; M[4][j] = e4[j]; M[2][j] = e2[j]; M[6][j] = e6[j]; }
    odd(C7_3, C7_2, C7_1, C7_0, M[7]);
    odd(C5_3, C5_2, C5_1, C5_0, M[5]);
    odd(C3_3, C3_2, C3_1, C3_0, M[3]);
    odd(C1_3, C1_2, C1_1, C1_0, M[1]);
}

__device__ __forceinline__ int dot2_k(uint32_t x, uint32_t k, int acc) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(k), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b))); }
__host__ __device__ constexpr uint32_t pk(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }

// the frame kernel's form (luma lane: 8 contiguous pixel bytes in (lo, hi))
__device__ __forceinline__ void rowpass_dot2(uint32_t lo, uint32_t hi, int (&d)[8]) {
    const uint32_t P0 = __builtin_amdgcn_perm(hi, lo, 0x0C010C00u), P1 = __builtin_amdgcn_perm(hi, lo, 0x0C030C02u);
    const uint32_t R0 = __builtin_amdgcn_perm(hi, lo, 0x0C060C07u), R1 = __builtin_amdgcn_perm(hi, lo, 0x0C040C05u);
    const uint32_t S0 = pk_add(P0, R0), S1 = pk_add(P1, R1), D0 = pk_sub(P0, R0), D1 = pk_sub(P1, R1);
    const int rnd = 1 << 8;
    d[0] = dot2_k(S0, pk(1, 1), dot2_k(S1, pk(1, 1), 0)) * 16 - 8 * 128 * 16;
    d[4] = dot2_k(S0, pk(1, -1), dot2_k(S1, pk(-1, 1), 0)) * 16;
    d[2] = dot2_k(S0, pk(A_, B_), dot2_k(S1, pk(-B_, -A_), rnd)) >> 9;
    d[6] = dot2_k(S0, pk(B_, C_), dot2_k(S1, pk(-C_, -B_), rnd)) >> 9;
    d[7] = dot2_k(D0, pk(C7_3, C7_2), dot2_k(D1, pk(C7_1, C7_0), rnd)) >> 9;
    d[5] = dot2_k(D0, pk(C5_3, C5_2), dot2_k(D1, pk(C5_1, C5_0), rnd)) >> 9;
    d[3] = dot2_k(D0, pk(C3_3, C3_2), dot2_k(D1, pk(C3_1, C3_0), rnd)) >> 9;
    d[1] = dot2_k(D0, pk(C1_3, C1_2), dot2_k(D1, pk(C1_1, C1_0), rnd)) >> 9;
}

// MFMA form: a = this lane's 8 bytes of the digit matrix, cinit = the per-row rounding constants in D's layout
__device__ __forceinline__ void rowpass_mfma(uint32_t lo, uint32_t hi, long a, const i32x16& cinit, int (&o)[8]) {
    const uint32_t blo = lo ^ 0x80808080u, bhi = hi ^ 0x80808080u;         // pixels - 128 as int8
    const long b = (long)(((unsigned long long)bhi << 32) | blo);
    const i32x16 acc = __builtin_amdgcn_mfma_i32_32x32x16_i8(a, b, cinit, 0, 0, 0);
    // regs 0-3: l digits, 4-7: h digits of this lane's four outputs for vector n; 8-11 / 12-15 the same for vector 32 + n
#pragma unroll
    for (int v = 0; v < 2; v++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int x = (acc[8 * v + 4 + r] << 8) + acc[8 * v + r];
            o[4 * v + r] = r == 0 ? x : x >> 9;          // outputs 0 and 4 carry no fraction
        }
}

template <int FORM>
__global__ __launch_bounds__(768, 6) void bench(const uint2* pix, const long* amat, int iters, int* sink, int mix) {
    const int tid = (int)(blockIdx.x * blockDim.x + threadIdx.x), lane = (int)(threadIdx.x & 63);
    uint2 p = pix[tid];
    const long a = amat[lane];
    i32x16 cinit;
#pragma unroll
    for (int r = 0; r < 16; r++) cinit[r] = ((r & 4) == 0 && (r & 3) != 0) ? 256 : 0;      // rnd joins the l digit of the shifted rows
    const int form = FORM == 2 ? ((int)(threadIdx.x >> 6) & 1) : FORM;     // 2: odd wavefronts MFMA, even dot2 (wave-uniform)
    int acc = 0;
    if (form == 0) {
        for (int it = 0; it < iters; it++) {
            int d[8];
            rowpass_dot2(p.x, p.y, d);
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= d[i];
            p.x += (uint32_t)acc & 0x01010101u;            // the next iteration depends on this one, cheaply
        }
    } else {
        for (int it = 0; it < iters; it++) {
            int o[8];
            rowpass_mfma(p.x, p.y, a, cinit, o);
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= o[i];
            p.x += (uint32_t)acc & 0x01010101u;
        }
    }
    (void)mix;
    if (acc == 0x7FFFFFFF) sink[0] = acc;                  // keeps the loop alive
}

// one evaluation of each form, results in canonical [vector][output] order
__global__ void check(const uint2* pix, const long* amat, int* out_dot2, int* out_mfma) {
    const int lane = (int)threadIdx.x;
    const uint2 p = pix[lane];
    int d[8], o[8];
    rowpass_dot2(p.x, p.y, d);
    i32x16 cinit;
#pragma unroll
    for (int r = 0; r < 16; r++) cinit[r] = ((r & 4) == 0 && (r & 3) != 0) ? 256 : 0;
    rowpass_mfma(p.x, p.y, amat[lane], cinit, o);
    for (int i = 0; i < 8; i++) out_dot2[lane * 8 + i] = d[i];
    const int n = lane & 31, half = lane >> 5;
    for (int v = 0; v < 2; v++)
        for (int r = 0; r < 4; r++) out_mfma[(32 * v + n) * 8 + 4 * half + r] = o[4 * v + r];
}

int main() {
    int M[8][8];
    row_matrix(M);
    // A (32 x 16): lane l holds row l % 32, k = 8 * (l / 32) .. + 7
    std::vector<long> amat(64);
    for (int l = 0; l < 64; l++) {
        const int row = l % 32, kg = l / 32;
        unsigned long long w = 0;
        const int grp = row / 16, digit = (row / 8) & 1, outp = row & 7;       // rows 0-15 act on k-group 0, 16-31 on k-group 1
        for (int k = 0; k < 8; k++) {
            int v = 0;
            if (grp == kg) {
                const int c = M[outp][k];
                const int lo = ((c + 128) & 255) - 128, hi = (c - lo) >> 8;
                if (hi < -128 || hi > 127) { printf("coefficient %d does not split\n", c); return 2; }
                v = digit ? hi : lo;
            }
            w |= (unsigned long long)(uint8_t)(int8_t)v << (8 * k);
        }
        amat[l] = (long)w;
    }
    const int waves_per_cu = 24, cus = 256, threads = cus * waves_per_cu * 64;
    std::vector<uint2> pix(threads);
    uint32_t s = 12345;
    for (auto& p : pix) { s = s * 1664525u + 1013904223u; p.x = s; s = s * 1664525u + 1013904223u; p.y = s; }
    uint2* d_pix; long* d_a; int *d_sink, *d_o0, *d_o1;
    hipMalloc(&d_pix, pix.size() * sizeof(uint2)); hipMalloc(&d_a, 64 * sizeof(long)); hipMalloc(&d_sink, 4);
    hipMalloc(&d_o0, 512 * 4); hipMalloc(&d_o1, 512 * 4);
    hipMemcpy(d_pix, pix.data(), pix.size() * sizeof(uint2), hipMemcpyHostToDevice);
    hipMemcpy(d_a, amat.data(), 64 * sizeof(long), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d_pix, d_a, d_o0, d_o1);
    std::vector<int> o0(512), o1(512);
    hipMemcpy(o0.data(), d_o0, 512 * 4, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d_o1, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 512; i++) bad += o0[i] != o1[i];
    // ... and both against the matrix itself on the host
    int bad_ref = 0;
    for (int v = 0; v < 64; v++) {
        const uint64_t raw = ((uint64_t)pix[v].y << 32) | pix[v].x;
        for (int i = 0; i < 8; i++) {
            long acc = 0;
            for (int k = 0; k < 8; k++) acc += (long)M[i][k] * (int)((raw >> (8 * k)) & 255);
            const int want = (i == 0) ? (int)(acc - 16384) : (i == 4 ? (int)acc : (int)((acc + 256) >> 9));
            bad_ref += want != o0[v * 8 + i];
        }
    }
    printf("row pass, 64 vectors: MFMA form vs dot2 form: %d of 512 outputs differ; dot2 form vs the host matrix: %d differ\n", bad, bad_ref);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"dot2 form (all wavefronts)", "MFMA form (all wavefronts)", "mixed 1:1 (odd wavefronts MFMA, even dot2)"};
    float ms[3];
    for (int form = 0; form < 3; form++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (form == 0) hipLaunchKernelGGL(bench<0>, dim3(cus * 2), dim3(768), 0, 0, d_pix, d_a, iters, d_sink, 0);
            else if (form == 1) hipLaunchKernelGGL(bench<1>, dim3(cus * 2), dim3(768), 0, 0, d_pix, d_a, iters, d_sink, 0);
            else hipLaunchKernelGGL(bench<2>, dim3(cus * 2), dim3(768), 0, 0, d_pix, d_a, iters, d_sink, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[form], e0, e1);
        }
        // 24 wavefronts per CU = 6 per SIMD; a SIMD works off 6 * iters units in ms
        printf("%-46s %8.3f ms for %d units per wavefront: %.2f ns per unit per SIMD (= per 64 row vectors)\n", names[form], ms[form], iters,
               ms[form] * 1e6 / (6.0 * iters));
    }
    printf("per macroblock (48 row vectors, one pass): dot2 %.2f ns, MFMA %.2f ns at 25%% idle columns / %.2f ns packed 4 macroblocks to 3 MFMAs\n",
           ms[0] * 1e6 / (6.0 * iters), ms[1] * 1e6 / (6.0 * iters), 0.75 * ms[1] * 1e6 / (6.0 * iters));
    return bad || bad_ref;
}
