// frontend_kernels.hip -- colour conversion + scaling front-end for MI355X (gfx950): RGB24 / YUV420P pictures of any size
// -> NV21 frames of the encoder's size, BT.601 full range, written straight into the buffer the MDEC kernel reads
// (include/psxav_hip.h, psxhip_scaler_*).
//
// The reference hands this step to FFmpeg's libswscale (psxavenc/decoding.c:287-311: sws_getContext(... AV_PIX_FMT_NV21,
// SWS_BICUBIC ...) + sws_setColorspaceDetails(dst = ITU-R BT.601, full range); :463-475: sws_scale into the frame buffer).
// libswscale is not part of the reference tree and not installed here, so there is nothing to be bit-exact TO: the
// arithmetic below is this library's own ("psxhip front-end v1", specified in DESIGN.md section 9 and restated
// independently in oracle/frontend_oracle.c, which the tests hold this kernel to bit for bit).  It follows libswscale's
// structure -- bicubic B = 0 / C = 0.6, taps widened when shrinking, 14-bit coefficients, a horizontal pass into 15-bit
// intermediates, range expansion on the intermediates, a vertical pass -- so results agree with it to within the rounding
// of those formats, but PARITY WITH THE REFERENCE'S SCALER IS UNPINNED and stays so until someone runs both side by side.
//
// Why it exists: the frame encoder consumes 750 GB/s of NV21 at its headline rate, a PCIe link delivers 50 GB/s.  Pictures
// that are decoded on the device (or uploaded once at source size) have to become encoder input without leaving HBM.
//
// Mapping: HBM-bound streaming work, no matrix shape in it.  One workgroup = one BAND of the output -- 64 luma columns (+ the
// 32 chroma pairs under them) -- of one frame, walked top to bottom in tiles of 16 rows; grid = bands x frames x vertical
// segments (>> 256 workgroups for any batch; a lone picture is cut into segments so that the chip still fills).  Per tile, the
// source rows its filters reach AND THE PREVIOUS TILE DID NOT are read, coalesced, converted (RGB -> Y, Cb, Cr) and parked in
// LDS as bytes; the horizontal pass runs LDS -> LDS into a RING of int16 rows that outlives the tile (a source row is
// converted and filtered once per band, not once per tile whose vertical taps touch it: at 2:1 that was 1.5x the rows), the
// vertical pass ring -> registers, and the tile leaves as dword stores (four luma bytes, or two interleaved Cr,Cb pairs, per
// lane).  The band's horizontal taps sit in LDS for the whole walk, the tile's vertical taps are reloaded per tile.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "psxhip_internal.h"

int psxhip_ensure_device(int device);

namespace {

struct Bank {                 // one separable filter bank on the device
    const int32_t* left;      // [n]
    const int16_t* coef;      // [n * taps]
    const uint32_t* digits;   // [n * 2 * taps4] horizontal banks: the taps as two balanced int8 digits (c = 256 h + l), four taps
                              //                 to a dword, zero-padded to taps4 dwords: first the l dwords, then the h dwords
    int taps, taps4;
};

struct ScalerJob {
    const uint8_t* src;
    size_t src_stride;
    uint8_t* out;
    size_t frame_stride;
    int sw, sh, dw, dh;
    int limited;              // YUV input in MPEG range: expand on the intermediates
    Bank lh, lv, ch, cv;      // luma / chroma, horizontal / vertical
    int csw, csh;             // chroma source plane size (sw/2 x sh/2 for YUV420P, sw x sh for RGB)
    int TW, TH;               // luma tile; the chroma tile is TW/2 x TH/2
    int tiles_x, tiles_y;
    int vsegs;                // vertical segments per band (blockIdx.z): each re-reads the rows its first tile reaches
    int reg_rows, reg_cols;   // LDS region capacity per plane (RGB: the union of the luma and chroma reach; YUV: luma)
    int creg_rows, creg_cols; // ... of a chroma plane (YUV)
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// out8(): (acc >> 21) clamped to 0..255.  The value passes through an empty asm statement on purpose: hipcc 7.2 (LLVM 22git)
// fuses "shift right, saturate to u8, pack two" into gfx950's v_ashr_pk_u8_i32 and then ORs the other two bytes of the dword
// into the same register -- but the instruction leaves the destination's upper 16 bits as they were (the old accumulator), so
// bytes 2 and 3 of every packed store came out as (right value | leftover bits).  Found by tests/test_gpu_frontend.py (first
// on an enlarging geometry where only three rows per tile hit it, then on all rows once register allocation shifted).  No other
// kernel of the library contains the instruction.
__device__ __forceinline__ uint32_t out8(int acc) {
    int v = acc >> 21;
    asm volatile("" : "+v"(v));
    return (uint32_t)clampi(v, 0, 255);
}

// (q, r) = divmod(start + k * step, d) for k = 0, 1, 2, ...: one division when the walk starts, a compare per step after that
// (every loop of the kernel walks item = tid, tid + 256, ... over a rows x columns grid whose width is a run-time value;
// a division per item cost more than the item's own work)
struct DivWalk {
    int q, r, dq, dr, d;
    __device__ __forceinline__ DivWalk(int start, int step, int d_) : d(d_) {
        q = start / d_; r = start - q * d_;
        dq = step / d_; dr = step - dq * d_;
    }
    __device__ __forceinline__ void next() {
        q += dq; r += dr;
        if (r >= d) { r -= d; q++; }
    }
};

template <int FMT>      // 0 = RGB24, 1 = YUV420P
__global__ __launch_bounds__(256) void scaler_kernel(const ScalerJob job) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = (int)threadIdx.x;
    const int tx = (int)blockIdx.x;
    const uint8_t* src = job.src + (size_t)blockIdx.y * job.src_stride;
    uint8_t* out = job.out + (size_t)blockIdx.y * job.frame_stride;
    const int TW = job.TW, TH = job.TH, CW = TW >> 1, CH = TH >> 1;
    // this segment's tiles
    const int per_seg = (job.tiles_y + job.vsegs - 1) / job.vsegs;
    const int ty0 = (int)blockIdx.z * per_seg, ty1 = min(job.tiles_y, ty0 + per_seg);
    if (ty0 >= ty1) return;

    // ---- the band's output columns and the source columns their taps reach
    const int X0 = tx * TW, X1 = min(job.dw, X0 + TW);
    const int tw = X1 - X0, cw = tw >> 1;
    const int cX0 = X0 >> 1;
    const int lxa = job.lh.left[X0], lxb = job.lh.left[X1 - 1] + job.lh.taps;
    const int cxa = job.ch.left[cX0], cxb = job.ch.left[cX0 + cw - 1] + job.ch.taps;

    // ---- LDS: three byte planes (the rows a tile adds), the two rings of intermediates, the taps
    const int plane_bytes = (job.reg_rows * job.reg_cols + 8 + 15) & ~15;
    const int cplane_bytes = FMT == 0 ? plane_bytes : ((job.creg_rows * job.creg_cols + 8 + 15) & ~15);
    uint8_t* p0 = (uint8_t*)smem;
    uint8_t* p1 = p0 + plane_bytes;
    uint8_t* p2 = p1 + cplane_bytes;
    const int ringL = job.reg_rows, ringC = FMT == 0 ? job.reg_rows : job.creg_rows;      // rows the rings hold
    int16_t* tmpL = (int16_t*)(p2 + cplane_bytes);                               // [ringL][TW]
    int16_t* tmpC = tmpL + (size_t)ringL * TW;                                    // [2][ringC][CW]
    uint32_t* g_lh = (uint32_t*)(tmpC + (size_t)2 * ringC * CW);                 // [TW][2 * taps4] digit dwords (l.., h..)
    uint32_t* g_ch = g_lh + TW * 2 * job.lh.taps4;                                // [CW][2 * taps4]
    int32_t* l_lh = (int32_t*)(g_ch + CW * 2 * job.ch.taps4);                     // [TW]
    int32_t* l_ch = l_lh + TW;                                                    // [CW]
    // the vertical tables of a tile, two copies taken in turn (a tile's are loaded while the previous tile's are still read)
    int32_t* v_tab = l_ch + CW;
    const int v_words = TH + CH + (TH * job.lv.taps + CH * job.cv.taps + 1) / 2;
    // the source rows every tile of this segment reaches (first and past-the-last row of its luma / chroma taps): looked up once
    // for the walk -- as four scalar loads at the top of every tile they stalled the whole workgroup for a memory round trip
    int32_t* t_rows = v_tab + 2 * v_words;                                       // [tiles of the segment][4]

    // (transposed on the way in: dword q of output column i at [q][i], so that the lanes of a wavefront -- consecutive columns --
    //  read consecutive dwords; column-major rows of 4 dwords put every eighth lane on the same LDS bank)
    {
        const int nl = 2 * job.lh.taps4, nc = 2 * job.ch.taps4;
        for (int e = tid; e < tw * nl; e += 256) { const int i = e / nl, q = e - i * nl; g_lh[q * TW + i] = job.lh.digits[(size_t)X0 * nl + e]; }
        for (int e = tid; e < cw * nc; e += 256) { const int i = e / nc, q = e - i * nc; g_ch[q * CW + i] = job.ch.digits[(size_t)cX0 * nc + e]; }
    }
    if (tid < tw) l_lh[tid] = job.lh.left[X0 + tid];
    if (tid < cw) l_ch[tid] = job.ch.left[cX0 + tid];
    for (int t = tid; t < ty1 - ty0; t += 256) {
        const int Y0 = (ty0 + t) * TH, Y1 = min(job.dh, Y0 + TH), cY0 = Y0 >> 1, chh = (Y1 - Y0) >> 1;
        t_rows[4 * t + 0] = job.lv.left[Y0];
        t_rows[4 * t + 1] = job.lv.left[Y1 - 1] + job.lv.taps;
        t_rows[4 * t + 2] = job.cv.left[cY0];
        t_rows[4 * t + 3] = job.cv.left[cY0 + chh - 1] + job.cv.taps;
    }
    __syncthreads();

    // the staged columns (whole groups of four samples: a lane moves dwords)
    int xa, xb, rcols, crcols, cxa4 = 0;
    if (FMT == 0) {
        xa = min(lxa, cxa) & ~3; xb = (max(lxb, cxb) + 3) & ~3;
        rcols = xb - xa; crcols = rcols;
    } else {
        // (YUV planes are staged sixteen bytes to a lane: regions start and end on multiples of sixteen samples)
        xa = lxa & ~15; xb = (lxb + 15) & ~15;
        rcols = xb - xa;
        cxa4 = cxa & ~15;
        crcols = ((cxb + 15) & ~15) - cxa4;
    }
    // ring position of source row r (r may be negative: edge replication reaches above the picture)
    const int offL = ringL * 128, offC = ringC * 128;

    // ---- horizontal pass of one output column over FOUR consecutive rows: four taps at a time; a window's bytes are fetched as
    //      aligned dwords and shifted into place (v_alignbit), biased to int8 (xor 0x80), and multiplied by the taps' two int8 digits
    //      with v_dot4_i32_i8: sum c s = 256 * sum h (s - 128) + sum l (s - 128) + 128 * 16384 (every row of taps sums to 16384).
    //      Exact.  The column's digits are read once for the four rows (the planes' pitch is a multiple of four bytes, so the
    //      rows' windows also share their alignment).
    //      `fixed` carries the number of tap dwords when it is one of the common ones (1: no scaling, 2: 2x down, 3, 4: 4x down):
    //      the loop is then unrolled -- window dwords at immediate offsets, no rotation of the current dword through registers,
    //      no pointer arithmetic (the rolled loop spent 22 of its 38 instructions per four rows and tap dword on those).
    auto hpass4 = [&](auto fixed, const uint8_t* plane, int pitch, int row, int col, const uint32_t* g, int gstride, int taps4, int (&res)[4]) {
        constexpr int T = decltype(fixed)::value;                           // 0: any number of tap dwords (rolled loop)
        const uint32_t at = (uint32_t)(__mul24(row, pitch) + col);          // byte offset of the first row's window in the plane
        const uint32_t* w = (const uint32_t*)(plane + (at & ~3u));
        const uint32_t sh = (at & 3u) * 8u;
        const int pw = pitch >> 2;
        const uint32_t* w1 = w + pw;
        const uint32_t* w2 = w1 + pw;
        const uint32_t* w3 = w2 + pw;
        int acc_l[4] = {128 << 14, 128 << 14, 128 << 14, 128 << 14}, acc_h[4] = {0, 0, 0, 0};
        if constexpr (T > 0) {
            uint32_t win[4][T + 1];
#pragma unroll
            for (int q = 0; q <= T; q++) { win[0][q] = w[q]; win[1][q] = w1[q]; win[2][q] = w2[q]; win[3][q] = w3[q]; }
#pragma unroll
            for (int q = 0; q < T; q++) {
                const int dl = (int)g[q * gstride], dh = (int)g[(T + q) * gstride];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t sv = __builtin_amdgcn_alignbit(win[r][q + 1], win[r][q], sh) ^ 0x80808080u;
                    acc_l[r] = __builtin_amdgcn_sdot4(dl, (int)sv, acc_l[r], false);
                    acc_h[r] = __builtin_amdgcn_sdot4(dh, (int)sv, acc_h[r], false);
                }
            }
        } else {
            const uint32_t* gl = g;
            const uint32_t* gh = g + __mul24(taps4, gstride);
            uint32_t cur[4] = {w[0], w1[0], w2[0], w3[0]};
            for (int q = 0; q < taps4; q++) {
                const int dl = (int)*gl, dh = (int)*gh;
                gl += gstride; gh += gstride;
                const uint32_t nxt[4] = {w[q + 1], w1[q + 1], w2[q + 1], w3[q + 1]};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t sv = __builtin_amdgcn_alignbit(nxt[r], cur[r], sh) ^ 0x80808080u;
                    acc_l[r] = __builtin_amdgcn_sdot4(dl, (int)sv, acc_l[r], false);
                    acc_h[r] = __builtin_amdgcn_sdot4(dh, (int)sv, acc_h[r], false);
                    cur[r] = nxt[r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) res[r] = (acc_h[r] << 8) + acc_l[r];
    };
    // calls `body` with the number of tap dwords as a compile-time constant when it is a common one
    auto with_taps = [&](int taps4, auto body) {
        switch (taps4) {
            case 1: body(std::integral_constant<int, 1>{}); break;
            case 2: body(std::integral_constant<int, 2>{}); break;
            case 3: body(std::integral_constant<int, 3>{}); break;
            case 4: body(std::integral_constant<int, 4>{}); break;
            default: body(std::integral_constant<int, 0>{}); break;
        }
    };

    int have_l = -(1 << 30), have_c = -(1 << 30);          // source rows below these are in the rings (luma / chroma stream)
    for (int ty = ty0; ty < ty1; ty++) {
        const int Y0 = ty * TH, Y1 = min(job.dh, Y0 + TH);
        const int th = Y1 - Y0, chh = th >> 1, cY0 = Y0 >> 1;
        const int lya = t_rows[4 * (ty - ty0) + 0], lyb = t_rows[4 * (ty - ty0) + 1];
        const int cya = t_rows[4 * (ty - ty0) + 2], cyb = t_rows[4 * (ty - ty0) + 3];
        int32_t* vt = v_tab + ((ty - ty0) & 1) * v_words;
        int32_t* l_lv = vt;                                   // [TH] ring position of the first tap's row
        int32_t* l_cv = l_lv + TH;                            // [CH]
        int16_t* f_lv = (int16_t*)(l_cv + CH);                // [TH][taps]
        int16_t* f_cv = f_lv + TH * job.lv.taps;              // [CH][ctaps]
        for (int i = tid; i < th * job.lv.taps; i += 256) f_lv[i] = job.lv.coef[(size_t)Y0 * job.lv.taps + i];
        for (int i = tid; i < chh * job.cv.taps; i += 256) f_cv[i] = job.cv.coef[(size_t)cY0 * job.cv.taps + i];
        if (tid < th) l_lv[tid] = (job.lv.left[Y0 + tid] + offL) % ringL;
        if (tid < chh) l_cv[tid] = (job.cv.left[cY0 + tid] + offC) % ringC;

        // ---- stage the source rows this tile adds: every byte once (edge replication = clamped coordinates)
        int n0, n1, cn0, cn1;          // new rows of the luma stream (RGB: of all three components) / of the chroma stream (YUV)
        if (FMT == 0) {
            // a lane converts four pixels = three dwords of the picture (12 bytes, aligned because a row is 3 * sw bytes and sw a
            // multiple of 4) into one dword of each LDS plane
            n0 = max(min(lya, cya), have_l); n1 = max(lyb, cyb);
            cn0 = n0; cn1 = n1;
            const int rows = n1 - n0, groups = rcols >> 2;
            const bool aligned = (job.sw & 3) == 0 && (((uintptr_t)src) & 3) == 0;
            auto convert = [](int R, int G, int B, uint32_t& y, uint32_t& cb, uint32_t& cr) {
                y = (uint32_t)((19595 * R + 38470 * G + 7471 * B + 32768) >> 16);
                cb = (uint32_t)clampi(((-11059 * R - 21709 * G + 32768 * B + 32768) >> 16) + 128, 0, 255);
                cr = (uint32_t)clampi(((32768 * R - 27439 * G - 5329 * B + 32768) >> 16) + 128, 0, 255);
            };
            // kU groups per trip, all their loads issued before the first conversion: a lane that loads, converts and stores one
            // group at a time spends a memory round trip per group (the first version: 27 of them per tile, 55 % of the kernel's
            // wave-cycles waiting)
            constexpr int kU = 4;
            const int n_items = rows * groups;
            DivWalk at(tid, 256, groups);
            for (int base = tid; base < n_items; base += 256 * kU) {
                uint32_t d[kU][3];
                bool fast[kU];
                int sy[kU], x0[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const int item = base + u * 256;
                    if (item - tid >= n_items) break;             // (uniform: no lane has an item in this slot or the ones after)
                    const int r = at.q, g = at.r;
                    at.next();
                    sy[u] = clampi(n0 + r, 0, job.sh - 1);
                    x0[u] = xa + 4 * g;
                    fast[u] = item < n_items && aligned && x0[u] >= 0 && x0[u] + 3 < job.sw;
                    // every lane loads, the others from the picture's first bytes: a load under a branch is waited for before the
                    // next one is issued (the compiler closes each divergent block with s_waitcnt vmcnt(0))
                    const uint32_t off = (__umul24((uint32_t)sy[u], (uint32_t)job.sw) + (uint32_t)x0[u]) * 3u;     // < 2^31: pictures are at most 16384 x 16384
                    __builtin_memcpy(d[u], src + (fast[u] ? off : 0u), 12);
                }
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const int item = base + u * 256;
                    if (item - tid >= n_items) break;
                    if (item >= n_items) continue;
                    uint32_t Y = 0, CB = 0, CR = 0;
                    if (fast[u]) {
                        // The same three forms on v_dot4_u32_u8: a pixel's bytes (R, G, B, one byte of its neighbour against a zero
                        // coefficient) times the coefficients' two base-256 digits.  Negative coefficients are applied to the
                        // complemented byte (-c p = c (255 - p) - 255 c): with R, G complemented Cb's coefficients are 11059, 21709,
                        // 32768, with G, B complemented Cr's are 32768, 27439, 5329, and either constant comes to
                        // 32768 + (128 << 16) - 32768 * 255 = 65536.  The sums are >= 65536, so the lower clamp never acts; the upper
                        // one is a min on the 24-bit sum.  The answer is byte 2 of each sum.
                        const uint32_t d0 = d[u][0], d1 = d[u][1], d2 = d[u][2];
                        uint32_t px[4] = {d0, __builtin_amdgcn_alignbit(d1, d0, 24), __builtin_amdgcn_alignbit(d2, d1, 16), d2 >> 8};
                        uint32_t ty[4], tb[4], tr[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t p = px[k], pb = p ^ 0x0000FFFFu, pr = p ^ 0x00FFFF00u;
                            ty[k] = (__builtin_amdgcn_udot4(p, 76u | 150u << 8 | 29u << 16, 0u, false) << 8) +
                                    __builtin_amdgcn_udot4(p, 139u | 70u << 8 | 47u << 16, 32768u, false);
                            tb[k] = min((__builtin_amdgcn_udot4(pb, 43u | 84u << 8 | 128u << 16, 0u, false) << 8) +
                                        __builtin_amdgcn_udot4(pb, 51u | 205u << 8, 65536u, false), 0xFFFFFFu);
                            tr[k] = min((__builtin_amdgcn_udot4(pr, 128u | 107u << 8 | 20u << 16, 0u, false) << 8) +
                                        __builtin_amdgcn_udot4(pr, 47u << 8 | 209u << 16, 65536u, false), 0xFFFFFFu);
                        }
                        Y = __builtin_amdgcn_perm(ty[1], ty[0], 0x0c0c0602u) | __builtin_amdgcn_perm(ty[3], ty[2], 0x06020c0cu);
                        CB = __builtin_amdgcn_perm(tb[1], tb[0], 0x0c0c0602u) | __builtin_amdgcn_perm(tb[3], tb[2], 0x06020c0cu);
                        CR = __builtin_amdgcn_perm(tr[1], tr[0], 0x0c0c0602u) | __builtin_amdgcn_perm(tr[3], tr[2], 0x06020c0cu);
                    } else {                              // the picture's edges (clamped coordinates), or an unaligned picture
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint8_t* px = src + (__umul24((uint32_t)sy[u], (uint32_t)job.sw) + (uint32_t)clampi(x0[u] + k, 0, job.sw - 1)) * 3u;
                            uint32_t y, cb, cr;
                            convert(px[0], px[1], px[2], y, cb, cr);
                            Y |= y << (8 * k); CB |= cb << (8 * k); CR |= cr << (8 * k);
                        }
                    }
                    ((uint32_t*)p0)[item] = Y;            // item = r * groups + g = dword index (rcols = 4 * groups)
                    ((uint32_t*)p1)[item] = CB;
                    ((uint32_t*)p2)[item] = CR;
                }
            }
        } else {
            n0 = max(lya, have_l); n1 = lyb;
            cn0 = max(cya, have_c); cn1 = cyb;
            const uint8_t* Y = src;
            const uint8_t* U = src + (size_t)job.sw * job.sh;
            const uint8_t* V = U + (size_t)job.csw * job.csh;
            // a lane moves sixteen samples: one 16-byte load (the picture's rows need not be 16-byte aligned; the loads are dword
            // aligned), one 16-byte LDS store (the planes' rows are).  The first version moved dwords: 25 instructions per dword
            // made this plain copy 38 % of the kernel.
            auto stage = [&](uint8_t* dst, const uint8_t* plane, int pw, int ph, int y0, int rows, int x0, int cols) {
                const int groups = cols >> 4, n_items = rows * groups;
                const bool aligned = (pw & 3) == 0 && (((uintptr_t)plane) & 3) == 0;
                constexpr int kU = 2;
                DivWalk at(tid, 256, groups);
                for (int base = tid; base < n_items; base += 256 * kU) {
                    uint4 v[kU];
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        const int item = base + u * 256;
                        const int r = at.q, g = at.r;
                        at.next();
                        const int sy = clampi(y0 + r, 0, ph - 1), x = x0 + 16 * g;
                        const uint32_t rowoff = __umul24((uint32_t)sy, (uint32_t)pw);      // (planes are at most 16384 x 16384)
                        v[u] = make_uint4(0u, 0u, 0u, 0u);
                        if (item < n_items) {
                            if (aligned && x >= 0 && x + 15 < pw) {
                                __builtin_memcpy(&v[u], plane + (rowoff + (uint32_t)x), 16);
                            } else {                          // the picture's edges: clamped coordinates, byte by byte
                                auto four = [&](int xk) {
                                    uint32_t d = 0u;
#pragma unroll
                                    for (int k = 0; k < 4; k++) d |= (uint32_t)plane[rowoff + (uint32_t)clampi(xk + k, 0, pw - 1)] << (8 * k);
                                    return d;
                                };
                                v[u] = make_uint4(four(x), four(x + 4), four(x + 8), four(x + 12));
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kU; u++)
                        if (base + u * 256 < n_items) ((uint4*)dst)[base + u * 256] = v[u];
                }
            };
            stage(p0, Y, job.sw, job.sh, n0, n1 - n0, xa, rcols);
            stage(p1, U, job.csw, job.csh, cn0, cn1 - cn0, cxa4, crcols);      // Cb
            stage(p2, V, job.csw, job.csh, cn0, cn1 - cn0, cxa4, crcols);      // Cr
        }
        have_l = n1; have_c = cn1;
        __syncthreads();

        // ---- horizontal pass over the new rows, LDS -> the rings: 15-bit intermediates (+ the range expansion of limited-range input)
        //      (rows past the last new one in the last group of four read whatever follows in LDS and are not stored)
        {
            const int rows = n1 - n0, c_off = -xa, taps4 = job.lh.taps4;
            const int ring0 = (n0 + offL) % ringL;
            with_taps(taps4, [&](auto fixed) {
                DivWalk la(tid, 256, tw);
                for (int item = tid; item < ((rows + 3) >> 2) * tw; item += 256, la.next()) {
                    const int r4 = la.q * 4, i = la.r;
                    int acc[4];
                    hpass4(fixed, p0, rcols, r4, l_lh[i] + c_off, g_lh + i, TW, taps4, acc);
                    int rr = ring0 + r4;
                    if (rr >= ringL) rr -= ringL;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        int t = clampi(acc[r] >> 7, 0, 32767);
                        if (job.limited) t = (__mul24(min(t, 30189), 19077) - 39057361) >> 14;        // (at most 30189 * 19077 < 2^31)
                        if (r4 + r < rows) tmpL[__mul24(rr, TW) + i] = (int16_t)t;
                        rr = rr + 1 == ringL ? 0 : rr + 1;
                    }
                }
            });
            const int crows = cn1 - cn0, cc_off = FMT == 0 ? -xa : -cxa4, ctaps4 = job.ch.taps4;
            const int cring0 = (cn0 + offC) % ringC;
            const int cgroups = (crows + 3) >> 2;
            with_taps(ctaps4, [&](auto fixed) {
                DivWalk ca(tid, 256, cw);
                for (int item = tid; item < 2 * cgroups * cw; item += 256, ca.next()) {
                    const int comp = ca.q >= cgroups ? 1 : 0, r4 = (ca.q - comp * cgroups) * 4, i = ca.r;      // groups 0..cgroups-1: Cr, then Cb
                    int acc[4];
                    hpass4(fixed, comp ? p1 : p2, crcols, r4, l_ch[i] + cc_off, g_ch + i, CW, ctaps4, acc);      // comp 0 = Cr, 1 = Cb
                    int rr = cring0 + r4;
                    if (rr >= ringC) rr -= ringC;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        int t = clampi(acc[r] >> 7, 0, 32767);
                        if (job.limited) t = (__mul24(min(t, 30775), 4663) - 9289992) >> 12;
                        if (r4 + r < crows) tmpC[__mul24(__mul24(comp, ringC) + rr, CW) + i] = (int16_t)t;
                        rr = rr + 1 == ringC ? 0 : rr + 1;
                    }
                }
            });
        }
        __syncthreads();

        // ---- vertical pass, rings -> registers -> HBM: a lane makes four adjacent luma bytes / two adjacent Cr,Cb pairs
        {
            const int taps = job.lv.taps, q = tw >> 2;
            DivWalk va(tid, 256, q);
            for (int item = tid; item < th * q; item += 256, va.next()) {
                const int j = va.q, i4 = va.r * 4;
                const int16_t* f = f_lv + j * taps;
                int rr = l_lv[j];
                const int16_t* t = tmpL + __mul24(rr, TW) + i4;
                int a0 = 1 << 20, a1 = 1 << 20, a2 = 1 << 20, a3 = 1 << 20;
                for (int k = 0; k < taps; k++) {
                    const int c = (int)f[k];
                    const uint2 v4 = *(const uint2*)t;                       // four int16 (i4 and TW are multiples of 4)
                    a0 += c * (int)(int16_t)(v4.x & 0xFFFFu); a1 += c * ((int)v4.x >> 16);
                    a2 += c * (int)(int16_t)(v4.y & 0xFFFFu); a3 += c * ((int)v4.y >> 16);
                    rr++; t += TW;
                    if (rr == ringL) { rr = 0; t = tmpL + i4; }
                }
                const uint32_t v = out8(a0) | out8(a1) << 8 | out8(a2) << 16 | out8(a3) << 24;
                *(uint32_t*)(out + (size_t)(Y0 + j) * job.dw + X0 + i4) = v;
            }
            const int ctaps = job.cv.taps, cq = cw >> 1;
            uint8_t* cout = out + (size_t)job.dw * job.dh;
            DivWalk vc(tid, 256, cq);
            for (int item = tid; item < chh * cq; item += 256, vc.next()) {
                const int j = vc.q, i2 = vc.r * 2;
                const int16_t* f = f_cv + j * ctaps;
                int rr = l_cv[j];
                const int cb_off = __mul24(ringC, CW);
                const int16_t* cr = tmpC + __mul24(rr, CW) + i2;
                int r0 = 1 << 20, r1 = 1 << 20, b0 = 1 << 20, b1 = 1 << 20;
                for (int k = 0; k < ctaps; k++) {
                    const int c = (int)f[k];
                    const uint32_t vr = *(const uint32_t*)cr, vb = *(const uint32_t*)(cr + cb_off);      // two int16 each (i2, CW, ringC * CW even)
                    r0 += c * (int)(int16_t)(vr & 0xFFFFu); r1 += c * ((int)vr >> 16);
                    b0 += c * (int)(int16_t)(vb & 0xFFFFu); b1 += c * ((int)vb >> 16);
                    rr++; cr += CW;
                    if (rr == ringC) { rr = 0; cr = tmpC + i2; }
                }
                // NV21: Cr at even bytes, Cb at odd (mdec.c:627-628)
                const uint32_t v = out8(r0) | out8(b0) << 8 | out8(r1) << 16 | out8(b1) << 24;
                *(uint32_t*)(cout + (size_t)(cY0 + j) * job.dw + (size_t)(cX0 + i2) * 2) = v;
            }
        }
    }
}

// ---- filter bank, host side: the specification's integer arithmetic (see the header of oracle/frontend_oracle.c for the
//      same text as prose; the two are written independently and compared tap by tap in tests/test_gpu_frontend.py)
int64_t floor_div64(int64_t a, int64_t b) {
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
    return q;
}
int64_t bicubic_weight(int64_t x) {          // x: |distance| / scale in 16.16; B = 0, C = 0.6, times 10 * 2^16
    const int64_t one = 65536;
    if (x < one) return ((14 * x * x * x) >> 32) - ((24 * x * x) >> 16) + 10 * one;
    if (x < 2 * one) return -((6 * x * x * x) >> 32) + ((30 * x * x) >> 16) - 48 * x + 24 * one;
    return 0;
}
struct HostBank {
    int taps = 0, taps4 = 0;
    std::vector<int32_t> left;
    std::vector<int16_t> coef;
    std::vector<uint32_t> digits;       // see Bank::digits
};
bool make_bank(int src, int dst, HostBank* b) {
    const int64_t xinc = (((int64_t)src << 16) + dst / 2) / dst;
    const int64_t scale = xinc > 65536 ? xinc : 65536;
    const int64_t R = 2 * scale;
    b->taps = (int)((2 * R + 65535) >> 16);
    if (b->taps > 64) return false;
    b->left.resize((size_t)dst);
    b->coef.resize((size_t)dst * b->taps);
    std::vector<int64_t> W((size_t)b->taps);
    for (int i = 0; i < dst; i++) {
        const int64_t c = (int64_t)i * xinc + ((xinc - 65536) >> 1);
        const int64_t l = floor_div64(c - R, 65536) + 1;
        int64_t sum = 0;
        int best = 0;
        for (int k = 0; k < b->taps; k++) {
            int64_t d = ((l + k) << 16) - c;
            if (d < 0) d = -d;
            W[(size_t)k] = bicubic_weight(d * 65536 / scale);
            sum += W[(size_t)k];
            if (W[(size_t)k] > W[(size_t)best]) best = k;
        }
        int64_t got = 0;
        for (int k = 0; k < b->taps; k++) {
            const int64_t q = W[(size_t)k] * 16384 / sum;
            b->coef[(size_t)i * b->taps + k] = (int16_t)q;
            got += q;
        }
        b->coef[(size_t)i * b->taps + best] = (int16_t)(b->coef[(size_t)i * b->taps + best] + (16384 - got));
        b->left[(size_t)i] = (int32_t)l;
    }
    // the taps as two balanced int8 digits, four to a dword (the horizontal pass's v_dot4_i32_i8 operands)
    b->taps4 = (b->taps + 3) / 4;
    b->digits.assign((size_t)dst * 2 * b->taps4, 0u);
    for (int i = 0; i < dst; i++)
        for (int k = 0; k < b->taps; k++) {
            const int c = b->coef[(size_t)i * b->taps + k];
            const int lo = ((c + 128) & 255) - 128, hi = (c - lo) >> 8;
            if (hi < -128 || hi > 127) return false;
            b->digits[((size_t)i * 2 + 0) * b->taps4 + k / 4] |= (uint32_t)(lo & 0xFF) << (8 * (k & 3));
            b->digits[((size_t)i * 2 + 1) * b->taps4 + k / 4] |= (uint32_t)(hi & 0xFF) << (8 * (k & 3));
        }
    return true;
}

}  // namespace

struct psxhip_scaler {
    int device, fmt, sw, sh, full_range, dw, dh;
    HostBank h[4];                 // lh, lv, ch, cv
    int32_t* d_left[4] = {nullptr, nullptr, nullptr, nullptr};
    int16_t* d_coef[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t* d_digits[4] = {nullptr, nullptr, nullptr, nullptr};
    ScalerJob job;
    size_t lds_bytes;
    size_t src_bytes;              // bytes of one source picture
    int n_cus;
};

#define HIP_TRY(expr, code)                                                                   \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            psxhip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (code);                                                                    \
        }                                                                                     \
    } while (0)

extern "C" void psxhip_scaler_destroy(psxhip_scaler_t* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (int i = 0; i < 4; i++) {
        if (s->d_left[i]) (void)hipFree(s->d_left[i]);
        if (s->d_coef[i]) (void)hipFree(s->d_coef[i]);
        if (s->d_digits[i]) (void)hipFree(s->d_digits[i]);
    }
    delete s;
}

extern "C" int psxhip_scaler_create(psxhip_scaler_t** out, int device, int src_format, int src_width, int src_height,
                                    int src_full_range, int dst_width, int dst_height) {
    if (!out) return PSXHIP_EINVAL;
    *out = nullptr;
    const bool yuv = src_format == PSXHIP_PIX_YUV420P;
    if ((src_format != PSXHIP_PIX_RGB24 && !yuv) || src_width < 2 || src_height < 2 || src_width > 16384 || src_height > 16384 ||
        dst_width < 16 || dst_height < 16 || (dst_width % 16) || (dst_height % 16) || dst_width > 1024 || dst_height > 1024 ||
        (yuv && ((src_width | src_height) & 1))) {
        psxhip_set_error("psxhip_scaler_create: bad geometry (%dx%d format %d -> %dx%d; the target must be a multiple of 16, YUV420P sources even)",
                         src_width, src_height, src_format, dst_width, dst_height);
        return PSXHIP_EINVAL;
    }
    int rc = psxhip_ensure_device(device);
    if (rc) return rc;
    psxhip_scaler* s = new (std::nothrow) psxhip_scaler;
    if (!s) return PSXHIP_ENOMEM;
    struct Guard { psxhip_scaler* p; ~Guard() { if (p) psxhip_scaler_destroy(p); } } guard{s};
    s->device = device; s->fmt = src_format; s->sw = src_width; s->sh = src_height; s->full_range = src_full_range;
    s->dw = dst_width; s->dh = dst_height;
    const int csw = yuv ? src_width / 2 : src_width, csh = yuv ? src_height / 2 : src_height;
    if (!make_bank(src_width, dst_width, &s->h[0]) || !make_bank(src_height, dst_height, &s->h[1]) ||
        !make_bank(csw, dst_width / 2, &s->h[2]) || !make_bank(csh, dst_height / 2, &s->h[3])) {
        psxhip_set_error("psxhip_scaler_create: shrinking by more than 16x is not supported");
        return PSXHIP_EINVAL;
    }
    s->src_bytes = yuv ? (size_t)src_width * src_height * 3 / 2 : (size_t)src_width * src_height * 3;
    ScalerJob& j = s->job;
    memset(&j, 0, sizeof j);
    j.sw = src_width; j.sh = src_height; j.dw = dst_width; j.dh = dst_height;
    j.csw = csw; j.csh = csh;
    j.limited = yuv && !src_full_range;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device), PSXHIP_EDEVICE);
    s->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // tile: the largest of these whose LDS working set leaves room for at least two workgroups per CU
    const int shapes[4][2] = {{64, 16}, {32, 16}, {32, 8}, {16, 8}};
    size_t need = 0;
    bool ok = false;
    for (int t = 0; t < 4 && !ok; t++) {
        const int TW = shapes[t][0], TH = shapes[t][1];
        // the largest source reach of any tile, from the tables
        auto reach = [](const HostBank& b, int n, int tile, int* lo_of_first, int* span) {
            int best = 0;
            for (int a = 0; a < n; a += tile) {
                const int e = (a + tile < n ? a + tile : n) - 1;
                const int sp = b.left[(size_t)e] + b.taps - b.left[(size_t)a];
                if (sp > best) best = sp;
            }
            (void)lo_of_first;
            *span = best;
        };
        int lc, lr, cc, cr;
        reach(s->h[0], dst_width, TW, nullptr, &lc);
        reach(s->h[1], dst_height, TH, nullptr, &lr);
        reach(s->h[2], dst_width / 2, TW / 2, nullptr, &cc);
        reach(s->h[3], dst_height / 2, TH / 2, nullptr, &cr);
        int reg_rows, reg_cols, creg_rows, creg_cols;
        if (yuv) {
            // every plane's region starts and ends on a multiple of sixteen samples (a lane stages sixteen bytes)
            auto reach4 = [](const HostBank& b, int n, int tile) {
                int best = 0;
                for (int a = 0; a < n; a += tile) {
                    const int e = (a + tile < n ? a + tile : n) - 1;
                    const int lo = b.left[(size_t)a] & ~15, hi = (b.left[(size_t)e] + b.taps + 15) & ~15;
                    if (hi - lo > best) best = hi - lo;
                }
                return best;
            };
            reg_rows = lr; reg_cols = reach4(s->h[0], dst_width, TW); creg_rows = cr; creg_cols = reach4(s->h[2], dst_width / 2, TW / 2);
            (void)lc; (void)cc;
        } else {
            // the union of the luma and the chroma reach over the same full-resolution picture: bounded by the larger span plus
            // the offset between the two windows (at most the larger filter's half width); take the exact maximum over the tiles
            reg_rows = 0; reg_cols = 0;
            for (int a = 0; a < dst_width; a += TW) {
                const int e = (a + TW < dst_width ? a + TW : dst_width) - 1;
                const int lo = std::min(s->h[0].left[(size_t)a], s->h[2].left[(size_t)(a / 2)]) & ~3;       // whole groups of four pixels
                const int hi = (std::max(s->h[0].left[(size_t)e] + s->h[0].taps, s->h[2].left[(size_t)(e / 2)] + s->h[2].taps) + 3) & ~3;
                reg_cols = std::max(reg_cols, hi - lo);
            }
            for (int a = 0; a < dst_height; a += TH) {
                const int e = (a + TH < dst_height ? a + TH : dst_height) - 1;
                const int lo = std::min(s->h[1].left[(size_t)a], s->h[3].left[(size_t)(a / 2)]);
                const int hi = std::max(s->h[1].left[(size_t)e] + s->h[1].taps, s->h[3].left[(size_t)(e / 2)] + s->h[3].taps);
                reg_rows = std::max(reg_rows, hi - lo);
            }
            creg_rows = reg_rows; creg_cols = reg_cols;
        }
        // + 8: the horizontal pass reads a window as whole dwords, up to 7 bytes past the last tap (zero digits there)
        const size_t plane = ((size_t)reg_rows * reg_cols + 8 + 15) & ~(size_t)15;
        const size_t cplane = yuv ? (((size_t)creg_rows * creg_cols + 8 + 15) & ~(size_t)15) : plane;
        const size_t crows_cap = yuv ? (size_t)creg_rows : (size_t)reg_rows;
        const size_t v_words = (size_t)TH + TH / 2 + ((size_t)TH * s->h[1].taps + (size_t)(TH / 2) * s->h[3].taps + 1) / 2;      // one tile's vertical tables
        need = plane + 2 * cplane + 2 * ((size_t)reg_rows * TW + 2 * crows_cap * (TW / 2)) +
               8 * ((size_t)TW * s->h[0].taps4 + (size_t)(TW / 2) * s->h[2].taps4) + 4 * ((size_t)TW + TW / 2) + 4 * 2 * v_words + 16 * (((size_t)dst_height + TH - 1) / TH) + 16;
        if (need * 2 <= (size_t)prop.maxSharedMemoryPerMultiProcessor || (t == 3 && need <= (size_t)prop.maxSharedMemoryPerMultiProcessor)) {
            ok = true;
            j.TW = TW; j.TH = TH;
            j.reg_rows = reg_rows; j.reg_cols = reg_cols; j.creg_rows = creg_rows; j.creg_cols = creg_cols;
        }
    }
    if (!ok) {
        psxhip_set_error("psxhip_scaler_create: the filters' reach (%zu bytes of LDS per tile) does not fit a compute unit", need);
        return PSXHIP_EINVAL;
    }
    s->lds_bytes = need;
    j.tiles_x = (dst_width + j.TW - 1) / j.TW;
    j.tiles_y = (dst_height + j.TH - 1) / j.TH;
    for (int i = 0; i < 4; i++) {
        HIP_TRY(hipMalloc((void**)&s->d_left[i], s->h[i].left.size() * sizeof(int32_t)), PSXHIP_ENOMEM);
        HIP_TRY(hipMalloc((void**)&s->d_coef[i], s->h[i].coef.size() * sizeof(int16_t)), PSXHIP_ENOMEM);
        HIP_TRY(hipMemcpy(s->d_left[i], s->h[i].left.data(), s->h[i].left.size() * sizeof(int32_t), hipMemcpyHostToDevice), PSXHIP_EDEVICE);
        HIP_TRY(hipMemcpy(s->d_coef[i], s->h[i].coef.data(), s->h[i].coef.size() * sizeof(int16_t), hipMemcpyHostToDevice), PSXHIP_EDEVICE);
        HIP_TRY(hipMalloc((void**)&s->d_digits[i], s->h[i].digits.size() * sizeof(uint32_t)), PSXHIP_ENOMEM);
        HIP_TRY(hipMemcpy(s->d_digits[i], s->h[i].digits.data(), s->h[i].digits.size() * sizeof(uint32_t), hipMemcpyHostToDevice), PSXHIP_EDEVICE);
    }
    Bank* banks[4] = {&j.lh, &j.lv, &j.ch, &j.cv};
    for (int i = 0; i < 4; i++) {
        banks[i]->left = s->d_left[i];
        banks[i]->coef = s->d_coef[i];
        banks[i]->digits = s->d_digits[i];
        banks[i]->taps = s->h[i].taps;
        banks[i]->taps4 = s->h[i].taps4;
    }
    if (yuv) HIP_TRY(hipFuncSetAttribute((const void*)scaler_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prop.maxSharedMemoryPerMultiProcessor), PSXHIP_EDEVICE);
    else HIP_TRY(hipFuncSetAttribute((const void*)scaler_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prop.maxSharedMemoryPerMultiProcessor), PSXHIP_EDEVICE);
    guard.p = nullptr;
    *out = s;
    return PSXHIP_OK;
}

extern "C" size_t psxhip_scaler_source_bytes(const psxhip_scaler_t* s) { return s ? s->src_bytes : 0; }

extern "C" int psxhip_scaler_filter(const psxhip_scaler_t* s, int which, int* taps, int32_t* left, int16_t* coef, int cap) {
    if (!s || which < 0 || which > 3) return PSXHIP_EINVAL;
    const HostBank& b = s->h[which];
    if (taps) *taps = b.taps;
    const int n = (int)b.left.size();
    if (left && coef) {
        if (cap < n * b.taps) return PSXHIP_EINVAL;
        memcpy(left, b.left.data(), (size_t)n * sizeof(int32_t));
        memcpy(coef, b.coef.data(), (size_t)n * b.taps * sizeof(int16_t));
    }
    return n;
}

extern "C" int psxhip_scaler_convert_device(psxhip_scaler_t* s, const uint8_t* d_src, size_t src_stride, int n_frames,
                                            uint8_t* d_frames, size_t frame_stride, void* stream) {
    if (!s || !d_src || !d_frames || n_frames < 0) {
        psxhip_set_error("psxhip_scaler_convert_device: NULL argument");
        return PSXHIP_EINVAL;
    }
    if (n_frames == 0) return PSXHIP_OK;
    if (src_stride < s->src_bytes || frame_stride < (size_t)s->dw * s->dh * 3 / 2 || (frame_stride & 3) || ((uintptr_t)d_frames & 3)) {
        psxhip_set_error("psxhip_scaler_convert_device: strides too small, or the output not 4-byte aligned");
        return PSXHIP_EINVAL;
    }
    HIP_TRY(hipSetDevice(s->device), PSXHIP_EDEVICE);
    ScalerJob j = s->job;
    j.src = d_src; j.src_stride = src_stride; j.out = d_frames; j.frame_stride = frame_stride;
    // a band walks the picture top to bottom; a small batch is cut into vertical segments until the chip has ~4 workgroups per CU
    // (each segment re-reads the rows its first tile reaches)
    const long long want = 4LL * s->n_cus;
    long long segs = (want + (long long)j.tiles_x * n_frames - 1) / ((long long)j.tiles_x * n_frames);
    if (segs < 1) segs = 1;
    if (segs > j.tiles_y) segs = j.tiles_y;
    if (const char* e = getenv("PSXHIP_SCALER_VSEGS")) { segs = atoi(e); if (segs < 1) segs = 1; if (segs > j.tiles_y) segs = j.tiles_y; }   // experiments
    j.vsegs = (int)segs;
    for (int f0 = 0; f0 < n_frames; f0 += 65535) {            // gridDim.y limit
        const int nf = n_frames - f0 < 65535 ? n_frames - f0 : 65535;
        j.src = d_src + (size_t)f0 * src_stride;
        j.out = d_frames + (size_t)f0 * frame_stride;
        const dim3 grid((unsigned)j.tiles_x, (unsigned)nf, (unsigned)j.vsegs);
        if (s->fmt == PSXHIP_PIX_YUV420P)
            hipLaunchKernelGGL(scaler_kernel<1>, grid, dim3(256), s->lds_bytes, (hipStream_t)stream, j);
        else
            hipLaunchKernelGGL(scaler_kernel<0>, grid, dim3(256), s->lds_bytes, (hipStream_t)stream, j);
    }
    HIP_TRY(hipGetLastError(), PSXHIP_EDEVICE);
    return PSXHIP_OK;
}

extern "C" int psxhip_scaler_convert_host(psxhip_scaler_t* s, const uint8_t* src, int n_frames, uint8_t* frames) {
    if (!s || !src || !frames || n_frames < 0) return PSXHIP_EINVAL;
    if (n_frames == 0) return PSXHIP_OK;
    HIP_TRY(hipSetDevice(s->device), PSXHIP_EDEVICE);
    const size_t fsz = (size_t)s->dw * s->dh * 3 / 2;
    uint8_t *d_src = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_src, s->src_bytes * (size_t)n_frames), PSXHIP_ENOMEM);
    if (hipMalloc((void**)&d_out, fsz * (size_t)n_frames) != hipSuccess) { (void)hipFree(d_src); return PSXHIP_ENOMEM; }
    hipError_t e = hipMemcpy(d_src, src, s->src_bytes * (size_t)n_frames, hipMemcpyHostToDevice);
    int rc = PSXHIP_OK;
    if (e == hipSuccess) rc = psxhip_scaler_convert_device(s, d_src, s->src_bytes, n_frames, d_out, fsz, nullptr);
    if (e == hipSuccess && rc == PSXHIP_OK) e = hipMemcpy(frames, d_out, fsz * (size_t)n_frames, hipMemcpyDeviceToHost);
    (void)hipFree(d_src);
    (void)hipFree(d_out);
    if (e != hipSuccess) {
        psxhip_set_error("psxhip_scaler_convert_host: %s", hipGetErrorString(e));
        return PSXHIP_EDEVICE;
    }
    return rc;
}
