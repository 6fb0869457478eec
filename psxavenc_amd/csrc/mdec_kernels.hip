// mdec_kernels.hip -- MDEC "BS" frame encoder for MI355X (gfx950), hand-written HIP.
//
// Replaces, for batches of frames resident in HBM, what the reference does per frame in
// psxavenc/mdec.c:580-755 (encode_frame_bs): NV21 -> macroblocks -> 8x8 forward DCT -> quantise ->
// zig-zag/RLE -> BS v2/v3 VLC -> bit-pack, inside the "first quant scale that fits" loop.
//
// Mapping (see DESIGN.md for the reasoning):
//   * one workgroup owns one frame at a time (persistent grid, frames handed out by an atomic ticket);
//   * one wavefront owns one macroblock at a time: every lane hands the 8 pixels of its (block, row) to ONE
//     v_mfma_i32_32x32x16_i8, which evaluates the row pass's eight exact integer forms for all 48 row vectors (pixels as one
//     int8 digit, coefficients as two; fdct8_rows_mfma()); 48 lanes run the column pass on packed int16 pairs
//     (v_pk_add_i16 + v_dot2_i32_i16; the 8x8 transposes are staged in LDS), then lane k owns zig-zag position k of every block;
//   * quantisation is an exact integer rounding division done with one fp32 fma (proof at quant_mag());
//     run lengths come from a 64-bit ballot + count-leading-zeros; code lengths and codes from
//     (run, |level|) LUTs held in LDS; bit offsets from DPP prefix sums;
//   * rate control does NOT evaluate every scale.  A pilot (one macroblock per wavefront, strided over
//     the frame) predicts the answer p; one fused pass over the frame then counts the bits at p-1 and
//     builds the bitstream at p.  The count also yields a proven lower bound for ALL finer scales
//     (mdec_search.h), so "p-1 and everything below it fail, p fits" is established exactly like the
//     reference's ascending loop would -- in one pass.  Mispredictions cost further passes (the DCT is
//     recomputed from the frame, which is L2 / Infinity-Cache resident), never exactness;
//   * macroblock bitstreams are built independently in an LDS staging area (a macroblock's position in the
//     frame's stream depends on all macroblocks before it), then an exclusive scan in encode order and a
//     funnel-shift merge place them; the frame leaves the CU as coalesced dword stores, header and zero
//     tail included (the reference's memset, mdec.c:676).
// There is no per-frame scratch in global memory: a frame's coefficients never leave the CU.
//
// The matrix pipe is used for exactly one thing: the DCT's row pass, as an exact int8-digit contraction (round 3; it frees
// 15 of 262 VALU instructions per macroblock on the pipe that bounds the kernel).  The column pass (int16 inputs: two more
// digits, three weight classes, a digit split of the row outputs) costs as many VALU instructions on the matrix pipe as off
// it, and everything after the DCT is table look-ups and bit manipulation.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

#include "bs_vlc_lut.h"
#include "mdec_search.h"
#include "psxhip_internal.h"
#include "wave_ops.h"

namespace {

// Two workgroup shapes: 12 wavefronts at 6 per SIMD (two frames per CU; 80 VGPRs) when two groups' LDS fits one CU,
// otherwise 16 wavefronts at 4 per SIMD (one frame per CU; 128 VGPRs) -- large frames / large budgets.
constexpr int kWavesSmall = 12, kOccSmall = 6;
constexpr int kWavesLarge = 16, kOccLarge = 4;
constexpr int kTileStride = 72;   // int16 per block in the transpose tile: 6 blocks land on disjoint LDS banks
constexpr int kZStride = 64;      // int16 per block in the coefficient tile: column-pass lane t stores its 8 outputs at bytes 16 t
constexpr int kPilotMax = 4;      // scales evaluated per pilot round
constexpr int kMaxTiles = 16;     // image tiles of 2048 dwords: budgets up to 128 KiB
// profiling builds (-DPSX_EXP_STOP_AFTER=n, WRONG BYTES, every frame ends after its first pass): a macroblock's work stops after
// 1 = ticket + pixel fetch, 2 = + DCT, 3 = + coefficient read and list build.  PMC differences between them and the product build
// are where the instruction inventory in DESIGN.md comes from (tools/gpu_pmc_quick.sh, tools/build_variant.sh).
#ifdef PSX_EXP_STOP_AFTER
constexpr int kStopAfter = PSX_EXP_STOP_AFTER;
#else
constexpr int kStopAfter = 0;
#endif
constexpr uint32_t kNoMb = 0xFFFFu;   // pass order entry without a macroblock (the last round of tickets may be partial)
constexpr uint32_t kRetryEmpty = 0xFFFFFFFFu, kRetryAbandoned = 0xFFFFFFFEu;
// Frame tickets and the retry queue's state share one 64-bit word of the ticket buffer, in a cache line of its own: fresh-frame
// tickets drawn (low 32 bits) | queue slots reserved (16 bits) | pop tickets drawn (16 bits).  (The queue is only used for launches
// of at most 8 frames per group, so 16 bits are plenty; the ticket counter is the word's low half so that a draw needs no
// arithmetic on the returned value -- anything computed from it on the spot would make the compiler wait for the atomic there.)
constexpr int kQueueWord = 64;
constexpr int kStartedWord = 96;         // groups of this launch that have started (a cache line of its own; self-resetting)
// Groups that have left | abandoned queue slots << 16 | what the groups learned about foreign hints: wrong << 32 | tried << 48 (each
// group adds at most kTrustCap of either) -- ONE 64-bit word, so that the group that leaves last sees every other group's sums
// without a fence: they were added by the same atomic that counted the group out.
constexpr int kLeaveWord = 8;
constexpr int kLeaveWrongShift = 32, kLeaveTriedShift = 48;
constexpr unsigned kTrustCap = 63u;
constexpr int kNoTicket = 0x7FFFFFFF;   // S_FRAME of a group that holds no fresh-frame ticket (any more): it takes frames from the retry queue
constexpr int kPilotWord = 10;           // (64 bits) what the groups learned about the PILOT's guesses: wrong | tried << 32, added before the group counts itself out
constexpr int kDistrustWord = 4;         // hint[kDistrustWord]: the launches before this one found foreign hints wrong more than one time in four (shared by the context's lanes, like the hint)
constexpr int kQueueReservedShift = 32, kQueueHeadShift = 48;
constexpr unsigned kQueueMask = 0xFFFFu;
// A look at a word other XCDs write: a device-scope load (it bypasses this XCD's L2); after many looks in vain, a read-modify-write
// -- which is performed at the memory side whatever the caches do -- so that progress never rests on the load's coherence alone.
template <typename T>
__device__ __forceinline__ T queue_peek(T* p, int looks) {
    return looks < 4096 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : atomicOr(p, (T)0);
}

__constant__ uint16_t c_ac_len16[BS_LUT_SIZE];
__constant__ uint32_t c_ac_code[BS_LUT_SIZE];
__constant__ uint8_t c_zagzig[64];
__constant__ uint8_t c_quant_zz[64];
__constant__ uint8_t c_dc_prefix[2][8];   // [0] chroma, [1] luma
__constant__ uint8_t c_dc_plen[2][8];

// One launch encodes the frames of up to kMaxBatches batches (include/psxav_hip.h: psxhip_mdec_encode_batches_device): frame f of the
// launch is frame f - first of the last batch whose `first` is <= f.  The table travels in the kernel arguments.
constexpr int kMaxBatches = PSXHIP_MDEC_MAX_BATCHES;
struct BatchDesc {
    const uint8_t* frames;
    uint8_t* out;
    psxhip_mdec_result_t* results;
    const int32_t* max_sizes;      // per-frame budgets of this batch, or NULL (the launch's uniform budget)
};
struct FrameJob {
    BatchDesc batch[kMaxBatches];   // (first member: batch_table())
    int first[kMaxBatches];         // index of each batch's first frame in the launch; INT_MAX for the entries past the last batch
    int n_batches;
    size_t frame_stride;
    int width, height, nx, ny, nmb;
    int n_frames;                  // over all batches
    int n_tickets, t4, t2;         // frame tickets of the launch: the first t4 are runs of 4 consecutive frames, the next t2 runs of 2, the rest single frames (ticket_run())
    int uniform_max_size;
    size_t out_stride;
    int out_words;           // LDS dwords reserved for the frame image tile (out_tile + 2)
    int out_tile;            // dwords of the frame image assembled in LDS at a time
    int max_frame_size;      // the context's largest budget
    int stg_words;           // LDS dwords of the macroblock staging area
    int trips;               // iterations of a pass: ceil(nmb / wavefronts per group)
    int it_step;             // iteration visiting stride (coprime with trips), see psxhip_mdec_pass_order()
    const uint32_t* order;   // [2 * trips * wavefronts per group] per pass ticket t: what the loop needs of its macroblock (psxhip_mdec_pass_table)
    unsigned int* ticket;    // [128] of this launch's LANE: [1] workgroups finished (self-resetting), [3] frames lost by the retry queue's watchdog (never reset; psxhip_mdec_watchdog), [64..65] frame tickets + the retry queue's state, one 64-bit word, [96] groups started (self-resetting)
    unsigned int* hint;      // one word shared by the context's lanes: answer | budget << 8 of the last frame (by index) of the launch that wrote it last -- a hint that survives launches
    unsigned int* retry;     // [retry_cap] retry queue: frame | scale to start from << 24, kRetryEmpty when vacant (NULL: frames are never handed on)
    int retry_patience;      // looks (about 3 us each) a group without work waits for a frame to be handed on
    int retry_cap;
    unsigned prio_pattern;       // [7:0] older group, [15:8] younger group: bit (iteration & 7) = raised priority
    int trust_mode;              // 0: the trust policy (below); experiments: 1 = foreign hints always trusted (the kernels before mdec-k3.7), 2 = never
    int ck_margin;               // quarter-pass checkpoint: how far (thousandths of its standard error) a projection has to be on the wrong side
    unsigned long long* stats;   // optional [PSXHIP_MDEC_STATS]: pass counters (diagnostics), NULL in normal runs
    uint32_t col_k[16];          // the column pass's sixteen coefficient pairs (col_coeffs()), fetched by ONE scalar load per macroblock
};
// the batch table where it lies in the kernel argument segment (FrameJob is the kernel's only argument, `batch` its first member)
typedef const BatchDesc __attribute__((address_space(4))) * BatchPtr;
typedef const int __attribute__((address_space(4))) * FirstPtr;
// n_batches, read where it is asked for (as a plain member the compiler kept "n_batches > 1" for the whole kernel -- as 0 / 1 in a
// vector register it then spilled)
#define PSX_BATCHES_MANY() (((FirstPtr)(batch_table() + kMaxBatches))[kMaxBatches] > 1)
__device__ __forceinline__ BatchPtr batch_table() {
    BatchPtr p = (BatchPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
// An int member of the kernel argument, loaded where it is asked for (a scalar load from the argument segment): as plain members
// the arguments the frame loop reads now and then -- the ticket plan, the trust mode, the queue's patience -- are loaded at kernel
// entry and kept, one scalar register each for the whole kernel, in a kernel that spills a hundred of them: mdec-k3.7's four new
// ones pushed the frame pointer of the macroblock loop out of its scalar pair (two v_readfirstlane and a v_readlane more per
// macroblock: +1.5 % vector instructions, found with the instruction counters of tools/gpu_r05_session_p.sh).
#define PSX_JOB_INT(member) (((FirstPtr)((const char __attribute__((address_space(4)))*)batch_table() + offsetof(FrameJob, member)))[0])
// the retry queue's slots, the address made where it is used (one thread, rarely): as a loop invariant the 64-bit address sat in
// two vector registers -- or a scratch slot -- for the whole kernel
__device__ __forceinline__ unsigned int* retry_slots(const FrameJob& job) {
    unsigned int* p = job.retry;
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ unsigned long long* queue_state(const FrameJob& job) { return (unsigned long long*)&job.ticket[kQueueWord]; }
// the next fresh-frame ticket: the low half of the state word (what the waiting groups look at)
__device__ __forceinline__ unsigned draw_ticket(const FrameJob& job) { return (unsigned)atomicAdd(queue_state(job), 1ull); }
__device__ __forceinline__ unsigned long long* leave_word(const FrameJob& job) { return (unsigned long long*)&job.ticket[kLeaveWord]; }
// Frame ticket t -> its run of consecutive frames.  Runs are what gives a frame a hint worth trusting: inside a run a group encodes
// neighbours in time one after the other, and the answer of frame f - 1 is the best predictor of frame f there is (decoded video:
// scenes of similar frames).  With single-frame tickets handed out in index order a group's previous frame lies a whole round of the
// grid back -- 512 frames: another scene.  Long runs first, short ones at the end (the host sizes t4 / t2 so that every round of
// the grid is whole, psxhip_mdec_launch): guided self-scheduling, the tail of a launch is still balanced frame by frame.
__device__ __forceinline__ void ticket_run(const FrameJob& job, int t, int& first, int& len) {
    const int t4 = PSX_JOB_INT(t4), t2 = PSX_JOB_INT(t2), t42 = t4 + t2;
    if (t < t4) { first = 4 * t; len = 4; }
    else if (t < t42) { first = 4 * t4 + 2 * (t - t4); len = 2; }
    else { first = 4 * t4 + 2 * t2 + (t - t42); len = 1; }
}

// scalars[] slots (LDS, per workgroup).  [0, S_KEEP0) are per-frame: cleared when a frame ends; [S_KEEP0, S_COUNT) live across frames.
enum {
    S_DC_BITS = 0,      // v3: sum of the DC code lengths
    S_STG_NEXT,         // staging bump allocator (dwords)
    S_FOREIGN,          // the hint this frame started with came from somewhere else than its neighbour in time (the group's previous run, the previous launch): its value, 0 if none -- what the trust policy learns from when the answer is known
    S_CNT_F,            // count pass: sum of AC code lengths
    S_CNT_D,            // count pass: sum of refinement deficits
    S_EMIT_BITS,        // emit pass: sum of macroblock stream lengths
    S_EMIT_D,           // emit pass: sum of refinement deficits
    S_NNZ,              // emit pass: non-zero AC coefficients
    S_PASS_COUNT,       // next pass: count scale
    S_PASS_EMIT,        // next pass: emit scale
    S_DONE,             // search finished
    S_RESULT,           // chosen scale (64 = nothing fits)
    S_TOTAL_BITS,       // bits of the staged stream incl. the end-of-frame code
    S_PILOT_N,          // scales in this pilot round (0 = pilot finished)
    S_PILOT_GUESS,
    S_PILOT_LO,
    S_PILOT_HI,
    S_CK_DONE,          // checkpoint: macroblocks finished so far in this pass
    S_MB_NEXT,          // pass tickets: next macroblock ticket to hand out
    S_CK_WAVES,         // checkpoint: wavefronts whose sums up to the quarter mark are in
    S_ABORT,            // checkpoint verdict: new guess | pass number << 8 (a verdict of an earlier pass is stale, not reset)
    S_ABORTS_LEFT,      // checkpoints still allowed for this frame
    S_RETRY,            // this frame came from the retry queue: the scale to start from (0: a fresh frame; -1: the queue is empty)
    S_DEFER,            // 1: the frame goes to the retry queue instead of into another pass here; 2: it starts over from the pilot (S_REPILOT)
    S_PILOTED,          // the pilot has run for this frame (its guess is a measurement, not somebody else's answer)
    S_SPARE0,           // (unused)
    S_SEARCH,           // MdecSearch (14 ints)
    S_PILOT_SCALE0 = S_SEARCH + 14,    // [kPilotMax]
    S_PILOT_BITS0 = S_PILOT_SCALE0 + kPilotMax,   // [kPilotMax]
    S_TILE_FIRST0 = S_PILOT_BITS0 + kPilotMax,    // [kMaxTiles + 1] first macroblock whose stream starts in image tile t (see the merge)
    S_KEEP0 = S_TILE_FIRST0 + kMaxTiles + 1,
    S_FRAME = S_KEEP0,  // the frame TICKET in hand (a ticket is a run of 1, 2 or 4 consecutive frames: ticket_run())
    S_FIDX,             // index of the frame being encoded (inside the ticket's run, or a frame taken from the retry queue)
    S_RUN_LEFT,         // frames of the ticket's run behind the one being encoded
    S_PUSHED,           // the previous fresh frame of this group was handed on (see hand_on: whose hint the next frame starts from)
    S_QUEUE,            // drawn with the last frame's end when no fresh ticket is left: >= 0 the queue slot to take, -1 nothing will come, <= -2 wait for slot -2 - x
    S_HINT,             // the previous frame's answer in this group (0 = none): the pilot starts from it
    S_HINT_BUDGET,      // ... and its budget
    S_HINT_FRAME,       // ... and its index: the hint is the neighbour's answer when that is this frame's index - 1 (inside a run), foreign otherwise
    S_SHARED_HINT,      // answer | budget << 8 of the previous launch's last frame (by index)
    S_NEXT_DRAW,        // thread 0's ticket for the run after this one, parked here over the passes (it is a register from the draw to the start of the next frame's passes: the atomic's round trip hides behind a frame's work, and the passes have no register to spare)
    S_REPILOT,          // the first pass, started from a hint, was stopped with a verdict FAR from the hint (a scene cut): the verdict.  The frame starts over from the pilot: one more turn of the frame loop for the same frame (taken back to 0 once the pilot has read it)
    S_DISTRUST,         // foreign hints are not trusted: frames without a neighbour's answer run the pilot (trust policy, below): bit 0 the launches before this one found them wrong more than one time in four, bits 8.. this group's foreign hints that failed in a row
    S_F_TRIED,          // foreign hints this group could judge (the frame's answer became known here)
    S_F_WRONG,          // ... and how many of them were not the answer
    S_P_TRIED,          // frames of this group whose first pass started from the PILOT's guess
    S_P_WRONG,          // ... and how many of those guesses were not the answer
    S_COUNT
};
static_assert((S_SEARCH % 2) == 0, "MdecSearch is read and written as 64-bit pairs");

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
// the three-address form with the (wave-uniform) coefficient pair in a scalar register: the accumulator operand is not
// overwritten, so a chain's first product needs no copy of the rounding constant -- one register holds it for all chains.
// (Left to itself the compiler picks the two-address v_dot2c with a literal, and a v_mov per chain to seed it.)
__device__ __forceinline__ int dot2_k(uint32_t x, uint32_t k, int acc) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(k), "v"(acc));
    return r;
}
template <int IMM>     // ... seeded with an inline constant (-16..64)
__device__ __forceinline__ int dot2_ki(uint32_t x, uint32_t k) {
    static_assert(IMM >= -16 && IMM <= 64, "inline constant");
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(k), "n"(IMM));
    return r;
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}
constexpr uint32_t pk(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }

// ---------------------------------------------------------------------------------------------
// 8-point forward DCT, IJG "jfdctint" (Loeffler-Ligtenberg-Moschytz) for 8-bit samples as
// libavcodec specialises it (ff_jpeg_fdct_islow_8, the routine psxavenc's AVDCT call resolves to in
// the release configuration; mdec.c:640, .github/scripts/build.sh:36-56).  13-bit constants, 4
// fractional bits kept after the row pass.  COLUMN selects the output scaling of the second pass.
//
// Restated as integer linear forms (exact in the ring Z / 2^32, no intermediate rounding is moved):
// with the folded sums / differences  s_ij = d_i + d_j,  o0..o3 = d3-d4, d2-d5, d1-d6, d0-d7,
//     out0 = s07 + s16 + s25 + s34                out4 = s07 - s16 - s25 + s34
//     out2 = (s07 - s34) (K541 + K765) + (s16 - s25) K541
//     out6 = (s07 - s34) K541 + (s16 - s25) (K541 - K1847)
//     out7, 5, 3, 1 = the jfdctint odd part multiplied out: four coefficients per output
// every output is two v_dot2_i32_i16 on packed (int16, int16) operands, the rounding constant riding in the
// accumulator.  Operand ranges: row pass |s| <= 510, |o| <= 255; column pass |s|, |o| <= 32768 - 128 (row outputs
// are bounded by 16 * 8 * 128), so the packed int16 adds cannot wrap.  Inputs: P0 = (d0, d1), P1 = (d2, d3),
// R0 = (d7, d6), R1 = (d5, d4).  The row pass takes RAW pixels 0..255: the level shift by -128
// (mdec.c:627-632) only moves out0, by -8 * 128.
// tests/test_mdec_oracle.py::test_dct_linear_forms checks these forms against the oracle's butterfly.
// ---------------------------------------------------------------------------------------------
template <bool COLUMN>
__device__ __forceinline__ void fdct8_pk(uint32_t P0, uint32_t P1, uint32_t R0, uint32_t R1, int (&d)[8]) {
    constexpr int K_0_298 = 2446, K_0_390 = 3196, K_0_541 = 4433, K_0_765 = 6270, K_0_899 = 7373,
                  K_1_175 = 9633, K_1_501 = 12299, K_1_847 = 15137, K_1_961 = 16069, K_2_053 = 16819,
                  K_2_562 = 20995, K_3_072 = 25172;
    constexpr int SH = COLUMN ? 13 + 4 : 13 - 4;
    constexpr int RND = 1 << (SH - 1);
    // even part
    constexpr int A = K_0_541 + K_0_765, B = K_0_541, C = K_0_541 - K_1_847;
    // odd part: coefficient of o_j in out_i
    constexpr int C7_0 = K_0_298 - K_0_899 - K_1_961 + K_1_175, C7_1 = K_1_175, C7_2 = K_1_175 - K_1_961, C7_3 = K_1_175 - K_0_899;
    constexpr int C5_0 = K_1_175, C5_1 = K_2_053 - K_2_562 - K_0_390 + K_1_175, C5_2 = K_1_175 - K_2_562, C5_3 = K_1_175 - K_0_390;
    constexpr int C3_0 = K_1_175 - K_1_961, C3_1 = K_1_175 - K_2_562, C3_2 = K_3_072 - K_2_562 - K_1_961 + K_1_175, C3_3 = K_1_175;
    constexpr int C1_0 = K_1_175 - K_0_899, C1_1 = K_1_175 - K_0_390, C1_2 = K_1_175, C1_3 = K_1_501 - K_0_899 - K_0_390 + K_1_175;
    static_assert(A < 32768 && C > -32768 && C7_0 > -32768 && C1_3 < 32768 && C5_2 > -32768 && C3_1 > -32768, "int16 operands");

    const uint32_t S0 = pk_add(P0, R0);   // (s07, s16)
    const uint32_t S1 = pk_add(P1, R1);   // (s25, s34)
    const uint32_t D0 = pk_sub(P0, R0);   // (o3, o2)
    const uint32_t D1 = pk_sub(P1, R1);   // (o1, o0)

    const int rnd = RND;
    if (COLUMN) {
        d[0] = dot2_k(S0, pk(1, 1), dot2_ki<8>(S1, pk(1, 1))) >> 4;
        d[4] = dot2_k(S0, pk(1, -1), dot2_ki<8>(S1, pk(-1, 1))) >> 4;
    } else {
        d[0] = dot2_k(S0, pk(1, 1), dot2_ki<0>(S1, pk(1, 1))) * 16 - 8 * 128 * 16;
        d[4] = dot2_k(S0, pk(1, -1), dot2_ki<0>(S1, pk(-1, 1))) * 16;
    }
    d[2] = dot2_k(S0, pk(A, B), dot2_k(S1, pk(-B, -A), rnd)) >> SH;
    d[6] = dot2_k(S0, pk(B, C), dot2_k(S1, pk(-C, -B), rnd)) >> SH;
    d[7] = dot2_k(D0, pk(C7_3, C7_2), dot2_k(D1, pk(C7_1, C7_0), rnd)) >> SH;
    d[5] = dot2_k(D0, pk(C5_3, C5_2), dot2_k(D1, pk(C5_1, C5_0), rnd)) >> SH;
    d[3] = dot2_k(D0, pk(C3_3, C3_2), dot2_k(D1, pk(C3_1, C3_0), rnd)) >> SH;
    d[1] = dot2_k(D0, pk(C1_3, C1_2), dot2_k(D1, pk(C1_1, C1_0), rnd)) >> SH;
}

// The COLUMN pass with all eight outputs in ONE form: acc[i] such that out[i] = acc[i] >> 17, the rounding constant 2^16
// included.  Outputs 0 and 4 -- (sum + 8) >> 4 in jfdctint -- are (8192 * sum + 2^16) >> 17: the same number (a factor of
// 2^13 on numerator and denominator; |sum| <= 4 * 32640, so 8192 * sum stays below 2^31), which lets one register hold the
// rounding constant for all eight chains and lets the caller take every output from the HIGH half of its accumulator:
// (x >> 17) == ((x >> 16) >> 1), i.e. a byte permute that packs two high halves and one packed 16-bit shift per PAIR of
// outputs -- 8 instructions for the eight outputs where eight 32-bit shifts and four permutes were 12.
__device__ __forceinline__ void fdct8_col_acc(uint32_t P0, uint32_t P1, uint32_t R0, uint32_t R1, int (&a)[8]) {
    constexpr int K_0_298 = 2446, K_0_390 = 3196, K_0_541 = 4433, K_0_765 = 6270, K_0_899 = 7373,
                  K_1_175 = 9633, K_1_501 = 12299, K_1_847 = 15137, K_1_961 = 16069, K_2_053 = 16819,
                  K_2_562 = 20995, K_3_072 = 25172;
    constexpr int A = K_0_541 + K_0_765, B = K_0_541, C = K_0_541 - K_1_847;
    constexpr int C7_0 = K_0_298 - K_0_899 - K_1_961 + K_1_175, C7_1 = K_1_175, C7_2 = K_1_175 - K_1_961, C7_3 = K_1_175 - K_0_899;
    constexpr int C5_0 = K_1_175, C5_1 = K_2_053 - K_2_562 - K_0_390 + K_1_175, C5_2 = K_1_175 - K_2_562, C5_3 = K_1_175 - K_0_390;
    constexpr int C3_0 = K_1_175 - K_1_961, C3_1 = K_1_175 - K_2_562, C3_2 = K_3_072 - K_2_562 - K_1_961 + K_1_175, C3_3 = K_1_175;
    constexpr int C1_0 = K_1_175 - K_0_899, C1_1 = K_1_175 - K_0_390, C1_2 = K_1_175, C1_3 = K_1_501 - K_0_899 - K_0_390 + K_1_175;
    constexpr int E = 8192;
    const uint32_t S0 = pk_add(P0, R0);   // (s07, s16)
    const uint32_t S1 = pk_add(P1, R1);   // (s25, s34)
    const uint32_t D0 = pk_sub(P0, R0);   // (o3, o2)
    const uint32_t D1 = pk_sub(P1, R1);   // (o1, o0)
    const int rnd = 1 << 16;
    a[0] = dot2_k(S0, pk(E, E), dot2_k(S1, pk(E, E), rnd));
    a[4] = dot2_k(S0, pk(E, -E), dot2_k(S1, pk(-E, E), rnd));
    a[2] = dot2_k(S0, pk(A, B), dot2_k(S1, pk(-B, -A), rnd));
    a[6] = dot2_k(S0, pk(B, C), dot2_k(S1, pk(-C, -B), rnd));
    a[7] = dot2_k(D0, pk(C7_3, C7_2), dot2_k(D1, pk(C7_1, C7_0), rnd));
    a[5] = dot2_k(D0, pk(C5_3, C5_2), dot2_k(D1, pk(C5_1, C5_0), rnd));
    a[3] = dot2_k(D0, pk(C3_3, C3_2), dot2_k(D1, pk(C3_1, C3_0), rnd));
    a[1] = dot2_k(D0, pk(C1_3, C1_2), dot2_k(D1, pk(C1_1, C1_0), rnd));
}
// The same with the sixteen coefficient pairs handed in (in col_coeffs() order) -- the frame kernel loads them into scalar
// registers with one s_load_dwordx16 per macroblock: as literals they were fifteen s_mov per macroblock, as loop invariants
// sixteen more scalar registers for a kernel that already spills them.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
struct ColCoeffs { uint32_t k[16]; };
constexpr ColCoeffs col_coeffs() {
    constexpr int K_0_298 = 2446, K_0_390 = 3196, K_0_541 = 4433, K_0_765 = 6270, K_0_899 = 7373,
                  K_1_175 = 9633, K_1_501 = 12299, K_1_847 = 15137, K_1_961 = 16069, K_2_053 = 16819,
                  K_2_562 = 20995, K_3_072 = 25172;
    constexpr int A = K_0_541 + K_0_765, B = K_0_541, C = K_0_541 - K_1_847;
    constexpr int C7_0 = K_0_298 - K_0_899 - K_1_961 + K_1_175, C7_1 = K_1_175, C7_2 = K_1_175 - K_1_961, C7_3 = K_1_175 - K_0_899;
    constexpr int C5_0 = K_1_175, C5_1 = K_2_053 - K_2_562 - K_0_390 + K_1_175, C5_2 = K_1_175 - K_2_562, C5_3 = K_1_175 - K_0_390;
    constexpr int C3_0 = K_1_175 - K_1_961, C3_1 = K_1_175 - K_2_562, C3_2 = K_3_072 - K_2_562 - K_1_961 + K_1_175, C3_3 = K_1_175;
    constexpr int C1_0 = K_1_175 - K_0_899, C1_1 = K_1_175 - K_0_390, C1_2 = K_1_175, C1_3 = K_1_501 - K_0_899 - K_0_390 + K_1_175;
    constexpr int E = 8192;
    // pairs (inner product with S1 / D1 first, then with S0 / D0), outputs 0, 4, 2, 6, 7, 5, 3, 1 as in fdct8_col_acc()
    return ColCoeffs{{pk(E, E), pk(E, E), pk(-E, E), pk(E, -E), pk(-B, -A), pk(A, B), pk(-C, -B), pk(B, C),
                      pk(C7_1, C7_0), pk(C7_3, C7_2), pk(C5_1, C5_0), pk(C5_3, C5_2), pk(C3_1, C3_0), pk(C3_3, C3_2), pk(C1_1, C1_0), pk(C1_3, C1_2)}};
}
__device__ __forceinline__ void fdct8_col_acc_k(uint32_t P0, uint32_t P1, uint32_t R0, uint32_t R1, const u32x16& k, int (&a)[8]) {
    const uint32_t S0 = pk_add(P0, R0), S1 = pk_add(P1, R1), D0 = pk_sub(P0, R0), D1 = pk_sub(P1, R1);
    const int rnd = 1 << 16;
    a[0] = dot2_k(S0, k[1], dot2_k(S1, k[0], rnd));
    a[4] = dot2_k(S0, k[3], dot2_k(S1, k[2], rnd));
    a[2] = dot2_k(S0, k[5], dot2_k(S1, k[4], rnd));
    a[6] = dot2_k(S0, k[7], dot2_k(S1, k[6], rnd));
    a[7] = dot2_k(D0, k[9], dot2_k(D1, k[8], rnd));
    a[5] = dot2_k(D0, k[11], dot2_k(D1, k[10], rnd));
    a[3] = dot2_k(D0, k[13], dot2_k(D1, k[12], rnd));
    a[1] = dot2_k(D0, k[15], dot2_k(D1, k[14], rnd));
}
// (hi >> 17, lo >> 17) as packed int16 (lo in the low half): the two high halves side by side, each shifted once more
__device__ __forceinline__ uint32_t pack_sh17(int hi, int lo) {
    const uint32_t p = __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x07060302u);
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, p) >> (s16x2){1, 1}));
}

// ---------------------------------------------------------------------------------------------
// The ROW pass on the matrix pipe.  It is the same eight exact integer linear forms over the lane's 8 raw pixels -- an
// 8 x 8 integer matrix times a pixel vector -- so v_mfma_i32_32x32x16_i8 evaluates it exactly in int32: the pixels are one
// int8 digit (biased by -128: every form but out0 sums its coefficients to 0, and out0's 16 * 8 * 128 IS the level shift),
// the <= 15-bit coefficients two balanced digits c = 256 h + l.  A (32 x 16) is block-diagonal over the two k-groups: rows
// 0-7 / 8-15 = l / h digits acting on k 0-7, rows 16-23 / 24-31 the same on k 8-15; B (16 x 32): lane n holds the 8 pixels of
// row vector n % 32 in k-group n / 32, i.e. every lane hands in its own vector -- 64 vectors per instruction, a macroblock's
// 48 and 16 idle columns.  D: lane n < 32 receives outputs 0-3 of vectors n and 32 + n, lane 32 + n outputs 4-7 of the same
// two, both digits of an output in one lane: out = (h << 8) + l, then the pass's rounding shift.  17 + 5 VALU instructions
// per macroblock instead of 33, and the matrix pipe runs beside the other wavefronts' VALU work
// (tools/microbench/dct_rowpass_mfma.hip: 0 of 512 outputs differ from the dot2 form; 41.5 against 66.3 ns per 64 vectors per
// SIMD at 6 wavefronts per SIMD).  The COLUMN pass stays on v_dot2_i32_i16: its int16 inputs need two digits as well, three
// weight classes to combine and a digit split of the row pass's outputs -- no instruction is saved.
// ---------------------------------------------------------------------------------------------
typedef int i32x16 __attribute__((ext_vector_type(16)));
namespace rowm {
constexpr int K_0_298 = 2446, K_0_390 = 3196, K_0_541 = 4433, K_0_765 = 6270, K_0_899 = 7373, K_1_175 = 9633, K_1_501 = 12299,
              K_1_847 = 15137, K_1_961 = 16069, K_2_053 = 16819, K_2_562 = 20995, K_3_072 = 25172;
constexpr int A = K_0_541 + K_0_765, B = K_0_541, C = K_0_541 - K_1_847;
constexpr int C7_0 = K_0_298 - K_0_899 - K_1_961 + K_1_175, C7_1 = K_1_175, C7_2 = K_1_175 - K_1_961, C7_3 = K_1_175 - K_0_899;
constexpr int C5_0 = K_1_175, C5_1 = K_2_053 - K_2_562 - K_0_390 + K_1_175, C5_2 = K_1_175 - K_2_562, C5_3 = K_1_175 - K_0_390;
constexpr int C3_0 = K_1_175 - K_1_961, C3_1 = K_1_175 - K_2_562, C3_2 = K_3_072 - K_2_562 - K_1_961 + K_1_175, C3_3 = K_1_175;
constexpr int C1_0 = K_1_175 - K_0_899, C1_1 = K_1_175 - K_0_390, C1_2 = K_1_175, C1_3 = K_1_501 - K_0_899 - K_0_390 + K_1_175;
}  // namespace rowm
// the row pass's forms over the pixels d0..d7 (with o3 = d0 - d7, o2 = d1 - d6, o1 = d2 - d5, o0 = d3 - d4): the same numbers as
// fdct8_pk<false>'s dot products, written out per pixel
__constant__ int16_t c_dct_row[8][8] = {
    {16, 16, 16, 16, 16, 16, 16, 16},
    {rowm::C1_3, rowm::C1_2, rowm::C1_1, rowm::C1_0, -rowm::C1_0, -rowm::C1_1, -rowm::C1_2, -rowm::C1_3},
    {rowm::A, rowm::B, -rowm::B, -rowm::A, -rowm::A, -rowm::B, rowm::B, rowm::A},
    {rowm::C3_3, rowm::C3_2, rowm::C3_1, rowm::C3_0, -rowm::C3_0, -rowm::C3_1, -rowm::C3_2, -rowm::C3_3},
    {16, -16, -16, 16, 16, -16, -16, 16},
    {rowm::C5_3, rowm::C5_2, rowm::C5_1, rowm::C5_0, -rowm::C5_0, -rowm::C5_1, -rowm::C5_2, -rowm::C5_3},
    {rowm::B, rowm::C, -rowm::C, -rowm::B, -rowm::B, -rowm::C, rowm::C, rowm::B},
    {rowm::C7_3, rowm::C7_2, rowm::C7_1, rowm::C7_0, -rowm::C7_0, -rowm::C7_1, -rowm::C7_2, -rowm::C7_3},
};
// lane l's 8 bytes of A: row l % 32 (k-group row / 16, digit (row / 8) & 1, output row & 7), k = 8 * (l / 32) .. + 7
__device__ inline uint2 dct_row_a_operand(int lane) {
    const int row = lane & 31, kg = lane >> 5;
    const int grp = row >> 4, digit = (row >> 3) & 1, outp = row & 7;
    uint32_t w[2] = {0u, 0u};
    if (grp == kg)
        for (int k = 0; k < 8; k++) {
            const int c = (int)c_dct_row[outp][k];
            const int lo = ((c + 128) & 255) - 128, hi = (c - lo) >> 8;
            w[k >> 2] |= (uint32_t)((digit ? hi : lo) & 0xFF) << (8 * (k & 3));
        }
    return make_uint2(w[0], w[1]);
}
// b_lo / b_hi: the lane's 8 raw pixels (bytes).  o[0..3] / o[4..7]: this lane's four outputs (c = 4 * (lane >> 5) + r) of
// row vector lane & 31 / of row vector 32 + (lane & 31), in the form store_row_outputs() takes
__device__ __forceinline__ void fdct8_rows_mfma(uint32_t b_lo, uint32_t b_hi, uint2 a, int (&o)[8]) {
    const long av = (long)(((unsigned long long)a.y << 32) | a.x);
    const long bv = (long)(((unsigned long long)(b_hi ^ 0x80808080u) << 32) | (b_lo ^ 0x80808080u));
    i32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; r++) zero[r] = 0;
    const i32x16 acc = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, bv, zero, 0, 0, 0);
    // outputs 0 and 4 carry no fraction: x = (h << 8) + l.  The others are (x + 2^8) >> 9, an int16: left in the HIGH half of
    // o[] as ((x + 2^8) << 7) = (h << 15) + ((l << 7) + 2^15) -- two shift-adds, and the store takes the high half
    // (ds_write_b16_d16_hi), so the shift right costs nothing.  |x| < 2^24: no overflow.
#pragma unroll
    for (int v = 0; v < 2; v++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int l = acc[8 * v + r], h = acc[8 * v + 4 + r];
            if (r == 0) {
                o[4 * v + r] = (h << 8) + l;
            } else {
                // two v_lshl_add_u32.  The empty asm keeps `t` a value of its own (left alone the compiler shifts both digits and
                // adds three ways); the instructions themselves stay the compiler's, because it is the compiler that has to see
                // them read the matrix instruction's result registers and keep the wait states in front of them -- a hand-written
                // v_lshl_add_u32 in an asm statement was scheduled straight behind the v_mfma.
                uint32_t t = ((uint32_t)l << 7) + 0x8000u;
                asm volatile("" : "+v"(t));
                o[4 * v + r] = (int)(((uint32_t)h << 15) + t);
            }
        }
}
// the four outputs of one row vector -> the transpose tile (t: the vector's slot for output c = 4 * (lane >> 5), stride 8 per output)
__device__ __forceinline__ void store_row_outputs(int16_t* t, const int* o) {
    t[0] = (int16_t)o[0];
#pragma unroll
    for (int r = 1; r < 4; r++) t[r * 8] = (int16_t)(o[r] >> 16);
}

// A constant that is only stored (by one thread, once per frame) should be made where it is stored: hoisted out of the
// frame loop it occupies a register for the whole kernel, and at 80 registers that means a scratch spill.
__device__ __forceinline__ int in_loop(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
// Ballots written as the compare itself.  __builtin_amdgcn_ballot_w64(a != b) folds into one v_cmp only while that compare has no
// other user; one that is also a predicate elsewhere is turned into 0 / 1 and compared again (two more vector instructions).
__device__ __forceinline__ uint64_t ballot_ne0(int x) { uint64_t m; asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(m) : "v"(x)); return m; }
__device__ __forceinline__ uint64_t ballot_eq0(int x) { uint64_t m; asm("v_cmp_eq_u32_e64 %0, 0, %1" : "=s"(m) : "v"(x)); return m; }
template <int C> __device__ __forceinline__ uint64_t ballot_eq(int x) { uint64_t m; asm("v_cmp_eq_u32_e64 %0, %2, %1" : "=s"(m) : "v"(x), "n"(C)); return m; }

// wave-level ordering point for LDS traffic between lanes of the same wavefront
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// DC: divisor is always 16 (mdec.c:671), clamp to [-512, 510] (mdec.c:260-267).
__device__ __forceinline__ int quant_dc(int c0) {
    // sgn(c0) * ((|c0| + 8) >> 4)  ==  (c0 + 8 + (c0 >> 31)) >> 4  (round half away from zero; checked over the whole range
    // in tests/test_mdec_oracle.py::test_integer_identities_of_the_kernel)
    const int q = (c0 + 8 + (c0 >> 31)) >> 4;
    return q < -512 ? -512 : (q > 510 ? 510 : q);
}

struct Lds {
    uint32_t* out;          // [out_words]   frame output staging, dword j = output bytes 4j..4j+3 (pre-swizzle)
    uint32_t* stg;          // [stg_words]   macroblock bitstreams, each from a dword boundary, same bit order as `out`
    uint32_t* rec;          // [nmb]         per macroblock (encode order): staging dword offset | stream bits << 16
    uint32_t* mb_off;       // [nmb]         bit offset of each macroblock in the frame's stream
    int* dc_fn;             // [4 * DC chunks]  v3 DC chains: per 64-element chunk (thr, lo, hi) of its composite + the value entering it
    int16_t* dcv;           // [nmb*6]       per block, encode order: v2 the quantised DC; v3 the quantised DC, then (after the
                            //               chain scan) the DPCM delta -- codes are derived where they are needed
    uint16_t* ac_len16;     // [BS_LUT_SIZE] bits | refinement deficit << 8
    uint32_t* ac_code;      // [BS_LUT_SIZE] bits << 24 | deficit << 17 | code
    uint8_t* dc_plen;       // [16]
    uint8_t* dc_prefix;     // [16]
    uint8_t* qzz;           // [64] quant matrix in scan order
    uint4* tab_sel;         // [64] per lane: the two v_perm selectors of the pixel gather (PixelLane::sel), the lane's 8 bytes of the row pass's A matrix
    uint4* tab_pix;         // [64] per lane: pixel row offset, second-half offset, macroblock row shift, -
    uint8_t* tab_nat;       // [64] per lane k: where scan position k sits in a block of the coefficient tile (column * 8 + row)
    int16_t* tiles;         // per-wave DCT staging / code list
    int* scalars;           // [S_COUNT]
};

static_assert(sizeof(MdecSearch) == 56 && (S_SEARCH % 2) == 0, "MdecSearch lives in scalars[S_SEARCH..+14)");
static_assert(offsetof(FrameJob, batch) == 0 && offsetof(FrameJob, first) == sizeof(BatchDesc) * kMaxBatches && offsetof(FrameJob, n_batches) == sizeof(BatchDesc) * kMaxBatches + 4 * kMaxBatches, "batch_table() reads the table at the start of the kernel argument segment");
constexpr int kWaveTileBytes = ((6 * kTileStride * 2 + 6 * kZStride * 2) + 15) / 16 * 16;   // transpose tile + coefficient tile
static_assert(kWaveTileBytes >= 384 * 4, "the per-wave code list (384 entries) aliases the tiles");

__host__ __device__ inline int dc_chunks(int nmb) { return 2 * ((nmb + 63) >> 6) + ((4 * nmb + 63) >> 6); }

// LDS layout: everything whose size is known at compile time (given the workgroup shape) comes FIRST, at constant offsets --
// the compiler folds those addresses into the instructions' offset fields, where run-time offsets each took a scalar register
// (sixteen base addresses in a kernel that spills a hundred scalar registers) -- then the arrays sized by the geometry.
__host__ __device__ constexpr size_t lds_fixed_bytes(int waves) {
    size_t b = 0;
    b += (size_t)S_COUNT * 4;             // scalars
    b += BS_LUT_SIZE * 2;                 // ac_len16
    b = (b + 3) & ~(size_t)3;
    b += BS_LUT_SIZE * 4;                 // ac_code
    b += 32;                              // dc tables
    b = (b + 15) & ~(size_t)15;
    b += 2 * 64 * 16 + 2 * 64;            // per-lane constant tables, scan-position table, quant matrix
    b += (size_t)waves * kWaveTileBytes;  // tiles
    return (b + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t lds_bytes(int nmb, int out_words, int stg_words, int waves) {
    size_t b = lds_fixed_bytes(waves);
    b += (size_t)out_words * 4;
    b += (size_t)stg_words * 4;
    b += (size_t)nmb * 4;         // rec
    b += (size_t)nmb * 4;         // mb_off
    b += (size_t)dc_chunks(nmb) * 16;   // dc_fn
    b += (size_t)nmb * 6 * 2;     // dcv
    return (b + 15) & ~(size_t)15;
}
template <int WAVES>
__device__ __forceinline__ Lds carve(char* base, int nmb, int out_words, int stg_words) {
    Lds L;
    size_t b = 0;
    L.scalars = (int*)(base + b);         b += (size_t)S_COUNT * 4;
    L.ac_len16 = (uint16_t*)(base + b);   b += BS_LUT_SIZE * 2;                    b = (b + 3) & ~(size_t)3;
    L.ac_code = (uint32_t*)(base + b);    b += BS_LUT_SIZE * 4;
    L.dc_plen = (uint8_t*)(base + b);     b += 16;
    L.dc_prefix = (uint8_t*)(base + b);   b += 16;                                 b = (b + 15) & ~(size_t)15;
    L.tab_sel = (uint4*)(base + b);       b += 64 * 16;
    L.tab_pix = (uint4*)(base + b);       b += 64 * 16;
    L.tab_nat = (uint8_t*)(base + b);     b += 64;
    L.qzz = (uint8_t*)(base + b);         b += 64;
    L.tiles = (int16_t*)(base + b);       b += (size_t)WAVES * kWaveTileBytes;     b = (b + 15) & ~(size_t)15;
    L.out = (uint32_t*)(base + b);        b += (size_t)out_words * 4;
    L.stg = (uint32_t*)(base + b);        b += (size_t)stg_words * 4;
    L.rec = (uint32_t*)(base + b);        b += (size_t)nmb * 4;
    L.mb_off = (uint32_t*)(base + b);     b += (size_t)nmb * 4;
    L.dc_fn = (int*)(base + b);           b += (size_t)dc_chunks(nmb) * 16;
    L.dcv = (int16_t*)(base + b);
    return L;
}

// OR `len` bits (value `v`, MSB first) into the staging buffer at bit position `pos` of the
// bitstream.  Staging dword j holds stream bits [32j, 32j+32) with bit 32j in its MSB.
__device__ __forceinline__ void put_bits(uint32_t* words, uint32_t pos, int len, uint32_t v) {
    const uint32_t w = pos >> 5, sh = pos & 31;
    const uint32_t top = v << (32 - len);                              // the code left-aligned (1 <= len <= 32)
    const uint32_t hi = top >> sh, lo = __builtin_amdgcn_alignbit(top, 0u, sh);   // lo = sh ? top << (32 - sh) : 0
    atomicOr(&words[w], hi);
    if (lo) atomicOr(&words[w + 1], lo);
}
// The same into the LDS staging area at dword `base_dw` (wave-uniform) + bit `bit`: the address written out -- one scalar
// shift-add for the base, a shift and a shift-add per lane (the compiler's own version of "(bit >> 5) * 4" is shift, mask, add).
__device__ __forceinline__ void put_bits_lds(uint32_t* stg, uint32_t base_dw, uint32_t bit, int len, uint32_t v) {
    typedef uint32_t __attribute__((address_space(3))) * LdsWord;
    uint32_t sb, a;
    asm("s_lshl2_add_u32 %0, %2, %3\n\tv_lshrrev_b32 %1, 5, %4\n\tv_lshl_add_u32 %1, %1, 2, %0"
        : "=&s"(sb), "=&v"(a) : "s"(base_dw), "s"((uint32_t)(uintptr_t)stg), "v"(bit) : "scc");
    const LdsWord wp = (LdsWord)(uintptr_t)a;
    const uint32_t top = v << (32 - len);
    const uint32_t hi = top >> (bit & 31), lo = __builtin_amdgcn_alignbit(top, 0u, bit & 31);
    __hip_atomic_fetch_or(wp, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lo) __hip_atomic_fetch_or(wp + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Per-lane constants of the AC path: lane k owns zig-zag position k.
struct LaneConst {
    int quant;          // quant matrix entry at this zig-zag position
    int lane;           // (masks and offsets derived from it are made where they are used: 80 registers)
    int zsrc;           // where this lane's scan position sits in a block of the coefficient tile
};

// ---------------------------------------------------------------------------------------------
// Quantiser.  The reference computes (int)round((double)n / (double)d)  (mdec.c:438): round half
// away from zero, i.e.  sgn(n) * floor((2|n| + d) / (2d)) = sgn(n) * floor(|n| / d + 1/2).  With
// N = 2|n| + d and D = 2d, floor(N / D) == floor((N + 0.5) / D), and (N + 0.5) / D is at least 0.5 / D away
// from every integer.  The kernel evaluates  trunc(fma(|n|, 1/d, 0.5 + 0.25/d))  in fp32 -- the same real
// number -- with 1/d the correctly rounded reciprocal: the result is within (N + 0.5) * 1.5 * 2^-23 / D +
// 2^-25 of it, which is < 0.5 / D for every reachable operand (N < 2^17), so truncation gives the exact
// floor (tests/test_mdec_oracle.py::test_fp32_reciprocal_quantiser_is_exact checks every operand, also
// with the reciprocal off by +-2 ulp).  `cf` = float(n) (the sign is dropped by the fma's |.| modifier),
// 0 on lane 0 (the DC slot never carries an AC code).
// ---------------------------------------------------------------------------------------------
struct QuantK {
    float inv;    // 1 / (quant * scale)
    float bias;   // 0.5 + 0.25 / (quant * scale)
};
__device__ __forceinline__ QuantK make_quant(int quant, int scale) {
    QuantK k;
    k.inv = 1.0f / (float)(quant * scale);      // IEEE division
    k.bias = __builtin_fmaf(0.25f, k.inv, 0.5f);
    return k;
}
__device__ __forceinline__ int quant_mag(float cf, const QuantK& k) {
    return (int)__builtin_fmaf(__builtin_fabsf(cf), k.inv, k.bias);
}

// number of zero coefficients between this lane and the previous non-zero one (bit 0 of the mask, the
// DC slot, is the sentinel): lane - 1 - (63 - clz(mask below me))
__device__ __forceinline__ int run_before(uint64_t nz_mask, const LaneConst& lc) {
    const int ln = in_loop(lc.lane);
    const uint64_t prev = (nz_mask | 1ull) & ((1ull << ln) - 1ull);
    return __clzll((long long)prev) + (ln - 64);
}

// index into the padded LUTs: row = min(|level|, MAX_LEVEL + 1), column = run (0..62).  Level 0 is row 0
// = 0 bits, so silent lanes need no predicate.  Only the LENGTH needs no level clamp at 510/512
// (mdec.c:260-267): every level > MAX_LEVEL is an escape of the same length.
__device__ __forceinline__ int lut_index(int q, int run) {
    const unsigned qc = (unsigned)q > (unsigned)(BS_LUT_MAX_LEVEL + 1) ? (unsigned)(BS_LUT_MAX_LEVEL + 1) : (unsigned)q;
    return (int)__umul24(qc, (unsigned)BS_LUT_W) + run;     // full-rate 24-bit multiply (v_mul_lo_u32 is quarter rate)
}

// AC code lengths of the six blocks of a macroblock at one scale: returns per lane  sum(bits) | sum(deficit) << 8
// (<= 6 * 22 and <= 6 * 9: the fields cannot carry into each other)
__device__ __forceinline__ int count_mb(const float (&cf)[6], const QuantK& k, const LaneConst& lc, const uint16_t* ac_len16) {
    int acc = 0;
#pragma unroll
    for (int b = 0; b < 6; b++) {
        const int q = quant_mag(cf[b], k);
        acc += (int)ac_len16[lut_index(q, run_before(wave::ballot(q != 0), lc))];
    }
    return acc;
}

// DC code of one block: v2 = the 10-bit value (mdec.c:451-453); v3 = VLC of the DPCM delta: size class = magnitude
// bits, then a sign-dependent offset (mdec.c:285-318).
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

// ---------------------------------------------------------------------------------------------
// Macroblock visiting order.  A macroblock's bitstream is built independently of its position in the frame's stream
// (staging + merge below), so the visiting order is free: wavefront w takes the macroblocks w, w + W, w + 2W, ... in
// RASTER order, i.e. at any time the W wavefronts of a group work on W horizontally adjacent macroblocks and share
// the frame's cache lines (a 128-byte line spans 8 macroblocks of a luma row).  The index in ENCODE order (fx outer,
// fy inner, mdec.c:689-690) is what the stream layout needs.  (fx, fy) advance incrementally, no divisions in the loop.
// ---------------------------------------------------------------------------------------------
struct MbCursor {
    int fx, fy;
};
__device__ __forceinline__ MbCursor mb_cursor(int raster_index, int nx) {
    MbCursor c;
    c.fy = raster_index / nx;
    c.fx = raster_index - c.fy * nx;
    return c;
}
template <int WAVES>
__device__ __forceinline__ void mb_advance(MbCursor& c, int nx) {
    c.fx += WAVES;
    while (c.fx >= nx) { c.fx -= nx; c.fy++; }
}

// Source bytes of a macroblock (mdec.c:619-633).  Lane t < 48 = (block t>>3, row t&7) fetches its own
// 8 pixels straight from the frame (no LDS staging): a luma row is 8 contiguous bytes, a chroma row is
// 16 bytes of interleaved Cr,Cb (NV21: Cr at even bytes, Cb at odd).  All offsets are 32-bit (a frame is
// < 2^31 bytes).
struct PixelLane {
    uint32_t lane_off;       // offset of this lane's pixel row inside macroblock (0, 0)
    uint32_t hi_off;         // chroma rows are 16 bytes long: second half
    uint32_t mb_row_shift;   // bytes per macroblock row for this lane's plane = 8 W << shift
    uint32_t sel[2];         // v_perm selectors building the lane's 8 pixels as bytes p0..p3, p4..p7
};
__device__ __forceinline__ PixelLane pixel_lane(int lane, int W, int H) {
    PixelLane p;
    const int blk = lane >> 3, r8 = lane & 7;
    const bool is_chroma = blk < 2;
    if (is_chroma) p.lane_off = (uint32_t)W * (uint32_t)H + (uint32_t)r8 * (uint32_t)W;
    else p.lane_off = ((uint32_t)(((blk - 2) >> 1) * 8 + r8)) * (uint32_t)W + (uint32_t)((blk - 2) & 1) * 8u;
    if (lane >= 48) p.lane_off = 0;                 // idle lanes read the frame's first bytes (unused)
    p.hi_off = is_chroma ? 8u : 0u;
    p.mb_row_shift = is_chroma ? 0u : 1u;
    // v_perm_b32(hi, lo, sel): result byte i = byte sel[i] of the 8-byte pool {lo: 0..3, hi: 4..7}.
    // Pools (see dct_mb): first four pixels from {plo.x, plo.y}, last four from {phi.x, phi.y}.
    //   luma: the row's 8 pixels are plo.x, plo.y (phi = plo: the second fetch has offset 0).  chroma: 16 bytes plo.x, plo.y,
    //   phi.x, phi.y with this block's samples at even (Cr, blk 0) or odd (Cb, blk 1) bytes.
    if (!is_chroma) {
        p.sel[0] = 0x03020100u;    // p0..p3 = plo.x
        p.sel[1] = 0x07060504u;    // p4..p7 = phi.y = plo.y
    } else {
        const uint32_t o = (uint32_t)blk;   // 0: even bytes, 1: odd bytes
        p.sel[0] = 0x06040200u + o * 0x01010101u;
        p.sel[1] = 0x06040200u + o * 0x01010101u;
    }
    return p;
}

// step function x -> (x < thr ? lo : hi) and its ordered scan over the 64 lanes (DPP only, no LDS): lane i ends up with
// f_0 then f_1 ... then f_i composed.  Hillis-Steele inside each 16-lane row, then row_bcast:15 / row_bcast:31.
struct StepFn {
    int thr, lo, hi;
};
__device__ __forceinline__ StepFn compose(const StepFn& earlier, const StepFn& later) {
    StepFn r;
    r.thr = earlier.thr;
    r.lo = earlier.lo < later.thr ? later.lo : later.hi;
    r.hi = earlier.hi < later.thr ? later.lo : later.hi;
    return r;
}
template <int CTRL>
__device__ __forceinline__ StepFn dpp_stepfn(const StepFn& f) {
    StepFn e;
    e.thr = __builtin_amdgcn_update_dpp(f.thr, f.thr, CTRL, 0xF, 0xF, false);
    e.lo = __builtin_amdgcn_update_dpp(f.lo, f.lo, CTRL, 0xF, 0xF, false);
    e.hi = __builtin_amdgcn_update_dpp(f.hi, f.hi, CTRL, 0xF, 0xF, false);
    return e;
}
__device__ __forceinline__ void scan_stepfn(StepFn& f, int lane) {
    StepFn e;
    e = dpp_stepfn<0x111>(f); if ((lane & 15) >= 1) f = compose(e, f);    // row_shr:1
    e = dpp_stepfn<0x112>(f); if ((lane & 15) >= 2) f = compose(e, f);    // row_shr:2
    e = dpp_stepfn<0x114>(f); if ((lane & 15) >= 4) f = compose(e, f);    // row_shr:4
    e = dpp_stepfn<0x118>(f); if ((lane & 15) >= 8) f = compose(e, f);    // row_shr:8
    e = dpp_stepfn<0x142>(f); if (lane & 16) f = compose(e, f);           // row_bcast:15 -> rows 1 and 3
    e = dpp_stepfn<0x143>(f); if (lane >= 32) f = compose(e, f);          // row_bcast:31 -> rows 2 and 3
}

// ---------------------------------------------------------------------------------------------
// Register budget: the small shape is compiled for 6 wavefronts per SIMD (<= 80 VGPRs), i.e. two frames in
// flight per CU.  The hot path is latency-bound (LDS look-ups, DPP scans, ballots), so it is written
// as short per-block bodies that rely on wave interleaving rather than on wide unrolled bodies.
// ---------------------------------------------------------------------------------------------
// STATS: the diagnostics instantiation (pass counters, per-group trace, per-phase clocks; PSXHIP_MDEC_STATS=1).  The production
// instantiation carries none of it -- not even the branches.
template <int CODEC, int WAVES, int OCC, bool STATS>
__global__ __launch_bounds__(WAVES * 64, OCC) void mdec_encode_frames_kernel(const FrameJob job) {
    constexpr int kWavesPerGroup = WAVES;
    constexpr int kThreads = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Lds L = carve<WAVES>(smem, job.nmb, job.out_words, job.stg_words);

    const int tid = (int)threadIdx.x;
    unsigned long long t_entry = 0, t_first = 0;
    if (STATS) t_entry = wall_clock64();
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nmb = job.nmb, nx = job.nx, ny = job.ny, W = job.width, H = job.height;
    const int nblk = nmb * 6;
    // the pilot's sample: one macroblock per wavefront, two on frames of 600 macroblocks and more (a dozen of 1200 is 1 %: at 640x480 the
    // pilot was right 60 % of the time) and in the 16-wavefront shape; macroblock i of the sample belongs to wavefront i % waves
    constexpr int kPilotPerWave = 2;
    const int pilot_rounds = (WAVES == kWavesLarge || nmb >= 600) ? 2 : 1;
    const int n_pilot = nmb < kWavesPerGroup * pilot_rounds ? nmb : kWavesPerGroup * pilot_rounds;

    // ---- once per workgroup: LUTs into LDS, per-lane constants.  All global loads are issued before the first result is
    //      waited for: one round trip to the L2 instead of seven in a row (3 us of prologue per group otherwise)
    constexpr int kLutTrips = (BS_LUT_SIZE + kThreads - 1) / kThreads;
    uint16_t lut_len[kLutTrips];
    uint32_t lut_code[kLutTrips];
#pragma unroll
    for (int j = 0; j < kLutTrips; j++) {
        const int i = tid + j * kThreads;
        lut_len[j] = i < BS_LUT_SIZE ? c_ac_len16[i] : (uint16_t)0;
        lut_code[j] = i < BS_LUT_SIZE ? c_ac_code[i] : 0u;
    }
    const uint8_t pro_plen = c_dc_plen[(tid >> 3) & 1][tid & 7], pro_prefix = c_dc_prefix[(tid >> 3) & 1][tid & 7];
    const uint8_t pro_qzz = c_quant_zz[tid & 63], pro_zagzig = c_zagzig[tid & 63];
    const unsigned pro_shared_hint = *job.hint, pro_distrust = job.hint[kDistrustWord];
#pragma unroll
    for (int j = 0; j < kLutTrips; j++) {
        const int i = tid + j * kThreads;
        if (i < BS_LUT_SIZE) { L.ac_len16[i] = lut_len[j]; L.ac_code[i] = lut_code[j]; }
    }
    if (tid < 16) {
        L.dc_plen[tid] = pro_plen;
        L.dc_prefix[tid] = pro_prefix;
    }
    int16_t* tileT = L.tiles + (size_t)wid * (kWaveTileBytes / 2);    // [6][kTileStride] row-pass output, transposed
    int16_t* tileZ = tileT + 6 * kTileStride;                          // [6][kZStride] coefficients, column-major within a block
    uint32_t* clist = (uint32_t*)tileT;                                // code list of a macroblock (aliases both tiles)

    // Per-lane constants of the pixel gather (seven registers each lane would otherwise carry -- or spill to scratch -- for
    // the whole kernel) and the scan-order table: the same for every wavefront and every frame, they live in LDS tables and
    // are read where they are used.
    {
        if (tid < 64) {
            L.qzz[tid] = pro_qzz;
            const PixelLane p = pixel_lane(tid, W, H);
            const uint2 am = dct_row_a_operand(tid);
            L.tab_sel[tid] = make_uint4(p.sel[0], p.sel[1], am.x, am.y);
            L.tab_pix[tid] = make_uint4(p.lane_off, p.hi_off, p.mb_row_shift, 0u);
            const int raster = (int)pro_zagzig;
            L.tab_nat[tid] = (uint8_t)((raster & 7) * 8 + (raster >> 3));
        }
        __syncthreads();
    }

    // Two groups share a CU in the small shape.  The SIMD arbiter serves the OLDER wavefront first (priority, then age),
    // so left alone the group that arrived first runs at full speed and its partner on the leftovers -- and the partner
    // ends up finishing its last frame alone on a half-empty CU.  Wave slot numbers tell the two groups apart (the
    // first group on a SIMD holds the low slots); the groups take turns at raised priority, one macroblock at a time.
    const unsigned hw_slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) ;   // HW_REG_HW_ID[3:0] = wave slot on its SIMD
    const unsigned prio_bits = (WAVES == kWavesSmall && hw_slot >= (unsigned)(kWavesSmall / 4)) ? (job.prio_pattern >> 8) & 0xFFu : job.prio_pattern & 0xFFu;
    const unsigned prio_bits4 = prio_bits * 0x01010101u;      // the pattern four times over: bit (iteration & 31) is bit (iteration & 7)
    if (tid == 0) {
        L.scalars[S_HINT] = 0; L.scalars[S_HINT_BUDGET] = 0; L.scalars[S_HINT_FRAME] = -2; L.scalars[S_SHARED_HINT] = (int)pro_shared_hint;
        L.scalars[S_DISTRUST] = (int)pro_distrust; L.scalars[S_F_TRIED] = 0; L.scalars[S_F_WRONG] = 0; L.scalars[S_P_TRIED] = 0; L.scalars[S_P_WRONG] = 0;
        L.scalars[S_PUSHED] = 0; L.scalars[S_QUEUE] = 0; L.scalars[S_NEXT_DRAW] = 0; L.scalars[S_REPILOT] = 0;
    }
    unsigned pass_sum = 0, pass_hist[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t_start = 0, t_mark = 0, phase_ticks[6] = {0, 0, 0, 0, 0, 0};
    auto mark = [&](int phase) {       // diagnostics: time since the previous mark goes to `phase`
        if (STATS && tid == 0) {
            const unsigned long long now = wall_clock64();
            phase_ticks[phase] += now - t_mark;      // summed up in registers, added to the global sums when the group leaves
            t_mark = now;
        }
    };
    unsigned long long wait_ticks = 0, wait_cat[6] = {0, 0, 0, 0, 0, 0};
    auto group_sync = [&](int c) {          // diagnostics: time each wavefront spends waiting at group barriers
        if (STATS) {
            const unsigned long long a = wall_clock64();
            __syncthreads();
            const unsigned long long w = wall_clock64() - a;
            wait_ticks += w;
            wait_cat[c] += w;
        } else {
            __syncthreads();
        }
    };
    // Bit offsets of the macroblocks: exclusive scan in encode order (one wavefront)
    auto scan_offsets = [&](int lane) {
        // ... and, for images of several tiles, the first macroblock that STARTS in each tile (streams are in encode
        // order, a macroblock's stream is shorter than a tile): tile t is then fed by macroblocks
        // [tile_first[t] - 1, tile_first[t + 1]) -- the one before may straddle into it
        if (lane <= kMaxTiles) L.scalars[S_TILE_FIRST0 + lane] = in_loop(nmb);
        uint32_t carry = 0;
        int prev_tile = -1;
        for (int base = 0; base < nmb; base += 64) {
            const int mbe = base + lane;
            const int bits = mbe < nmb ? (int)(L.rec[mbe] >> 16) : 0;
            const int incl = wave::inclusive_scan_add(bits);
            const uint32_t D = carry + (uint32_t)(incl - bits);
            if (mbe < nmb) L.mb_off[mbe] = D;
            int tile = mbe < nmb ? (int)((2u + (D >> 5)) / (uint32_t)job.out_tile) : 0x7FFF;
            const int before = __builtin_amdgcn_update_dpp(prev_tile, tile, 0x138, 0xF, 0xF, false);   // wave_shr:1, lane 0 <- carry
            if (mbe < nmb && tile != before && tile <= kMaxTiles) L.scalars[S_TILE_FIRST0 + tile] = mbe;
            prev_tile = __builtin_amdgcn_readlane(tile, 63);
            carry += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
        }
    };
    // small frames: wavefront 1 scans after every emit pass, while thread 0 decides what happens next (the offsets are needed
    // if this pass's stream turns out to be the answer; otherwise the next emit pass's scan overwrites them) -- one barrier
    // and one single-wavefront phase fewer per frame.  Large frames (a scan of 19 chunks at 640x480) scan once, when the
    // answer is known.
    const bool scan_early = nmb <= 512;
    int n_done = 0;
    int carry_pass = 0, carry_guess0 = 0, carry_abort = 0;      // diagnostics: passes and first guess of a frame's first attempt (a frame sent back to the pilot)
    if (STATS) t_start = t_mark = wall_clock64();
    // thread 0: the ticket for the frame after next, drawn one frame ahead and kept AS DRAWN (ticket - gridDim.x): nothing is computed
    // from it before the next frame's decisions, so the atomic's round trip hides behind a frame's work (adding gridDim.x -- a
    // scalar load -- on the spot made the compiler wait for the atomic right there, at every frame's end)
    unsigned next_draw = 0;
    // draws below this are real tickets (the first gridDim.x tickets are the groups' own; a launch may have more groups than tickets --
    // the surplus only ever takes frames from the retry queue)
    auto fresh_draws = [&]() -> unsigned { const unsigned nt = (unsigned)PSX_JOB_INT(n_tickets); return nt > gridDim.x ? nt - gridDim.x : 0u; };     // (made where it is asked for: hand_on, rarely)
    // a group's first frame is its own index (no waiting for an atomic every group issues at the same moment); the counter
    // hands out the frames after those
    if (tid == 0) {
        // Workgroup b runs on XCD b % 8 (observed, not promised), so "first frame = b" hands XCD x exactly the frames = x (mod 8),
        // and content whose cost has a period of 8 frames -- the synthetic generator shifts its chroma pattern one pixel per
        // frame, i.e. it re-aligns with the 8x8 block grid every 8 frames -- piles every expensive frame of the first round
        // on one XCD (traced on noise +-8: the groups of XCD 0 ended at 222 us on average, the others at 172).  Inside each
        // octet of workgroups the frame <-> workgroup assignment is rotated by the octet's number: every XCD sees all eight
        // phases, and still a time-interleaved sample of the batch (which is what real, slowly varying video wants).
        unsigned f0 = blockIdx.x;
        if ((f0 | 7u) < gridDim.x) f0 = (f0 & ~7u) + (((f0 & 7u) + (f0 >> 3)) & 7u);
        const int n_tickets = PSX_JOB_INT(n_tickets);
        L.scalars[S_FRAME] = (int)f0 < n_tickets ? (int)f0 : in_loop(kNoTicket);
        // A group draws the ticket for its NEXT run when it enters the LAST frame of the run in hand -- one frame ahead of the
        // need, and not before: "this group holds a further fresh frame" (what allows it to hand a frame on, hand_on) is then
        // "frames left in the run, or the ticket in hand is a real one", and the draw that comes up blank still marks the moment
        // the group enters its last fresh frame, which is what the retry queue's protocol counts on.
        int first = 0, len = 1;
        if ((int)f0 < n_tickets) ticket_run(job, (int)f0, first, len);
        L.scalars[S_FIDX] = first;
        L.scalars[S_RUN_LEFT] = len - 1;
        if ((int)f0 >= n_tickets) {
            // more groups than tickets (a launch of at most one run per group whose frames may be handed on): this group only ever
            // takes frames from the queue; it draws its place there right away
            if (job.retry) {
                const unsigned long long w = atomicAdd(queue_state(job), 1ull << kQueueHeadShift);
                const unsigned h = (unsigned)(w >> kQueueHeadShift) & kQueueMask, reserved = (unsigned)(w >> kQueueReservedShift) & kQueueMask;
                L.scalars[S_QUEUE] = reserved > h ? (int)h : (unsigned)w >= (unsigned)n_tickets ? -1 : -2 - (int)h;
            }
        } else if (len == 1) {
            next_draw = draw_ticket(job);
        }
        if (job.retry) atomicAdd(&job.ticket[kStartedWord], 1u);      // (nobody waits for a frame to be handed on before all groups are here)
    }
    // Per-frame state (frame image tile, staging area, scalars) is cleared, and the next ticket published, while a frame's
    // last tile is written out: the barrier that ends a frame is also the one that starts the next.
    auto end_of_frame = [&](int tid, bool parked = true) {
        for (int i = tid; i < job.stg_words; i += kThreads) L.stg[i] = 0u;
        if (tid < S_KEEP0) L.scalars[tid] = 0;
        if (tid == 0) {
            // tickets hand runs out in order, so workgroups that draw cheap frames simply draw more
            if (parked) next_draw = (unsigned)L.scalars[S_NEXT_DRAW];
            const int left = L.scalars[S_RUN_LEFT];
            bool entered_last;                 // the frame this group moves to is the last of its run
            const int held = L.scalars[S_FRAME];
            if (left > 0 && held != kNoTicket) {
                L.scalars[S_RUN_LEFT] = left - 1;
                L.scalars[S_FIDX] = L.scalars[S_FIDX] + 1;
                entered_last = left == 1;
            } else {
                const int n_tickets = PSX_JOB_INT(n_tickets);
                int t = held != kNoTicket ? (int)(next_draw + gridDim.x) : in_loop(kNoTicket);    // (a group that only takes frames from the queue holds no ticket)
                if (t >= n_tickets) t = in_loop(kNoTicket);
                L.scalars[S_FRAME] = t;
                entered_last = false;
                if (t != kNoTicket) {
                    int first, len;
                    ticket_run(job, t, first, len);
                    L.scalars[S_FIDX] = first;
                    L.scalars[S_RUN_LEFT] = len - 1;
                    entered_last = len == 1;
                } else if (job.retry) {
                    // no fresh frame left for this group: in place of the ticket it draws its place in the retry queue (see the top of
                    // the frame loop), in the shadow of the same write-out
                    L.scalars[S_RUN_LEFT] = 0;
                    const unsigned long long w = atomicAdd(queue_state(job), 1ull << kQueueHeadShift);
                    const unsigned h = (unsigned)(w >> kQueueHeadShift) & kQueueMask, reserved = (unsigned)(w >> kQueueReservedShift) & kQueueMask;
                    L.scalars[S_QUEUE] = reserved > h ? (int)h : (unsigned)w >= (unsigned)n_tickets ? -1 : -2 - (int)h;
                }
            }
            if (entered_last) next_draw = draw_ticket(job);
        }
    };
    for (int i = tid; i < job.out_tile + 1; i += kThreads) L.out[i] = 0u;
    for (int i = tid; i < job.stg_words; i += kThreads) L.stg[i] = 0u;
    if (tid < S_KEEP0) L.scalars[tid] = 0;
    __syncthreads();
    for (;;) {
        // thread- and lane-derived values (loop bases, masks, LDS addresses, even the predicate "thread 0") are re-derived per
        // frame from an opaque copy of the thread index: hoisted out of the frame loop they would be spilled to scratch once per
        // wavefront (4 KB each, 25 MB per 1000 frames at two frames per group)
        int tid_f = (int)threadIdx.x;
        asm volatile("" : "+v"(tid_f));
        const int tid = tid_f;               // shadows the kernel-scope copy on purpose
        int f = L.scalars[S_FIDX];
        if (L.scalars[S_FRAME] == kNoTicket) {
            // No fresh frame left for this group: frames that other groups handed on instead of running another pass over them
            // (see the end of the pass loop).  Everything here goes through read-modify-write atomics -- the queue is shared by
            // groups on all XCDs, whose L2s are not coherent for plain loads.  A group leaves when it finds the queue empty;
            // whoever pushes later still holds a fresh frame and will come through here itself.
            if (!job.retry || L.scalars[S_QUEUE] == -1) break;
            __syncthreads();           // everybody has read S_FRAME
            if (tid == 0) {
                // Pop ticket h owns queue slot h.  The state word holds, under one atomic, the pop tickets drawn, the slots
                // reserved, and the fresh-frame tickets drawn.  Only a group that holds a further fresh ticket hands a frame on, and
                // every group draws exactly one ticket that lies past the batch (it stops drawing then): once n_frames tickets are
                // out -- n_frames - grid real ones and one blank per group -- nobody can push any more.  So slot h will be filled
                // iff reserved > h, and never once the ticket count has reached n_frames with reserved <= h.  Until either holds
                // the group waits -- it has nothing else to do.
                // Nothing says all groups of the launch are resident at once (another context's kernel may share the device), and a
                // group that has not started draws no ticket -- it may even be waiting for the place this group holds.  So a group
                // waits only once every group of the launch has started, and not for ever; a group that leaves marks its slot
                // abandoned on the way out, and whoever reserves that slot later learns it from the exchange and keeps its frame.
                // (Two contexts' launches sharing the GPU ran 4x slower while waiting groups sat out their patience.)
                int got = -1;
                // (a frame taken from the queue that starts over from the pilot -- S_REPILOT, at the end of the frame loop -- comes through
                //  here again: it is still in hand.  The test stands HERE, with thread 0 on the cold side: as a second condition of the
                //  branch above it cost the 16-wavefront shape 1.7 % on every frame)
                if (L.scalars[S_REPILOT]) {
                    got = L.scalars[S_RETRY];
                } else {
                const int q = L.scalars[S_QUEUE];
                const unsigned h = q >= 0 ? (unsigned)q : (unsigned)(-2 - q);
                bool there = q >= 0;
                for (int looks = 0; !there; looks++) {
                    const unsigned long long w = queue_peek(queue_state(job), looks);
                    const unsigned started = queue_peek(&job.ticket[kStartedWord], looks);
                    if (((unsigned)(w >> kQueueReservedShift) & kQueueMask) > h) { there = true; break; }
                    if ((unsigned)w >= (unsigned)PSX_JOB_INT(n_tickets)) break;
                    // a group that has not started may be waiting for THIS group's place on a CU: then nobody waits
                    if (started < gridDim.x || looks >= PSX_JOB_INT(retry_patience)) {
                        if (h < (unsigned)PSX_JOB_INT(retry_cap)) {
                            if (atomicCAS(&retry_slots(job)[h], (unsigned)in_loop((int)kRetryEmpty), (unsigned)in_loop((int)kRetryAbandoned)) != kRetryEmpty) there = true;      // filled this very moment
                            else atomicAdd(leave_word(job), 0x10000ull);          // a note for the group that re-arms the queue
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(64);
                }
                if (there) {
                    // The slot is filled right after it was reserved: the pusher sits between two adjacent atomics, and it is resident
                    // (it pushes from inside its frame loop).  Should it never come -- a faulted or preempted pusher -- the wait ends
                    // after about a second with the frame LOST: the slot is left to the pusher as "abandoned" (which then keeps its
                    // frame, see hand_on) or, if it was filled at the last moment, taken; a word of the lane counts the watchdog's
                    // bites for the host (psxhip_mdec_watchdog: non-zero = results of that launch are incomplete).
                    unsigned v = kRetryEmpty;
                    for (int looks = 0; v == kRetryEmpty && looks < (1 << 20); looks++) {
                        v = queue_peek(&retry_slots(job)[h], looks);
                        if (looks >= 4096) __builtin_amdgcn_s_sleep(32);
                    }
                    if (v == kRetryEmpty) {
                        v = atomicCAS(&retry_slots(job)[h], (unsigned)in_loop((int)kRetryEmpty), (unsigned)in_loop((int)kRetryAbandoned));
                        if (v == kRetryEmpty) { atomicAdd(&job.ticket[3], 1u); atomicAdd(leave_word(job), 0x10000ull); }
                    }
                    if (v != kRetryEmpty) {
                        L.scalars[S_FIDX] = (int)(v & 0xFFFFFFu);
                        got = (int)(v >> 24);
                        retry_slots(job)[h] = (unsigned)in_loop((int)kRetryEmpty);          // vacated for the next launch (nobody looks at it again in this one)
                    }
                }
                }
                L.scalars[S_RETRY] = got;
            }
            __syncthreads();
            if (L.scalars[S_RETRY] < 0) break;
            f = L.scalars[S_FIDX];
        }

        const int lane = tid & 63;
        const int blk = lane >> 3, r8 = lane & 7;
        LaneConst lc;
        lc.quant = L.qzz[lane];
        lc.lane = lane;
        lc.zsrc = (int)L.tab_nat[lane];
        // Which batch the frame belongs to (wave-uniform: f comes from one LDS word).  The batch table is read from the kernel
        // argument segment WHERE it is needed, through a pointer the compiler cannot see through -- here for the input, again at
        // the frame's end for the output: as plain members of `job` all 8 x 5 words were loaded at kernel entry and kept (spilled
        // into vector registers: two more than the 12-wavefront shape's 80 has room for).
        const int fu = __builtin_amdgcn_readfirstlane(f);
        const uint8_t* frame;
        const int32_t* b_sizes;
        int fl;                                            // index inside its batch
        int bi = 0;
        fl = fu;                                           // (the scalar copy: the frame's address below is then scalar arithmetic -- made from the vector copy it stayed in two vector registers, read back with two v_readfirstlane in every macroblock)
        b_sizes = job.batch[0].max_sizes;                  // one batch (the common launch): plain kernel arguments, nothing to look up
        frame = job.batch[0].frames + (size_t)fu * job.frame_stride;
        if (PSX_BATCHES_MANY()) {                           // (straight-line scalar code: a loop here cost the 12-wavefront shape two vector registers it does not have)
            BatchPtr bt = batch_table();
            FirstPtr ft = (FirstPtr)(bt + kMaxBatches);
#pragma unroll
            for (int i = 1; i < kMaxBatches; i++) bi += fu >= ft[i] ? 1 : 0;
            fl = fu - ft[bi];
            b_sizes = bt[bi].max_sizes;
            frame = bt[bi].frames + (size_t)fl * job.frame_stride;
        }
        int max_size = b_sizes ? b_sizes[fl] : job.uniform_max_size;
        // per-frame budgets live in device memory the host cannot vet: a budget outside [8, min(the context's maximum,
        // the output row)] is treated as "nothing fits" (result quant_scale 64, no bytes written)
        const bool bad_budget = max_size < 8 || max_size > job.max_frame_size || (size_t)max_size > job.out_stride;
        if (bad_budget) max_size = 8;
        // fits <=> 8 + 2*ceil(bits/16) <= max_size <=> bits <= 16 * floor((max_size - 8) / 2)   (mdec.c:321-333 in closed form)
        const int limit_bits = 16 * ((max_size - 8) >> 1);

        mark(0);   // ticket + idle

        // =====================================================================================
        // v3 / v3dc: the DC terms come first, because a block's DC code depends on the previous block of the same
        // component (mdec.c:454-479).  The DC coefficient of the islow DCT is exactly the sum of the block's 64
        // level-shifted samples (row pass 16 * row sum, column pass (16 * sum + 8) >> 4), so a pass of byte sums does it.
        // =====================================================================================
        if (CODEC != 0) {
            // Work item = 8 pixel rows x 128 bytes of one plane: lane = (16-byte chunk c = lane >> 3, row r = lane & 7), so a
            // wavefront reads whole cache lines.  A luma chunk spans the two blocks of one macroblock column, a chroma chunk
            // (8 interleaved Cr,Cb pairs) one Cr and one Cb block; the 8 row sums of a chunk sit in 8 adjacent lanes.
            // Two items per iteration keep two loads in flight per lane.
            {
                const int cpr = W >> 4;                       // 16-byte chunks per row (both planes)
                const int gpr = (cpr + 7) >> 3;               // chunk groups per row
                const int luma_items = (H >> 3) * gpr, total_items = luma_items + (H >> 4) * gpr;
                const int r = lane & 7, cl = lane >> 3;
                auto item_addr = [&](int item, bool& chroma, int& R, int& c, bool& ok) -> const uint8_t* {
#ifdef PSX_EXP_DC_REVERSE      // experiment: walk the frame bottom to top, luma and chroma interleaved, so that what the main pass reads first was read last
                    if (item < total_items) {
                        const int t = total_items - 1 - item, k = t / (3 * gpr), r = t - k * 3 * gpr;
                        item = r < 2 * gpr ? 2 * k * gpr + r : luma_items + k * gpr + (r - 2 * gpr);
                    }
#endif
                    chroma = item >= luma_items;
                    const int it2 = chroma ? item - luma_items : item;
                    R = it2 / gpr;
                    c = (it2 - R * gpr) * 8 + cl;
                    ok = item < total_items && c < cpr;
                    // (a lane without an item reads the frame's first bytes -- every lane loads, nothing sits under a branch -- and
                    //  its sums are not stored)
                    return frame + (ok ? (chroma ? (uint32_t)W * (uint32_t)H : 0u) + __umul24((uint32_t)(R * 8 + r), (uint32_t)W) + (uint32_t)c * 16u : 0u);
                };
                auto item_sums = [&](const uint4& v, bool chroma, int R, int c, bool ok) {
                    int s0, s1;
                    if (!chroma) {
                        s0 = (int)__builtin_amdgcn_sad_u8(v.x, 0u, __builtin_amdgcn_sad_u8(v.y, 0u, 0u));
                        s1 = (int)__builtin_amdgcn_sad_u8(v.z, 0u, __builtin_amdgcn_sad_u8(v.w, 0u, 0u));
                    } else {
                        const uint32_t m8 = 0x00FF00FFu;      // NV21: Cr (V) at even bytes
                        s0 = (int)__builtin_amdgcn_sad_u8(v.x & m8, 0u, __builtin_amdgcn_sad_u8(v.y & m8, 0u,
                                  __builtin_amdgcn_sad_u8(v.z & m8, 0u, __builtin_amdgcn_sad_u8(v.w & m8, 0u, 0u))));
                        s1 = (int)__builtin_amdgcn_sad_u8(v.x, 0u, __builtin_amdgcn_sad_u8(v.y, 0u,
                                  __builtin_amdgcn_sad_u8(v.z, 0u, __builtin_amdgcn_sad_u8(v.w, 0u, 0u)))) - s0;
                    }
                    s0 += wave::dpp_or_zero<0xB1, 0xF, 0xF>(s0);    // quad_perm [1,0,3,2]
                    s1 += wave::dpp_or_zero<0xB1, 0xF, 0xF>(s1);
                    s0 += wave::dpp_or_zero<0x4E, 0xF, 0xF>(s0);    // quad_perm [2,3,0,1]
                    s1 += wave::dpp_or_zero<0x4E, 0xF, 0xF>(s1);
                    s0 += wave::dpp_or_zero<0x141, 0xF, 0xF>(s0);   // row_half_mirror: all 8 rows of the chunk
                    s1 += wave::dpp_or_zero<0x141, 0xF, 0xF>(s1);
                    if (r == 0 && ok) {
                        const int mbe = chroma ? c * ny + R : c * ny + (R >> 1);
                        const int b0 = chroma ? 0 : 2 + (R & 1) * 2;
                        L.dcv[mbe * 6 + b0] = (int16_t)quant_dc(s0 - 64 * 128);
                        L.dcv[mbe * 6 + b0 + 1] = (int16_t)quant_dc(s1 - 64 * 128);
                    }
                };
                constexpr int kDcItems = 8;      // loads in flight per lane
                // (diagnostics build only: bit 24 of the priority word runs the byte-sum read TWICE -- same results; the difference
                //  between the two launches is what the read costs, i.e. what a v3 frame would gain if the DC terms came for free)
                const int dc_reps = STATS && ((job.prio_pattern >> 24) & 1u) ? 2 : 1;
                for (int rep = 0; rep < dc_reps; rep++)
                for (int item = wid; item < total_items; item += kDcItems * kWavesPerGroup) {
                    bool ch[kDcItems], ok[kDcItems];
                    int R[kDcItems], c[kDcItems];
                    uint4 v[kDcItems];
#pragma unroll
                    for (int u = 0; u < kDcItems; u++) {
                        const uint8_t* pp = item_addr(item + u * kWavesPerGroup, ch[u], R[u], c[u], ok[u]);
                        v[u] = *(const uint4*)pp;
                    }
#pragma unroll
                    for (int u = 0; u < kDcItems; u++) item_sums(v[u], ch[u], R[u], c[u], ok[u]);
                }
            }
            group_sync(1);
            // The DPCM chains (mdec.c:454-479): Cr, Cb and Y (4 blocks per macroblock), each in encode order.
            // Element i maps last -> new_last:
            //   dc % 4 != 2 : constant 4*round(dc/4)                (last is always a multiple of 4)
            //   dc % 4 == 2 : last < dc ? dc + 2 : dc - 2           (tie, rounds away from zero)
            // Both are step functions (thr, lo, hi); composition g(f(x)) = (thr_f, g(lo_f), g(hi_f)) is associative, so
            // a chain is an inclusive scan under composition.  Chunks of 64 elements are spread over all wavefronts:
            // (A) scan inside each chunk, keep the chunk's total; (B) scan the totals of each chain -> the value
            // entering every chunk; (C) scan again inside each chunk, apply, emit deltas.
            // A lane takes kPerLane CONSECUTIVE elements of a chain (composed serially in registers), the DPP scan then runs over
            // the 64 lane totals: a chunk is 64 * kPerLane elements and costs one DPP scan (~80 VALU) + ~12 per element, instead of
            // one DPP scan per 64 elements (the chain scan was ~22 of a 640x480 macroblock's ~250 VALU instructions, twice over).
            {
                constexpr int kPerLane = 8, kChunk = 64 * kPerLane;
                const int cc = (nmb + kChunk - 1) / kChunk, cy = (4 * nmb + kChunk - 1) / kChunk, n_chunks = 2 * cc + cy;
                auto chunk_of = [&](int q, int& chain, int& base, int& count) {
                    chain = q < cc ? 0 : (q < 2 * cc ? 1 : 2);
                    base = (q - (chain == 0 ? 0 : (chain == 1 ? cc : 2 * cc))) * kChunk;
                    count = chain == 2 ? 4 * nmb : nmb;
                };
                auto element = [&](int chain, int i, bool live, int& idx) -> StepFn {
                    idx = chain == 2 ? ((i >> 2) * 6 + 2 + (i & 3)) : (i * 6 + chain);
                    const int dc = live ? (int)L.dcv[idx] : 0;
                    StepFn f;
                    if ((dc & 3) == 2) {
                        f.thr = dc; f.lo = dc + 2; f.hi = dc - 2;
                    } else {
                        const int a = dc < 0 ? -dc : dc;
                        const int rq = ((a + 2) >> 2) << 2;
                        f.thr = 0; f.lo = f.hi = dc < 0 ? -rq : rq;
                    }
                    if (!live) { f.thr = -100000; f.lo = f.hi = 0; }   // dead elements sit after all live ones, never feed them
                    return f;
                };
                // this lane's elements of chunk (chain, base) composed in order
                auto lane_total = [&](int chain, int base, int count) -> StepFn {
                    const int i0 = base + lane * kPerLane;
                    int idx;
                    StepFn f = element(chain, i0, i0 < count, idx);
#pragma unroll
                    for (int j = 1; j < kPerLane; j++) f = compose(f, element(chain, i0 + j, i0 + j < count, idx));
                    return f;
                };
                // (A)
                for (int q = wid; q < n_chunks; q += kWavesPerGroup) {
                    int chain, base, count;
                    chunk_of(q, chain, base, count);
                    StepFn f = lane_total(chain, base, count);
                    scan_stepfn(f, lane);
                    if (lane == 63) { L.dc_fn[4 * q + 0] = f.thr; L.dc_fn[4 * q + 1] = f.lo; L.dc_fn[4 * q + 2] = f.hi; }
                }
                group_sync(1);
                // (B) wavefront c scans chain c's chunk totals (lanes = chunks)
                if (wid < 3) {
                    const int q0 = wid == 0 ? 0 : (wid == 1 ? cc : 2 * cc), nq = wid == 2 ? cy : cc;
                    int carry = 0;
                    for (int b0 = 0; b0 < nq; b0 += 64) {
                        const int j = b0 + lane;
                        StepFn f;
                        f.thr = -100000; f.lo = f.hi = 0;
                        if (j < nq) { f.thr = L.dc_fn[4 * (q0 + j) + 0]; f.lo = L.dc_fn[4 * (q0 + j) + 1]; f.hi = L.dc_fn[4 * (q0 + j) + 2]; }
                        scan_stepfn(f, lane);
                        const int cur = carry < f.thr ? f.lo : f.hi;          // value after chunk j
                        int before = __builtin_amdgcn_update_dpp(carry, cur, 0x138, 0xF, 0xF, false);   // wave_shr:1, lane 0 <- carry
                        if (j < nq) L.dc_fn[4 * (q0 + j) + 3] = before;       // value entering chunk j
                        carry = __builtin_amdgcn_readlane(cur, 63);
                    }
                }
                group_sync(1);
                // (C) the value entering each lane's run, then the run itself, element by element
                int bits = 0;
                for (int q = wid; q < n_chunks; q += kWavesPerGroup) {
                    int chain, base, count;
                    chunk_of(q, chain, base, count);
                    StepFn f = lane_total(chain, base, count);
                    scan_stepfn(f, lane);
                    const int cin = L.dc_fn[4 * q + 3];
                    const int after = cin < f.thr ? f.lo : f.hi;              // last value after this lane's run
                    int prev = __builtin_amdgcn_update_dpp(cin, after, 0x138, 0xF, 0xF, false);      // ... entering it
                    const int i0 = base + lane * kPerLane;
#pragma unroll
                    for (int j = 0; j < kPerLane; j++) {
                        int idx;
                        const bool live = i0 + j < count;
                        const StepFn e = element(chain, i0 + j, live, idx);
                        const int cur = prev < e.thr ? e.lo : e.hi;
                        int delta = (cur - prev) >> 2;                         // exact: both multiples of 4
                        if (CODEC == 2) {                                      // v3dc wrap (mdec.c:469-474)
                            if (delta < -0x80) delta += 0x100;
                            else if (delta > 0x80) delta -= 0x100;
                        }
                        int dlen;
                        uint32_t dcode;
                        dc_code<CODEC>(delta, chain == 2, L.dc_plen, L.dc_prefix, dlen, dcode);
                        if (live) {
                            L.dcv[idx] = (int16_t)delta;
                            bits += dlen;
                        }
                        prev = cur;
                    }
                }
                bits = wave::reduce_add(bits);
                if (lane == 0 && bits) atomicAdd(&L.scalars[S_DC_BITS], bits);
            }
            group_sync(1);
        }
        mark(1);   // reset + DC pre-pass
        const int dc_bits = CODEC == 0 ? 10 * nblk : L.scalars[S_DC_BITS];
        const int fixed_bits = dc_bits + 2 * nblk + 10;   // DC codes + end-of-block codes + end-of-frame code

        // =====================================================================================
        // One pass over the frame's macroblocks: DCT, then (optionally) AC bits at `count_scale`, then (optionally) the
        // macroblock's bitstream at `emit_scale` into the staging area.  `pilot` runs the DCT of the wavefront's first
        // macroblock only and leaves its coefficients in cf[]; the following pass picks them up (`resume`).
        // =====================================================================================
        float cf[6];            // this lane's coefficient of each block of the current macroblock, as float; lane 0 holds 0
        uint2 plo = make_uint2(0, 0), phi = make_uint2(0, 0);

        auto fetch_at = [&](uint32_t row, uint32_t col) {       // row = fy * 8 W, col = fx * 16
            // 32-bit offsets from the (wave-uniform) frame pointer; the macroblock row's offset is scalar arithmetic, a lane
            // only shifts it (a macroblock row is 8 W bytes of chroma, 16 W of luma)
            const uint4 tp = L.tab_pix[lane];
            const uint32_t o_lo = tp.x + (row << tp.z) + col, o_hi = o_lo + tp.y;
            plo = *(const uint2*)(frame + o_lo);
            phi = *(const uint2*)(frame + o_hi);
        };
        auto fetch = [&](int fx, int fy) { fetch_at((uint32_t)fy * (uint32_t)(8 * W), (uint32_t)fx * 16u); };
        // the lane's 8 pixels as bytes (b_lo: p0..p3, b_hi: p4..p7) from what fetch() left in (plo, phi)
        auto mb_bytes = [&](const uint4& ts, uint32_t& b_lo, uint32_t& b_hi) {
            b_lo = __builtin_amdgcn_perm(plo.y, plo.x, ts.x);
            b_hi = __builtin_amdgcn_perm(phi.y, phi.x, ts.y);
        };
        // ... and the fetch of the wavefront's NEXT macroblock behind them: always issued (a wavefront without a next macroblock
        // re-reads the frame's first one), and tied to the bytes above by an empty asm, so that the loads land in the very
        // registers the permutes have just read.  (Issued under `if (there is a next one)`, and hoisted above the permutes by
        // the scheduler, the loads needed a second set of four registers and 8 v_mov per macroblock to shuttle between the sets
        // -- 3 % of the loop's VALU instructions.)
        auto fetch_behind = [&](uint32_t row, uint32_t col, uint32_t& b_lo, uint32_t& b_hi) {      // row = fy * 8 W, col = fx * 16 (what the pass table holds)
            asm volatile("" : "+v"(b_lo), "+v"(b_hi));
            const uint4 tp = L.tab_pix[lane];
            uint32_t o_lo = tp.x + (row << tp.z) + col;
            asm volatile("" : "+v"(o_lo) : "v"(b_lo), "v"(b_hi));      // the address is "made from" the bytes: the loads stay behind the permutes
            const uint32_t o_hi = o_lo + tp.y;
            plo = *(const uint2*)(frame + o_lo);
            phi = *(const uint2*)(frame + o_hi);
        };
        // DCT of the macroblock whose pixel bytes are (b_lo, b_hi)
        auto dct_mb = [&](const uint4& ts, uint32_t b_lo, uint32_t b_hi) {
            int d[8];
            // the column pass's coefficient pairs: one scalar load from the kernel argument segment, issued here, waited for where
            // the column pass starts (the row pass in between hides it).  The compiler does not know the load is in flight: between
            // the two statements nothing may touch the registers -- tests/test_kernel_resources.py reads the disassembly for that.
            u32x16 ck;
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(ck) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(offsetof(FrameJob, col_k)));
            {
                // -- row pass on the matrix pipe: every lane hands in its own row vector (lane = (block, row); lanes 48..63 idle
                //    columns), and gets back four outputs of vector lane & 31 and four of vector 32 + (lane & 31)
                fdct8_rows_mfma(b_lo, b_hi, make_uint2(ts.z, ts.w), d);
                const int n = lane & 31;
                int16_t* t0 = tileT + (n >> 3) * kTileStride + (n & 7) + (lane >> 5) * 32;      // output c = 4 * (lane >> 5) + r at c * 8
                store_row_outputs(t0, d);
                if (n < 16) store_row_outputs(t0 + 4 * kTileStride, d + 4);                      // vectors 32..47: blocks 4, 5
            }
            wave_sync();
            if (lane < 48) {
                // -- column pass: lane = (block, column); 8 int16 of that column are contiguous
                // (the column's read and the wait for it AND for the coefficients in one statement: one s_waitcnt)
                uint4 q;
                asm volatile("ds_read2_b64 %0, %2 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(q), "+s"(ck) : "v"((uint32_t)(uintptr_t)&tileT[blk * kTileStride + r8 * 8]) : "memory");
                fdct8_col_acc_k(q.x, q.y, __builtin_amdgcn_alignbit(q.w, q.w, 16), __builtin_amdgcn_alignbit(q.z, q.z, 16), ck, d);
                // -- column 0 holds the block's DC term in d[0] (as an accumulator: value << 17).  v2: its quantised value
                //    (mdec.c:447-453) takes its place at scan position 0 and travels with the block's DC slot in the code list.
                //    v3: the DC codes come from the pre-pass (DPCM chain), position 0 holds 0.  Either way lane 0 is never treated
                //    as an AC coefficient.
                if (r8 == 0) d[0] = CODEC == 0 ? (int)((uint32_t)quant_dc(d[0] >> 17) << 17) : 0;
                // the column's 8 outputs (rows 0..7) leave as one 16-byte store; scan order is applied by the readers (LaneConst::zsrc)
                uint4 o;
                o.x = pack_sh17(d[1], d[0]);
                o.y = pack_sh17(d[3], d[2]);
                o.z = pack_sh17(d[5], d[4]);
                o.w = pack_sh17(d[7], d[6]);
                *(uint4*)&tileZ[lane * 8] = o;
            }
            wave_sync();     // tileZ holds the macroblock's coefficients: block b, column c, row r at [b * 64 + c * 8 + r]
        };

        // ---- pilot: kPilotPerWave macroblocks per wavefront, spread evenly over the frame in raster order; their AC bits
        //      at a few scales, scaled up to the frame, predict the answer
        // A group that has just encoded a frame with the same budget skips the pilot and starts from that frame's answer
        // (consecutive tickets are neighbouring frames); the quarter-pass checkpoint catches the cases where it is off.
        // A group's first frame borrows the answer of the previous launch's last frame (the tail of the previous batch --
        // temporally adjacent when batches follow each other in a stream).
        // Trust policy.  A hint is LOCAL when it is the answer of this frame's neighbour in time (the frame before it in the group's
        // run): trusted, the quarter-pass checkpoint catches the scene cuts.  Every other hint is FOREIGN -- the last frame of the
        // group's previous run, a whole round of the grid away, or the previous launch's last answer: right on content that does not
        // change (the uniform synthetic batches), wrong three times in four on scene-structured video (tools/gpu_r05_diag.py: 2.3
        // passes per frame).  Whether foreign hints are trusted is decided per LAUNCH: every group counts how many of its foreign
        // hints were (or would have been) the answer and hands the counts in when it leaves; more than one in four wrong, and the
        // next launch's groups run the pilot on every frame that has no neighbour's answer (the foreign hint then only tells the
        // pilot where to look first) -- and keep counting, so that the verdict turns back when the content does.  Inside a launch a
        // group switches on its own only after TWO foreign hints in a row have failed (a change of scene).  One miss says nothing:
        // on content whose answer flips between two scales (640x480 at 8 KiB, noise +-4: one frame in eight lands on 5 instead of
        // 6) a pilot after every miss did worse than the hint it replaced -- a dozen macroblocks of 1200 are right 60 % of the time
        // there, the hint 87 % -- 1.8 M frames/s became 1.2 M.
        // The decision is thread 0's alone (seven scalars read, a dozen scalar instructions: made by every wavefront it was 1 % of the
        // uniform headline's instructions): S_PILOT_GUESS = the hint the frame starts from, or 0 -- the pilot runs, and S_PILOT_HI says
        // where it looks first.
        if (tid == 0) {
            int hint = L.scalars[S_HINT];
            int hint_budget = L.scalars[S_HINT_BUDGET];
            bool local = hint >= 1 && L.scalars[S_HINT_FRAME] == f - 1;
            if (hint < 1) {
                const int sh = L.scalars[S_SHARED_HINT];
                hint = sh & 0xFF;
                hint_budget = sh >> 8;
            }
            const int retry_scale = L.scalars[S_RETRY];
            if (retry_scale > 0) {             // handed on by another group together with the scale its search wanted next
                hint = retry_scale;
                hint_budget = max_size;
                local = true;
            }
            const bool hint_ok = hint >= 1 && hint <= 63 && hint_budget == max_size;
            const int dts = L.scalars[S_DISTRUST];
            const int rp = L.scalars[S_REPILOT];     // this frame is being started over from the pilot (the end of the frame loop): the verdict that sent it back
            const int trust_mode = PSX_JOB_INT(trust_mode);
            const bool distrust = trust_mode == 0 ? ((dts & 1) != 0 || (dts >> 8) >= 2) : trust_mode == 2;
            const bool trust_hint = hint_ok && (local || !distrust) && rp == 0;
            L.scalars[S_ABORTS_LEFT] = rp ? 1 : 2;          // (a frame sent back to the pilot has used one)
            L.scalars[S_FOREIGN] = hint_ok && !local ? hint : 0;
            L.scalars[S_PILOT_GUESS] = trust_hint ? hint : 0;
            int h0 = hint_budget == max_size ? hint : 0;        // (not trusted, or for another budget: it still says where to look first)
            if (rp) {
                h0 = rp;
                L.scalars[S_REPILOT] = in_loop(0);
            }
            if (h0 < 1) {
                // no frame of its own yet: another group's last answer for the same budget is a good place to start looking
                const int sh = L.scalars[S_SHARED_HINT];
                if ((sh >> 8) == max_size) h0 = sh & 0xFF;
            }
            L.scalars[S_PILOT_HI] = h0;
        }
        // A frame started from a hint (no pilot) whose first pass is stopped with a verdict FAR from the hint -- a scene cut: the
        // neighbour's answer said 23 and the checkpoint's projection says 3 -- starts over from the pilot: the verdict is a
        // two-point extrapolation from the wrong end of the curve and used to cost such frames three to five passes; the pilot
        // brackets the answer on a sample for a tenth of a pass.  Once per frame at most.
        auto search_reset = [&](MdecSearch& st) {
            mdec_search_init(st);
            st.best = in_loop(st.best);
            {       // (the zeros too: as loop-invariant constants they took two registers for the whole kernel -- and a scratch slot)
                const int z = in_loop(0);
                st.lo = z; st.staged = z; st.pad = z;
                st.fail = (uint64_t)(uint32_t)z | ((uint64_t)(uint32_t)in_loop(0) << 32);
                st.fs[0] = st.fs[1] = st.fb[0] = st.fb[1] = z;
                st.gs[0] = st.gs[1] = st.gb[0] = st.gb[1] = z;
            }
        };
        unsigned long long trace = 0;       // diagnostics: the first four passes (of the last attempt): emit scale, or count scale | 0x40; | 0x80 stopped at the checkpoint
        int n_pass = 0, first_abort = 0, guess0 = 0;      // (first_abort, guess0: diagnostics)
        int guess;
        MdecSearch* srch = (MdecSearch*)&L.scalars[S_SEARCH];
        group_sync(1);
        if (L.scalars[S_PILOT_GUESS] == 0) {
        float cfp[kPilotPerWave][6];
#pragma unroll
        for (int i = 0; i < kPilotPerWave; i++) {
            const int pi = wid + i * kWavesPerGroup;
            if (pi < n_pilot) {
                // The sample: rows spread evenly, columns by the golden ratio (a Kronecker lattice).  Evenly spaced RASTER indices looked
                // even and were not: at 640x480 (40 x 30 macroblocks, 12 samples) the stride of 100 lands in two columns only, and a
                // picture with one busy column -- the synthetic frames' wrap-around edge, a real one's vertical bar -- was estimated
                // from six macroblocks of it or from none (pilot guesses 2 .. 61 for an answer of 6).
                const unsigned u = (unsigned)(2 * pi + 1);
                const int p_fy = (int)((u * (unsigned)ny) / (unsigned)(2 * n_pilot));
                const int p_fx = (int)((((u * 40503u) >> 1) & 0xFFFFu) * (unsigned)nx >> 16);          // frac((pi + 1/2) * 0.618) * nx
                fetch(n_pilot == nmb ? pi % nx : p_fx, n_pilot == nmb ? pi / nx : p_fy);
                {
                    const uint4 ts = L.tab_sel[lane];
                    uint32_t b_lo, b_hi;
                    mb_bytes(ts, b_lo, b_hi);
                    dct_mb(ts, b_lo, b_hi);
                }
#pragma unroll
                for (int b = 0; b < 6; b++) cf[b] = lane == 0 ? 0.0f : (float)(int)tileZ[b * kZStride + lc.zsrc];
                wave_sync();
            }
#pragma unroll
            for (int b = 0; b < 6; b++) cfp[i][b] = pi < n_pilot ? cf[b] : 0.0f;
        }
        if (tid == 0) {
            L.scalars[S_PILOTED] = 1;
            const int h0 = L.scalars[S_PILOT_HI];
            // The pilot is steered by the search's own two-point model (mdec_pilot_next, mdec_search.h; its state is a MdecSearch of
            // ESTIMATES kept where the exact search's state will live -- that one is set up after the pilot): a hint is checked first
            // (h0 - 1, h0), then the model's prediction and its neighbours -- two rounds and five evaluations on average, where
            // bracketing by halves took three to five rounds of four (17 % of a group's time on scene-structured content).
            MdecSearch st;
            search_reset(st);
            *srch = st;
            const MdecPilot pl = mdec_pilot_next(st, h0, limit_bits, fixed_bits, 0);
            L.scalars[S_PILOT_N] = pl.n;
            L.scalars[S_PILOT_SCALE0 + 0] = pl.s[0];
            L.scalars[S_PILOT_SCALE0 + 1] = pl.s[1];
            L.scalars[S_PILOT_SCALE0 + 2] = pl.s[2];
            L.scalars[S_PILOT_LO] = in_loop(0);       // rounds evaluated so far
        }
        for (;;) {
            group_sync(1);
            const int np = L.scalars[S_PILOT_N];
            if (np == 0) break;
            if (wid < n_pilot) {
                for (int j = 0; j < np; j++) {
                    const int s = L.scalars[S_PILOT_SCALE0 + j];
                    const QuantK k = make_quant(lc.quant, s);
                    int acc = 0;
#pragma unroll
                    for (int i = 0; i < kPilotPerWave; i++)
                        if (i < pilot_rounds) acc += count_mb(cfp[i], k, lc, L.ac_len16) & 0xFF;
                    const int t = wave::reduce_add(acc);
                    if (lane == 0) atomicAdd(&L.scalars[S_PILOT_BITS0 + j], t);
                }
            }
            group_sync(1);
            if (tid == 0) {
                MdecSearch st = *srch;
                for (int j = 0; j < np; j++) {
                    const int s = L.scalars[S_PILOT_SCALE0 + j];
                    const int est = (int)((long long)L.scalars[S_PILOT_BITS0 + j] * nmb / n_pilot) + fixed_bits;
                    mdec_search_note(st, s, est, est, limit_bits);
                    L.scalars[S_PILOT_BITS0 + j] = 0;
                }
                *srch = st;
                const int round = L.scalars[S_PILOT_LO] + 1;
                const MdecPilot pl = mdec_pilot_next(st, L.scalars[S_PILOT_HI], limit_bits, fixed_bits, round);
                L.scalars[S_PILOT_N] = pl.n;
                L.scalars[S_PILOT_LO] = round;
                L.scalars[S_PILOT_SCALE0 + 0] = pl.s[0];
                L.scalars[S_PILOT_SCALE0 + 1] = pl.s[1];
                L.scalars[S_PILOT_SCALE0 + 2] = pl.s[2];
                if (pl.n == 0) L.scalars[S_PILOT_GUESS] = pl.guess;
            }
        }
        }
        guess = L.scalars[S_PILOT_GUESS];
        if (STATS) { guess0 = carry_pass ? carry_guess0 : guess; if (carry_pass) first_abort = carry_abort; }
        mark(2);   // pilot

        // ---- exact search (mdec_search.h): the state lives in LDS, thread 0 advances it between passes; every pass is
        //      described by two scalars
        if (tid == 0) {
            L.scalars[S_NEXT_DRAW] = (int)next_draw;
            MdecSearch st;
            search_reset(st);
            MdecPass np;
            if (limit_bits < fixed_bits || bad_budget) {
                np.done = 1; np.count_scale = 0; np.emit_scale = 0;
            } else {
                np = mdec_search_next(st, guess, limit_bits, fixed_bits);
            }
            *srch = st;
            L.scalars[S_PASS_COUNT] = np.count_scale;
            L.scalars[S_PASS_EMIT] = np.emit_scale;
            L.scalars[S_DONE] = np.done;
            L.scalars[S_RESULT] = st.best;
            L.scalars[S_MB_NEXT] = 2 * kWavesPerGroup;
        }
        group_sync(1);

        while (!L.scalars[S_DONE]) {
            const int count_scale = L.scalars[S_PASS_COUNT], emit_scale = L.scalars[S_PASS_EMIT];
            n_pass++;
            if (STATS && n_pass <= 4) trace |= (unsigned long long)(emit_scale ? emit_scale : count_scale | 0x40) << (24 + 8 * n_pass);
            if (emit_scale && n_pass > 1) {
                // a further emitting pass rebuilds the staging area
                for (int i = tid; i < job.stg_words; i += kThreads) L.stg[i] = 0u;
                if (tid == 0) L.scalars[S_STG_NEXT] = 0;
                group_sync(5);
            }
            // Macroblocks are handed out by ticket (an LDS counter): wavefronts that draw cheap macroblocks draw more, and all of
            // them reach the end of the pass within one macroblock of each other.  Ticket t visits macroblock order[t]; the order
            // (psxhip_mdec_pass_order) spreads every run of tickets evenly over the frame.  A wavefront's first two tickets are
            // its own; from then on it draws the ticket after next while it works, so that neither the counter's round trip nor
            // the order look-up (a scalar load) nor the pixel fetch of the next macroblock is waited for.
            // A table entry is what the loop needs of its macroblock, ready made (psxhip_mdec_pass_table): x = fy * 8 W (the
            // macroblock row's byte offset in the chroma plane; luma lanes shift it) | valid << 31, y = fx * 16 | 4 * encode-order
            // index << 16.  An entry without a macroblock is all zero: its fetch reads macroblock (0, 0), no select needed.
            // The table ends with one all-zero entry: a ticket past the end (the last draws of a pass; the checkpoint's "stop")
            // is clamped onto it -- one scalar min and a load at a 32-bit byte offset, no compare, branch or 64-bit address.
            typedef const char __attribute__((address_space(4))) * OrderBytes;
            typedef const uint32_t __attribute__((address_space(4))) * OrderPtr;
            const OrderBytes order_b = (OrderBytes)(uintptr_t)job.order;
            const int n_tickets = job.trips * kWavesPerGroup;
            auto order_at = [&](int t) -> uint2 {
                const uint32_t byte = min((uint32_t)t, (uint32_t)n_tickets) * 8u;
                const OrderPtr e = (OrderPtr)(order_b + byte);
                return make_uint2(e[0], e[1]);
            };
            int cur_t = wid, nxt_t = wid + kWavesPerGroup;
            uint2 cur_o = order_at(cur_t);
            uint2 nxt_o = order_at(nxt_t);
            fetch_at(cur_o.x & 0x7FFFFFFFu, cur_o.y & 0xFFFFu);
            const QuantK kc = make_quant(lc.quant, count_scale ? count_scale : 1);
            const QuantK ke = make_quant(lc.quant, emit_scale ? emit_scale : 1);
            int acc_cnt = 0;             // per lane: bits | deficit << 16 over this wavefront's macroblocks (count scale)
            int acc_edef = 0;            // per lane: deficit over the emitted codes
            int n_codes = 0;             // wave-uniform: codes emitted (AC codes + DC slots)
            int emit_bits = 0, mb_done = 0;  // wave-uniform
            int prev_len = 0;                // wave-uniform: length of the previous macroblock's list at the count scale (above 128: too long to walk)
            // compaction threshold of this lane: smallest |n| that quantises to non-zero at the list's scale
            // (|n| >= t  <=>  (unsigned)(n + t - 1) >= 2 t - 1: one add and one compare, no absolute value)
            const uint32_t thr_low = (uint32_t)((lc.quant * (count_scale ? count_scale : 1) + 1) >> 1);
            const uint32_t thr_emit = (uint32_t)((lc.quant * (emit_scale ? emit_scale : 1) + 1) >> 1);
            // (lane 0 -- scan position 0, the block's DC slot -- is always kept: its span is 0, and x + off >= 0 always holds)
            // (the list's scale: the count scale when the pass counts, else the emit scale -- chosen here, once per pass)
            const uint32_t thr_list = count_scale ? thr_low : thr_emit;
            const uint32_t low_off = thr_list - 1u, low_span = lane == 0 ? 0u : 2u * thr_list - 1u;
            const uint32_t emit_off = thr_emit - 1u, emit_span = lane == 0 ? 0u : 2u * thr_emit - 1u;
            // per-wavefront totals -> LDS (also used by the checkpoint: flushing resets the partial sums)
            auto flush = [&]() {
                if (count_scale) {
                    const int tf = wave::reduce_add(acc_cnt & 0xFFFF), td = wave::reduce_add((int)((unsigned)acc_cnt >> 16));
                    if (lane == 0) { atomicAdd(&L.scalars[S_CNT_F], tf); atomicAdd(&L.scalars[S_CNT_D], td); }
                    acc_cnt = 0;
                }
                if (emit_scale) {
                    const int td = wave::reduce_add(acc_edef), tc = n_codes;
                    if (lane == 0) {
                        atomicAdd(&L.scalars[S_EMIT_BITS], emit_bits);
                        atomicAdd(&L.scalars[S_EMIT_D], td);
                        atomicAdd(&L.scalars[S_NNZ], tc - 6 * mb_done);      // AC codes = all codes - the DC slots
                    }
                    acc_edef = 0; n_codes = 0; emit_bits = 0;
                }
                if (lane == 0) atomicAdd(&L.scalars[S_CK_DONE], mb_done);
                mb_done = 0;
            };
            // Checkpoint after a quarter of the pass: the macroblocks done so far are an even sample of the frame (psxhip_mdec_pass_order);
            // if their bits, scaled up, say that this pass's scales cannot be the answer, stop and start over with a better guess
            // instead of finding out at the end.  Projections only steer: nothing they say enters the search state.
            const int check_t = (job.trips >= 8 && L.scalars[S_ABORTS_LEFT] > 0) ? (job.trips >> 2) * kWavesPerGroup : n_tickets;
            bool aborted = false, checked = false;

            for (int it = 0; cur_t < n_tickets; it++) {
                // (every wavefront gets here exactly once, with its first ticket past the quarter mark: the sums then cover
                //  exactly the first check_t tickets)
                if (!checked && cur_t >= check_t) {
                    // No barrier: every wavefront adds its sums when it gets here and carries on; the one that arrives last
                    // (LDS atomics of a wavefront complete in order, so by then all sums are in) judges the projection.  A
                    // verdict "stop" empties the ticket counter: every wavefront finishes the (at most two) macroblocks it
                    // holds tickets for and falls out of the loop by itself -- 8 % of a pass wasted when a pass is stopped,
                    // two group barriers saved in every pass.
                    checked = true;
                    flush();
                    int arrived = 0;
                    if (lane == 0) arrived = atomicAdd(&L.scalars[S_CK_WAVES], 1);
                    const bool judge = __builtin_amdgcn_readfirstlane(arrived) == kWavesPerGroup - 1;
                    // the spread of the sample: sums of x and x^2, x = a macroblock's stream bits >> 2, over the tickets before the
                    // mark -- read back from the records the macroblocks left (stage_alloc), by the whole judging wavefront, once
                    // per pass (kept as running sums they were six scalar instructions in every macroblock of every wavefront)
                    unsigned ck_s1 = 0, ck_sq = 0;
                    // (a pass that has run out of staging room -- S_STG_NEXT past the area: offsets above 16 bits have spilled into the
                    //  records' bit counts -- keeps the default margin: its stream is not going to be used anyway)
                    // (an atomic load, inside the judge's branch: as a plain load the compiler hoisted it to the top of the macroblock
                    //  loop -- a read and a wait for ALL outstanding LDS traffic, the ticket draw included, in every macroblock of
                    //  every wavefront: 1.5 % of the kernel)
                    bool records_ok = false;
                    if (judge && emit_scale) records_ok = __hip_atomic_load(&L.scalars[S_STG_NEXT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= job.stg_words;
                    if (records_ok) {
                        for (int t = lane; t < check_t; t += 64) {
                            const OrderPtr oe = (OrderPtr)(order_b + (uint32_t)t * 8u);
                            if ((int)oe[0] < 0) {
                                const unsigned x = *(const uint32_t*)((const char*)L.rec + (oe[1] >> 16)) >> 18;
                                ck_s1 += x;
                                ck_sq += x * x;
                            }
                        }
                        ck_s1 = (unsigned)wave::reduce_add((int)ck_s1);
                        ck_sq = (unsigned)wave::reduce_add((int)ck_sq);
                    }
                    if (judge && lane == 0) {
                        const int done = L.scalars[S_CK_DONE];
                        long long pa = 0, pb = 0;
                        if (count_scale) pa = (long long)L.scalars[S_CNT_F] * nmb / done + fixed_bits;
                        if (emit_scale) pb = (long long)L.scalars[S_EMIT_BITS] * nmb / done + 10;
                        // How far on the wrong side is "clearly"?  The projection is a sample mean scaled up; its standard error
                        // follows from the spread of the sample's macroblocks (finite-population form).  Stopping costs a quarter
                        // pass and is right when the verdict is; carrying on costs a whole pass when it is wrong: stop when the
                        // projection is wrong-sided by more than ck_margin / 1000 standard errors (1.0; 0.8 until mdec-k3.7's closing session -- swept over nine workloads
                        // with tools/gpu_ckmargin_sweep.py: content whose answer flips between neighbouring scales gains 14 %
                        // over a fixed 5 % margin, stable content is unaffected).
                        int margin = (limit_bits - fixed_bits) / 50;
                        if (emit_scale && records_ok && done > 1 && done < nmb) {
                            const float n = (float)done, mean = (float)ck_s1 / n;      // of x = bits >> 2
                            float var = (float)ck_sq / n - mean * mean;
                            var = var > 0.0f ? var : 0.0f;
                            const float se = 4.0f * __builtin_sqrtf(var * n * (1.0f - n / (float)nmb)) * ((float)nmb / n);
                            margin = (int)(se * (float)PSX_JOB_INT(ck_margin) * 0.001f);
                        }
                        int g = mdec_search_checkpoint_bits(*srch, count_scale, (int)pa, emit_scale, (int)pb, limit_bits, fixed_bits, margin);
                        // A verdict FAR from where a PILOTED frame started is not taken: it is a two-point extrapolation from a quarter of
                        // the pass, the pilot was a measurement of its own, and on pictures that are not the same everywhere (a quarter of
                        // the frame full-contrast bars, the rest flat: answer 5, pilot 6, verdict 39) the quarter's rows over-weigh one
                        // part -- such frames took five passes.  The pass runs to its end and the search goes on from what it counted.
                        // ... unless the projection is on the wrong side by FOUR times the margin: then it is the pilot that was wrong.
                        {
                            const int cur = emit_scale ? emit_scale : count_scale + 1, far = cur > 8 ? cur >> 2 : 2;
                            if (g && L.scalars[S_PILOTED] && (g - cur >= far || cur - g >= far)) {
                                // (the test of mdec_search_checkpoint_bits with four times the margin; the new guess stays the one just made)
                                const bool too_low = emit_scale && pb > (long long)limit_bits + 4 * margin;
                                const bool too_high = count_scale && pa <= (long long)limit_bits - 4 * margin;
                                if (!too_low && !too_high) g = 0;
                            }
                        }
                        if (g) {
                            L.scalars[S_ABORT] = g | (n_pass << 8);
                            L.scalars[S_ABORTS_LEFT] = L.scalars[S_ABORTS_LEFT] - 1;
                            atomicMax(&L.scalars[S_MB_NEXT], 0x40000000);
                        }
                    }
                }
                if (WAVES == kWavesSmall) {
                    // (s_setprio takes an immediate, so this IS a branch; written out, because the compiler's own if / else around
                    //  the two builtins came to eight scalar instructions and two branches per macroblock)
                    //  (s_bitcmp1 takes the bit number from a register's low five bits)
                    asm volatile("s_bitcmp1_b32 %0, %1\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n\ts_branch 2f\n1:\ts_setprio 1\n2:" : : "s"(prio_bits4), "s"(it) : "scc");
                }
                int drawn = 0;
                if (lane == 0) drawn = atomicAdd(&L.scalars[S_MB_NEXT], 1);
                const bool valid = (int)cur_o.x < 0;
                const uint32_t rec_byte = cur_o.y >> 16;        // 4 * the macroblock's encode-order index: its record's byte offset
                const int mbe = (int)(rec_byte >> 2);
                {
                    const uint4 ts = L.tab_sel[lane];
                    uint32_t b_lo, b_hi;
                    mb_bytes(ts, b_lo, b_hi);
                    fetch_behind(nxt_o.x & 0x7FFFFFFFu, nxt_o.y & 0xFFFFu, b_lo, b_hi);
                    if (kStopAfter == 1) { asm volatile("" :: "v"(b_lo), "v"(b_hi)); }
                    else if (valid) dct_mb(ts, b_lo, b_hi);
                }
                cur_t = nxt_t;
                cur_o = nxt_o;
                nxt_t = __builtin_amdgcn_readfirstlane(drawn);
                nxt_o = order_at(nxt_t);
                if (!valid) continue;
                mb_done++;
                if (kStopAfter == 1 || kStopAfter == 2) continue;

                int ci[6];       // this lane's coefficient (scan position = lane) of each block; lane 0 (the DC slot) holds 0
#pragma unroll
                for (int b = 0; b < 6; b++) ci[b] = (int)tileZ[b * kZStride + lc.zsrc];
                wave_sync();     // the tiles are free again (the code list aliases them)

                if (!emit_scale) {
                    // count-only pass (rare: closing a gap below a scale that is already staged)
                    float cff[6];
#pragma unroll
                    for (int b = 0; b < 6; b++) cff[b] = lane == 0 ? 0.0f : (float)ci[b];
                    const int a = count_mb(cff, kc, lc, L.ac_len16);
                    acc_cnt += (a & 0xFF) | ((a >> 8) << 16);
                    continue;
                }
                // An emitting pass, with (cs) or without counting at the scale below: the macroblock's work is instantiated for
                // each -- "does this pass count" is then no test at all, and what remains of the dense-macroblock logic is two
                // compares of scalar integers (run-time flags here were some twenty scalar instructions and branches per macroblock).
                constexpr std::true_type yes{};
                constexpr std::false_type no{};
                auto emit_mb = [&](auto cs_tag) {
                    constexpr bool cs = decltype(cs_tag)::value;
                    // ---- 1. compaction.  At the scales that matter only a few of a block's 64 coefficients are non-zero, so
                    //      everything expensive (VLC look-ups, bit positions, LDS writes) runs on a COMPACTED list, built once
                    //      per macroblock at the pass's LOWER scale (a coefficient that is non-zero at a coarser scale is non-zero
                    //      at every finer one): |n| quantises to non-zero  <=>  2|n| >= d  <=>  |n| >= ceil(d / 2).
                    //      Entry: [16:0] n (17-bit two's complement), [22:17] scan position.  Lane 0 (scan position 0) is always kept: it
                    //      marks the block's DC slot, so the list is the macroblock's code sequence DC, AC..., DC, AC...
                    //      A macroblock that is dense at the count scale (a list of more than two chunks) is listed again at the
                    //      emit scale and counted in place (count_mb): walking a long list costs more than it saves.
                    auto build_list = [&](uint32_t off, uint32_t span) -> int {
                        // Written out: per block add, compare, (independent) entry, the mask into exec, rank (mbcnt), address, store,
                        // exec back, population count, running byte address -- 6 vector + 4 scalar instructions and no branch, where
                        // the compiler's version of the same six lines was 6 + 6 and a branch around every store (scalar
                        // instructions weigh as much as vector ones here).  gfx950 wants two wait states between
                        // a VALU write of vcc and a VALU read of it as an operand (v_mbcnt): the entry and the saveexec stand there.
                        // Lane 0 keeps by construction (see low_span); the tag is made here from the lane number (no register
                        // held across the loop).
                        const uint32_t cb0 = (uint32_t)(uintptr_t)clist;                 // LDS byte address (wave-uniform)
                        uint32_t cb = cb0, t, e, tag, n;
                        unsigned long long sv;
#define PSX_LIST_BLOCK(CI)                                                \
                            "v_add_u32 %[t], %[off], %[" CI "]\n\t"                \
                            "v_cmp_ge_u32 vcc, %[t], %[span]\n\t"                  \
                            "v_and_or_b32 %[e], %[" CI "], %[m17], %[tag]\n\t"      \
                            "s_and_saveexec_b64 %[sv], vcc\n\t"                    \
                            "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"               \
                            "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"            \
                            "v_lshl_add_u32 %[t], %[t], 2, %[cb]\n\t"              \
                            "ds_write_b32 %[t], %[e]\n\t"                          \
                            "s_mov_b64 exec, %[sv]\n\t"                            \
                            "s_bcnt1_i32_b64 %[n], vcc\n\t"                        \
                            "s_lshl2_add_u32 %[cb], %[n], %[cb]\n\t"
                        asm volatile("v_lshlrev_b32 %[tag], 17, %[lane]\n\t"
                                     PSX_LIST_BLOCK("c0") PSX_LIST_BLOCK("c1") PSX_LIST_BLOCK("c2")
                                     PSX_LIST_BLOCK("c3") PSX_LIST_BLOCK("c4") PSX_LIST_BLOCK("c5")
                                     : [cb] "+s"(cb), [t] "=&v"(t), [e] "=&v"(e), [tag] "=&v"(tag), [n] "=&s"(n), [sv] "=&s"(sv)
                                     : [off] "v"(off), [span] "v"(span), [m17] "s"(0x1FFFFu), [lane] "v"(lane),
                                       [c0] "v"(ci[0]), [c1] "v"(ci[1]), [c2] "v"(ci[2]), [c3] "v"(ci[3]), [c4] "v"(ci[4]), [c5] "v"(ci[5])
                                     : "vcc", "scc", "memory");
#undef PSX_LIST_BLOCK
                        // (the count from a statement whose ONLY output is scalar by constraint: out of the statement above, or as
                        //  plain arithmetic on the LDS address, the compiler took it for lane-dependent -- and every branch on it)
                        asm("s_sub_u32 %0, %1, %2\n\ts_lshr_b32 %0, %0, 2" : "=s"(n) : "s"(cb), "s"(cb0) : "scc");
                        return (int)n;
                    };
                    // (built straight away; only after a dense macroblock the next one is sized up first -- busy content comes in runs)
                    bool dense = false;
                    int count = 0;
                    if (cs && prev_len > 128) {      // (prev_len: the previous macroblock's list length at the count scale, the six DC slots included)
                        int n_low = 0;
#pragma unroll
                        for (int b = 0; b < 6; b++) n_low += (int)__builtin_popcountll(wave::ballot((uint32_t)ci[b] + low_off >= low_span));
                        prev_len = n_low;
                        dense = n_low > 128;
                    }
                    if (!dense) {
                        count = build_list(low_off, low_span);
                        if (cs) {
                            prev_len = count;
                            dense = count > 128;
                        }
                    }
                    if (kStopAfter == 3) { asm volatile("" :: "s"(count)); wave_sync(); return; }
                    if (dense) {
                        float cff[6];
#pragma unroll
                        for (int b = 0; b < 6; b++) cff[b] = lane == 0 ? 0.0f : (float)ci[b];
                        const int a = count_mb(cff, kc, lc, L.ac_len16);
                        acc_cnt += (a & 0xFF) | ((a >> 8) << 16);
                        wave_sync();
                        count = build_list(emit_off, emit_span);
                    }
                    wave_sync();

                    // ---- 2. one chunk of <= 64 list entries.
                    //      Count scale (= the list's scale): every entry is a code; the run before an AC coefficient is
                    //      (its position - its predecessor's - 1), one DPP shift.
                    //      Emit scale (coarser): entries whose level drops to 0 fall out; the run is taken from the previous
                    //      SURVIVING entry (ballot, count-leading-zeros below this lane, ds_bpermute).  DC slots always survive,
                    //      so a run never crosses a block.  Each block's 2-bit end-of-block code "10" (mdec.c:501-503) travels
                    //      as two extra leading bits of the NEXT block's DC code; the macroblock's last one is appended by
                    //      stage_alloc().
                    int kcarry_a = 0, kcarry_b = 0, bcarry = 0;
                    // (low / do_count / single are compile-time: as run-time flags they cost a dozen scalar instructions and several
                    //  branches per chunk -- and scalar instructions weigh as much as vector ones here)
                    auto chunk = [&](auto low_tag, auto count_tag, auto single_tag, int base, int& len, uint32_t& code, int& deficit, int& cnt16, int& ncodes) {
                        constexpr bool low = decltype(low_tag)::value, do_count = decltype(count_tag)::value, single = decltype(single_tag)::value;
                        const int i = base + lane;
                        const bool live = i < count;
                        uint32_t e = clist[i];                                  // (always inside the wavefront's tiles: i < 384)
                        e = live ? e : (63u << 17);                             // dead lanes: |n| = 0 at scan position 63
                        const int k = (int)__builtin_amdgcn_ubfe(e, 17, 6);
                        const uint32_t negbit = __builtin_amdgcn_ubfe(e, 16, 1);      // the sign as a number: only the escape code needs it as a predicate
                        const bool is_dc = k == 0;
                        const uint64_t dcm = CODEC == 0 ? 0ull : wave::ballot(k == 0);        // (v3: the DC slots' block numbers)
                        const bool is_ac = live && !is_dc;
                        // the coefficient (an int16: the tiles hold 16-bit values, bit 16 repeats the sign) as a float, one instruction:
                        // the convert sign-extends the entry's low half itself (SDWA); the quantiser takes |.|
                        float magf;
                        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(magf) : "v"(e));
                        cnt16 = 0;
                        int kprev;
                        bool is_last;
                        // the quantiser constants belong to the entry's scan position k, i.e. they sit in lane k's registers
                        QuantK ek;
                        ek.inv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(k << 2, __builtin_bit_cast(int, ke.inv)));
                        ek.bias = __builtin_fmaf(0.25f, ek.inv, 0.5f);
                        // (the survivors' slots, below: the base is wave-uniform and lives in a spilled scalar register.  Fetched HERE, in
                        //  front of the table look-ups: fetched where it is used, behind them, the compiler has put a wait for ALL
                        //  outstanding LDS traffic next to the reload -- one exposed LDS round trip per block, 1.5 % of the kernel)
                        uint32_t kbase = (uint32_t)(uintptr_t)(clist + 64);
                        if (low) asm volatile("" : "+s"(kbase));
                        if (low) {
                            const int ka = __builtin_amdgcn_update_dpp(kcarry_a, k, 0x138, 0xF, 0xF, false);   // wave_shr:1, lane 0 <- carry
                            if (do_count) {
                                QuantK ck;
                                ck.inv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(k << 2, __builtin_bit_cast(int, kc.inv)));
                                ck.bias = __builtin_fmaf(0.25f, ck.inv, 0.5f);
                                const int qa = quant_mag(magf, ck);
                                // bits | deficit << 16, spread here (carried out of the chunk as the 16-bit table entry it was masked again)
                                uint32_t lw = L.ac_len16[lut_index(is_ac ? qa : 0, is_ac ? k - ka - 1 : 0)];
                                asm("" : "+v"(lw));          // (pins the zero-extension to the load: used in a later block it is a separate mask)
                                cnt16 = (int)__builtin_amdgcn_perm(0u, lw, 0x0C010C00u);
                            }
                            if (!single) kcarry_a = __builtin_amdgcn_readlane(k, 63);
                        }
                        const int q = quant_mag(magf, ek);                     // <= 2048
                        if (low) {
                            // Previous surviving entry.  A list at the count scale is ONE chunk (longer ones are recompacted first), so
                            // the slots behind it are free: every survivor leaves its scan position at slot (its rank among the
                            // survivors) and reads its predecessor's one slot down -- rank (mbcnt), one address, a masked store and a
                            // load, where the mask-below-me / count-leading-zeros / bpermute route was thirteen vector instructions.
                            // Rank 0 is lane 0, the macroblock's first DC slot: what it reads is never used.
                            static_assert(single || !low, "a list at the count scale is one chunk");
                            const uint64_t sm = ballot_ne0(q) | ballot_eq0(k);     // (dead lanes: |n| = 0 at position 63)
                            ncodes = (int)__builtin_popcountll(sm);
                            const uint32_t kslot = kbase + 4u * (uint32_t)wave::popc_below(sm);
                            unsigned long long sv;
                            asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                                         : "=&s"(sv) : "s"(sm), "v"(kslot), "v"(k) : "memory");          // survivors only
                            typedef const uint32_t __attribute__((address_space(3))) * LdsWord;
                            kprev = (int)*(LdsWord)(uintptr_t)(kslot - 4u);                      // (a wavefront's LDS accesses complete in order)
                            int last = 63 - __builtin_clzll(sm);              // (DC slots survive: sm != 0)
                            asm("" : "+s"(last));                             // (the subtraction stays scalar: one vector compare)
                            is_last = lane == last;
                        } else {
                            is_last = i == count - 1;
                            kprev = __builtin_amdgcn_update_dpp(kcarry_b, k, 0x138, 0xF, 0xF, false);
                            if (!single) kcarry_b = __builtin_amdgcn_readlane(k, 63);
                            ncodes = count - base < 64 ? count - base : 64;      // a list at the emit scale: every entry is a code
                        }
                        // one index, one select: row min(q, 41), column run = k - kprev - 1 (silent lanes: entry 0, no bits)
                        const int run_raw = k + ~kprev;
                        const unsigned qrow = (unsigned)q > (unsigned)(BS_LUT_MAX_LEVEL + 1) ? (unsigned)(BS_LUT_MAX_LEVEL + 1) : (unsigned)q;
                        const int li = (int)__umul24(qrow, (unsigned)BS_LUT_W) + run_raw;
                        const uint32_t entry = L.ac_code[is_ac ? li : 0];
                        len = (int)(entry >> 24);
                        code = (entry & 0x1FFFFu) | negbit;
                        if (ballot_eq<BS_ESCAPE_BITS>(len)) {
                            // (rare, so behind a wave-uniform branch) escape: 6 bits 000001, run, 10-bit level (mdec.c:258) clamped to
                            // -512 .. 510 (mdec.c:260-267; only the escape's payload ever sees a level that large)
                            asm volatile("; escape codes");     // (an asm statement keeps the branch around this block: left alone, the
                                                                //  compiler runs the eight instructions under an empty mask instead)
                            const int lim = 510 + 2 * (int)negbit;
                            const int qc = q > lim ? lim : q;
                            const int sl = (qc ^ -(int)negbit) + (int)negbit;
                            const uint32_t esc = (1u << 16) | ((uint32_t)run_raw << 10) | ((uint32_t)sl & 0x3FFu);
                            code = len == BS_ESCAPE_BITS ? esc : code;
                        }
                        deficit = (int)((entry >> BS_LUT_DEFICIT_SHIFT) & 0xFu);
                        if (CODEC == 0) {
                            // v2 DC slot: the entry carries the quantised DC; 10 bits (mdec.c:451-453), every slot but the
                            // macroblock's first also carries the previous block's end-of-block code
                            if (is_dc) {
                                const uint32_t eob = i != 0 ? 2u << 10 : 0u;       // (a per-lane constant of a one-chunk list: one and-or)
                                code = (e & 0x3FFu) | eob;
                                len = i != 0 ? 12 : 10;
                            }
                        } else {
                            // v3 DC slots: block index = number of DC slots before this one in the macroblock's list
                            const uint64_t dcmask = dcm;
                            if (is_dc) {
                                const int bi = bcarry + wave::popc_below(dcmask);
                                dc_code<CODEC>((int)L.dcv[mbe * 6 + bi], bi >= 2, L.dc_plen, L.dc_prefix, len, code);
                                if (bi > 0) {                               // carry the previous block's end-of-block code
                                    code |= 2u << len;
                                    len += 2;
                                }
                            }
                            if (!single) bcarry += (int)__builtin_popcountll(dcmask);
                        }
                        // the macroblock's LAST code carries the last block's end-of-block code behind it, the way every other
                        // end-of-block code rides in front of the next block's DC code: no separate two-bit write by one lane
                        {
                            const int add = is_last ? 2 : 0;
                            code = (code << add) | (uint32_t)add;
                            len += add;
                        }
                    };
                    // staging for a macroblock of `mb_bits` bits (all six end-of-block codes included): returns its bit position
                    bool have_room = true;
                    auto stage_alloc = [&](uint32_t mb_bits) -> uint32_t {
                        const int ndw = (int)((mb_bits + 31u) >> 5);
                        int off = 0;
                        if (lane == 0) {
                            // (one masked region: the allocation and the macroblock's record -- position | bits << 16, both below 2^16)
                            off = atomicAdd(&L.scalars[S_STG_NEXT], ndw);
                            *(uint32_t*)((char*)L.rec + rec_byte) = (uint32_t)off | (mb_bits << 16);
                        }
                        off = __builtin_amdgcn_readfirstlane(off);
                        have_room = off + ndw <= job.stg_words;      // (a pass that ran out of room is seen at its end: S_STG_NEXT > stg_words)
                        emit_bits += (int)mb_bits;
                        return (uint32_t)off;            // (in dwords: base and bit offset stay apart, put_codes)
                    };
                    // (lanes without a code stay out: an OR of nothing is still an LDS atomic on a neighbour's dword -- tried, +57 % bank
                    //  conflict cycles and 12 % slower on 640x480, whose chunks have more idle lanes)
                    auto put_codes = [&](uint32_t base_dw, uint32_t bit, int len, uint32_t code) {
                        if (have_room) {        // (wave-uniform: a branch, not a mask)
                            if (len) put_bits_lds(L.stg, base_dw, bit, len, code);
                        }
                    };
                    bool low = cs && !dense;       // the list holds the count scale's codes
                    if (low && count > 64) {
                        // A long list at the count scale (busy macroblocks, fine scales): count its codes chunk by chunk and keep
                        // only the entries that are still non-zero at the emit scale -- typically fewer than half.  They are
                        // compacted in place (a survivor's slot is never behind its own entry, and a wavefront's LDS accesses
                        // complete in order); what follows then sees a list made at the emit scale.
                        int sc = 0, kcarry = 0;
                        for (int base = 0; base < count; base += 64) {
                            const int i = base + lane;
                            const bool live = i < count;
                            uint32_t e = clist[i];
                            e = live ? e : (63u << 17);
                            const int k = (int)__builtin_amdgcn_ubfe(e, 17, 6);
                            const bool is_dc = k == 0;
                            const bool is_ac = live && !is_dc;
                            float magf;                                            // (as in chunk(): one SDWA convert)
                            asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(magf) : "v"(e));
                            QuantK ck, ek;
                            ck.inv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(k << 2, __builtin_bit_cast(int, kc.inv)));
                            ck.bias = __builtin_fmaf(0.25f, ck.inv, 0.5f);
                            ek.inv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(k << 2, __builtin_bit_cast(int, ke.inv)));
                            ek.bias = __builtin_fmaf(0.25f, ek.inv, 0.5f);
                            const int ka = __builtin_amdgcn_update_dpp(kcarry, k, 0x138, 0xF, 0xF, false);   // wave_shr:1, lane 0 <- carry
                            kcarry = __builtin_amdgcn_readlane(k, 63);
                            const int qa = quant_mag(magf, ck);
                            uint32_t lw = L.ac_len16[lut_index(is_ac ? qa : 0, is_ac ? k - ka - 1 : 0)];
                            asm("" : "+v"(lw));
                            acc_cnt += (int)__builtin_amdgcn_perm(0u, lw, 0x0C010C00u);     // bits | deficit << 16
                            const int qe = quant_mag(magf, ek);
                            const uint64_t sm = ballot_ne0(qe) | ballot_eq0(k);      // (dead lanes: |n| = 0 at position 63)
                            const uint32_t slot = (uint32_t)(uintptr_t)(clist + sc + wave::popc_below(sm));
                            unsigned long long sv;
                            asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                                         : "=&s"(sv) : "s"(sm), "v"(slot), "v"(e) : "memory");          // survivors only
                            sc += (int)__builtin_popcountll(sm);
                        }
                        wave_sync();
                        count = sc;
                        low = false;
                    }
                    if (count <= 64) {
                        int len, deficit, cnt16, nc;
                        uint32_t code;
                        if (low) chunk(yes, yes, yes, 0, len, code, deficit, cnt16, nc);
                        else chunk(no, no, yes, 0, len, code, deficit, cnt16, nc);
                        const int incl = wave::inclusive_scan_add(len);
                        const uint32_t base_dw = stage_alloc((uint32_t)__builtin_amdgcn_readlane(incl, 63));
                        put_codes(base_dw, (uint32_t)(incl - len), len, code);
                        acc_edef += deficit;
                        acc_cnt += cnt16;
                        n_codes += nc;
                    } else {
                        // longer lists: add up the lengths first (the allocation needs the macroblock's total), then write
                        int lsum = 0;
                        for (int base = 0; base < count; base += 64) {
                            int len, deficit, cnt16, nc;
                            uint32_t code;
                            chunk(no, no, no, base, len, code, deficit, cnt16, nc);      // (a list of several chunks is at the emit scale: see the recompaction above)
                            lsum += len;
                        }
                        const uint32_t base_dw = stage_alloc((uint32_t)wave::reduce_add(lsum));
                        uint32_t pos = 0;
                        kcarry_a = 0;
                        kcarry_b = 0;
                        bcarry = 0;
                        for (int base = 0; base < count; base += 64) {
                            int len, deficit, cnt16, nc;
                            uint32_t code;
                            chunk(no, no, no, base, len, code, deficit, cnt16, nc);
                            const int incl = wave::inclusive_scan_add(len);
                            put_codes(base_dw, pos + (uint32_t)(incl - len), len, code);
                            pos += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
                            acc_edef += deficit;
                            n_codes += nc;
                        }
                    }
                    wave_sync();   // the list is overwritten by the next macroblock's tiles
                };
                if (count_scale) emit_mb(yes);
                else emit_mb(no);
            }
            // outside the passes a group runs short, latency-bound phases (decisions, scans, merge): let them cut ahead of the
            // partner group's VALU stream instead of queueing behind it
            if (WAVES == kWavesSmall) {
                switch ((job.prio_pattern >> 16) & 3u) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
            flush();
            group_sync(3);
            const int verdict = L.scalars[S_ABORT];
            aborted = (verdict >> 8) == n_pass;          // (this pass's verdict; an older one is stale)
            if (STATS && tid == 0 && aborted && !first_abort) first_abort = verdict & 0xFF;
            if (STATS && aborted && n_pass <= 4) trace |= 0x80ull << (24 + 8 * n_pass);
            if (scan_early && wid == 1 && emit_scale && !aborted) scan_offsets(lane);
            if (tid == 0) {
                MdecSearch st = *srch;
                // A frame that needs ANOTHER full pass (a wrong first guess) is started from scratch anyway -- the DCT is recomputed,
                // the staging area rebuilt -- so it need not be this group that does it.  While this group still holds a ticket for
                // a fresh frame it hands the frame on: (frame, scale to start from) goes into a queue in global memory and the group
                // moves to its next fresh frame; groups that run out of fresh frames empty the queue (top of the frame loop).  The
                // launch used to last as long as the group that drew two such frames (noise +-8: 250 us against a median group's
                // 170); retries are now drawn like tickets.  The result of a frame never depends on who encodes it or from which guess.
                auto hand_on = [&](const MdecPass& np) -> bool {
                    if (!job.retry || np.done || !np.emit_scale || L.scalars[S_RETRY] != 0 ||
                        (L.scalars[S_RUN_LEFT] == 0 && (unsigned)L.scalars[S_NEXT_DRAW] >= fresh_draws())) return false;
                    const unsigned slot = (unsigned)(atomicAdd(queue_state(job), 1ull << kQueueReservedShift) >> kQueueReservedShift) & kQueueMask;
                    if (atomicExch(&retry_slots(job)[slot], (unsigned)f | ((unsigned)np.emit_scale << 24)) == kRetryAbandoned) {    // (the host sizes the queue for one entry per frame)
                        atomicExch(&retry_slots(job)[slot], (unsigned)in_loop((int)kRetryEmpty));      // the group this slot belonged to has left: the frame stays here
                        return false;
                    }
                    L.scalars[S_DEFER] = 1;
                    L.scalars[S_DONE] = 1;
                    // What does the failed guess say about the NEXT frame?  One frame out of line (content at a scale boundary: every
                    // eighth frame of the synthetic +-8 batch) says nothing -- the group keeps the answer of the last frame it
                    // finished; two in a row are a change of scene -- the group takes over the scale this frame wants.  (Always taking
                    // it over cost 3 % on +-8 content, never taking it over 4 % on a cold context.)
                    int old_hint = L.scalars[S_HINT_BUDGET] == max_size ? L.scalars[S_HINT] : 0;
                    if (old_hint < 1 && (L.scalars[S_SHARED_HINT] >> 8) == max_size) old_hint = L.scalars[S_SHARED_HINT] & 0xFF;   // (a group's first frame starts from the previous launch's last answer)
                    L.scalars[S_HINT] = old_hint < 1 || L.scalars[S_PUSHED] ? np.emit_scale : old_hint;
                    L.scalars[S_HINT_BUDGET] = max_size;
                    L.scalars[S_HINT_FRAME] = in_loop(-2);          // (not the neighbour's ANSWER: foreign)
                    L.scalars[S_PUSHED] = 1;
                    if (L.scalars[S_FOREIGN] && !L.scalars[S_PILOTED]) {      // the frame left with its answer unknown; the foreign hint it started from was not it (a frame that started from the pilot's guess says nothing about the hint)
                        L.scalars[S_F_TRIED] = L.scalars[S_F_TRIED] + 1;
                        L.scalars[S_F_WRONG] = L.scalars[S_F_WRONG] + 1;
                        L.scalars[S_DISTRUST] = L.scalars[S_DISTRUST] + 0x100;
                    }
                    return true;
                };
                if (aborted) {
                    // the pass was cut short: no evaluation to record, the staging area holds a partial stream
                    st.staged = 0;
                    const MdecPass np = mdec_search_next(st, verdict & 0xFF, limit_bits, fixed_bits);
                    *srch = st;
                    L.scalars[S_PASS_COUNT] = np.count_scale;
                    L.scalars[S_PASS_EMIT] = np.emit_scale;
                    L.scalars[S_DONE] = np.done;
                    L.scalars[S_CNT_F] = 0; L.scalars[S_CNT_D] = 0; L.scalars[S_CK_DONE] = 0; L.scalars[S_CK_WAVES] = 0;
                    L.scalars[S_EMIT_BITS] = 0; L.scalars[S_EMIT_D] = 0; L.scalars[S_NNZ] = 0;
                    L.scalars[S_MB_NEXT] = 2 * kWavesPerGroup;
                    const int vg = verdict & 0xFF, far = guess > 8 ? guess >> 2 : 2;
                    if (n_pass == 1 && !L.scalars[S_PILOTED] && !np.done && (vg - guess >= far || guess - vg >= far)) {
                        L.scalars[S_REPILOT] = vg;
                        L.scalars[S_DEFER] = 2;
                        L.scalars[S_DONE] = 1;          // leaves the pass loop; the frame starts over from the pilot (below)
                    } else {
                        (void)hand_on(np);
                    }
                } else {
                if (count_scale) {
                    const int tb = L.scalars[S_CNT_F] + fixed_bits;
                    mdec_search_note(st, count_scale, tb, tb - L.scalars[S_CNT_D], limit_bits);
                }
                if (emit_scale) {
                    const int tb = L.scalars[S_EMIT_BITS] + 10;        // + end-of-frame code
                    mdec_search_note(st, emit_scale, tb, tb - L.scalars[S_EMIT_D], limit_bits);
                    st.staged = L.scalars[S_STG_NEXT] > job.stg_words ? 0 : emit_scale;       // (the staging area ran out during this pass)
                    L.scalars[S_TOTAL_BITS] = tb;
                }
                const MdecPass np = mdec_search_next(st, guess, limit_bits, fixed_bits);
                *srch = st;
                L.scalars[S_PASS_COUNT] = np.count_scale;
                L.scalars[S_PASS_EMIT] = np.emit_scale;
                L.scalars[S_DONE] = np.done;
                L.scalars[S_RESULT] = st.best;
                if (!np.done) {
                    L.scalars[S_CNT_F] = 0; L.scalars[S_CNT_D] = 0; L.scalars[S_CK_DONE] = 0; L.scalars[S_CK_WAVES] = 0;
                    L.scalars[S_MB_NEXT] = 2 * kWavesPerGroup;
                    if (np.emit_scale) { L.scalars[S_EMIT_BITS] = 0; L.scalars[S_EMIT_D] = 0; L.scalars[S_NNZ] = 0; }
                    (void)hand_on(np);
                }
                }
            }
            if (kStopAfter && tid == 0) L.scalars[S_DONE] = 1;
            group_sync(4);
        }
        mark(3);   // passes
        const int defer = L.scalars[S_DEFER];
        if (defer) {
            // handed on: nothing of this frame is written here
            group_sync(5);      // everyone has read the verdict: the scalars may go
            if (defer == 2) {
                // ... or sent back to the pilot: the frame loop's next turn is the same frame again, with the per-frame state as a
                // fresh frame finds it and the ticket state untouched.  (A turn of the frame loop, not a loop around the pilot and
                // the passes: that loop cost EVERY frame 1.5 % -- values of the frame's start kept alive across the passes, in a
                // kernel that has no register to spare -- for the one frame in fifteen that takes it.)
                for (int i = tid; i < job.stg_words; i += kThreads) L.stg[i] = 0u;
                if (tid < S_KEEP0 && tid != S_RETRY) L.scalars[tid] = 0;       // (a frame taken from the retry queue stays one: it is not handed on again)
                if (tid == 0) next_draw = (unsigned)L.scalars[S_NEXT_DRAW];
                if (STATS) { carry_pass += n_pass; carry_guess0 = guess0; carry_abort = first_abort; }
            } else {
                n_done++;
                end_of_frame(tid);
            }
            group_sync(5);
            continue;
        }
        n_done++;
        if (STATS) { n_pass += carry_pass; carry_pass = 0; }      // (a frame sent back to the pilot: its first attempt's pass counts)
        if (STATS && tid == 0 && f < PSXHIP_MDEC_TRACE_FRAMES)      // per-frame record: first guess | first abort verdict << 8 | answer << 16 | passes << 24
            job.stats[PSXHIP_MDEC_STATS_FRAME0 + f] = (unsigned long long)(guess0 & 0xFF) | (unsigned long long)(first_abort & 0xFF) << 8 |
                                                      (unsigned long long)(L.scalars[S_RESULT] & 0xFF) << 16 | (unsigned long long)(n_pass & 0xFF) << 24 | trace;
        if (STATS && tid == 0) {
            pass_sum += (unsigned)n_pass;
            for (int c = 0; c < 6; c++) pass_hist[c] += (n_pass > 5 ? 5 : n_pass) == c ? 1u : 0u;
        }

        const int scale = L.scalars[S_RESULT];
        if (tid == 0) {
            // The hint the next launch's groups start from is the answer of this launch's LAST frame (by index, not by time of
            // finishing): the neighbour of the next batch's first frame when batches follow each other, and an unbiased draw
            // from the batch's answers otherwise.  Not "the last frame finished": the frames that finish last are the ones
            // that needed a second pass, i.e. the minority answer -- it would poison the start of every following launch
            // (measured: 48 % of the frames restarted at the quarter mark instead of 16 %).
            if (scale < 64 && f == PSX_JOB_INT(n_frames) - 1) *job.hint = (unsigned)scale | ((unsigned)max_size << 8);
            L.scalars[S_HINT] = scale < 64 ? scale : 0;
            L.scalars[S_HINT_BUDGET] = max_size;
            L.scalars[S_HINT_FRAME] = f;
            L.scalars[S_PUSHED] = 0;
            if (L.scalars[S_PILOTED]) {
                L.scalars[S_P_TRIED] = L.scalars[S_P_TRIED] + 1;
                L.scalars[S_P_WRONG] = L.scalars[S_P_WRONG] + (guess != scale ? 1 : 0);
            }
            const int foreign = L.scalars[S_FOREIGN];
            if (foreign) {
                const int wrong = foreign != scale ? 1 : 0;
                L.scalars[S_F_TRIED] = L.scalars[S_F_TRIED] + 1;
                L.scalars[S_F_WRONG] = L.scalars[S_F_WRONG] + wrong;
                const int d0 = L.scalars[S_DISTRUST];
                L.scalars[S_DISTRUST] = wrong ? d0 + 0x100 : (d0 & 1);
            }
        }
        uint8_t* outp;
        psxhip_mdec_result_t* b_results;
        {
            outp = job.batch[0].out + (size_t)fl * job.out_stride;
            b_results = job.batch[0].results + fl;
            if (PSX_BATCHES_MANY()) {
                BatchPtr bt = batch_table();
                outp = bt[bi].out + (size_t)fl * job.out_stride;
                b_results = bt[bi].results + fl;
            }
        }

        if (scale >= 64 || bad_budget) {
            // nothing fits (the reference asserts, mdec.c:723): zero output, flag the result
            if (!bad_budget)
                for (int i = tid; i < max_size; i += kThreads) outp[i] = 0;
            if (tid == 0) {
                psxhip_mdec_result_t r;
                r.quant_scale = in_loop(64); r.bytes_used = 0; r.blocks_used = 0; r.uncomp_hwords_used = 0;
                *b_results = r;
            }
            group_sync(5);      // everyone has read the verdict: the scalars may go
            end_of_frame(tid);
            group_sync(5);
            continue;
        }

        if (!scan_early) {
            if (wid == 0) scan_offsets(lane);
            group_sync(5);
        }

        // =====================================================================================
        // Merge: macroblock streams (dword-aligned in staging) -> their bit positions in the frame image.
        // Output dword j of a macroblock at bit offset D = 32 w + sh receives  stg[j-1] << (32 - sh) | stg[j] >> sh
        // (one v_alignbit); neighbouring macroblocks meet inside a dword, hence ds_or.
        // =====================================================================================
        // The frame image (8-byte header + bitstream, one dword per 4 output bytes) is assembled in LDS one TILE of
        // out_tile dwords at a time -- one tile for every budget up to 8 KiB -- so that the LDS need does not grow with
        // the budget twice (staging + image).
        const int total_bits = L.scalars[S_TOTAL_BITS];
        const int image_words = (max_size + 3) >> 2;          // dwords of the output row, the last one possibly partial
        uint32_t* o32 = (uint32_t*)outp;
        for (int t0 = 0; t0 < image_words; t0 += job.out_tile) {
            const int t1 = t0 + job.out_tile;
            {
                // four macroblocks per wavefront, 16 lanes each (a macroblock's stream is typically 6..16 dwords)
                const int j0 = lane & 15;
                const int tile = t0 / job.out_tile;
                int mb_lo = 0, mb_hi = nmb;
                if (image_words > job.out_tile) {
                    // tiles without a starting macroblock (behind the end of the stream) keep the sentinel nmb
                    mb_lo = L.scalars[S_TILE_FIRST0 + tile] - 1;
                    if (mb_lo < 0) mb_lo = 0;
                    mb_hi = tile + 1 <= kMaxTiles ? L.scalars[S_TILE_FIRST0 + tile + 1] : nmb;
                    for (int u = tile + 2; mb_hi == nmb && u <= kMaxTiles; u++) mb_hi = L.scalars[S_TILE_FIRST0 + u];   // (a tile nobody starts in)
                }
                for (int base = mb_lo + wid * 4; base < mb_hi; base += kWavesPerGroup * 4) {
                    const int mbe = base + (lane >> 4);
                    const bool valid = mbe < mb_hi;
                    const uint32_t r = valid ? L.rec[mbe] : 0u;
                    const int off = (int)(r & 0xFFFFu);
                    int ndw = valid ? (int)(((r >> 16) + 31u) >> 5) : -1;
                    const uint32_t D = valid ? L.mb_off[mbe] : 0u;
                    const int g0 = 2 + (int)(D >> 5);             // image dword of this macroblock's first staging dword
                    const uint32_t sh = D & 31u;
                    if (g0 + ndw < t0 || g0 >= t1) ndw = -1;      // nothing of it in this tile
                    for (int j = j0; wave::ballot(j <= ndw) != 0; j += 16) {
                        const uint32_t cur = j < ndw ? L.stg[off + j] : 0u;
                        const uint32_t prv = (j >= 1 && j <= ndw) ? L.stg[off + j - 1] : 0u;
                        const uint32_t v = __builtin_amdgcn_alignbit(prv, cur, sh);
                        const int g = g0 + j;
                        if (v && j <= ndw && g >= t0 && g < t1) atomicOr(&L.out[g - t0], v);
                    }
                }
            }
            // ---- end-of-frame code, header, results (mdec.c:710-754): one thread, alongside the merge (the end-of-frame code
            //      shares its dwords with the last macroblock, hence atomic; nothing else lands on the two header dwords)
            if (tid == kThreads - 1) {
                // end-of-frame code (mdec.c:647-651,710): 10 bits at stream bit total_bits - 10 = image bit 64 + that
                const uint32_t pos = 64u + (uint32_t)(total_bits - 10);
                const int w = (int)(pos >> 5);
                const uint32_t sb = pos & 31u;
                const uint64_t tt = (uint64_t)(CODEC == 0 ? 0x1FFu : 0x3FFu) << (64 - sb - 10);
                const uint32_t hi = (uint32_t)(tt >> 32), lo = (uint32_t)tt;
                if (w >= t0 && w < t1) atomicOr(&L.out[w - t0], hi);
                if (lo && w + 1 >= t0 && w + 1 < t1) atomicOr(&L.out[w + 1 - t0], lo);
                if (t0 == 0) {
                    int hwords = L.scalars[S_NNZ] + 2 * nblk + 2;
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
                    *b_results = r;
                }
            }
            group_sync(5);
            if (t0 == 0) mark(4);   // scan + merge + header

            // ---- write-out: staging dword holds two MSB-first 16-bit words; each word is stored low byte
            //      first (mdec.c:321-333), i.e. the output dword is the staging dword rotated by 16.
            {
                const int full = max_size >> 2;                  // whole output dwords
                const int n = (full < t1 ? full : t1) - t0;      // ... of them in this tile
                const int tail = max_size & 3;
                if (tail && full >= t0 && full < t1 && tid < tail) {
                    const uint32_t v = L.out[full - t0];
                    const uint32_t o = (v >> 16) | (v << 16);
                    outp[full * 4 + tid] = (uint8_t)(o >> (8 * tid));
                }
                if (tail) group_sync(5);      // (the tail bytes' dword is read by up to three threads before it is cleared)
                // ... and the tile is cleared for whatever is assembled in it next, by the thread that read each dword
                for (int i = tid; i < job.out_tile + 1; i += kThreads) {
                    const uint32_t v = L.out[i];
                    if (i < n) o32[t0 + i] = (v >> 16) | (v << 16);
                    L.out[i] = 0u;
                }
                if (t1 >= image_words) end_of_frame(tid);
            }
            group_sync(5);
        }
        mark(5);   // header + write-out
        if (STATS && n_done == 1) t_first = wall_clock64();
    }

    unsigned long long t_end = 0;
    if (STATS) t_end = wall_clock64();
    if (STATS && tid == 0) {
        atomicAdd(&job.stats[0], (unsigned long long)n_done);
        atomicAdd(&job.stats[1], (unsigned long long)pass_sum);
        for (int c = 0; c < 6; c++) {
            atomicAdd(&job.stats[2 + c], (unsigned long long)pass_hist[c]);
            atomicAdd(&job.stats[PSXHIP_MDEC_STATS_PHASE0 + c], phase_ticks[c]);
        }
    }
    if (STATS) {
        // per-wavefront sums -> one set of global atomics per group (thousands of wavefronts adding to the same few words
        // at the end of the kernel would dominate what is being measured)
        const unsigned long long t_leave = wall_clock64();
        __syncthreads();
        if (tid < 8) L.stg[tid] = 0u;
        __syncthreads();
        if ((tid & 63) == 0) {
            atomicAdd(&L.stg[6], (uint32_t)wait_ticks);
            atomicAdd(&L.stg[7], (uint32_t)(t_leave - t_entry));
            for (int c = 0; c < 6; c++) atomicAdd(&L.stg[c], (uint32_t)wait_cat[c]);
        }
        __syncthreads();
        if (tid == 0) {
            atomicAdd(&job.stats[PSXHIP_MDEC_STATS_PHASE0 + 6], (unsigned long long)L.stg[6]);
            atomicAdd(&job.stats[PSXHIP_MDEC_STATS_PHASE0 + 7], (unsigned long long)L.stg[7]);
            for (int c = 0; c < 6; c++) atomicAdd(&job.stats[PSXHIP_MDEC_STATS_PHASE0 + 8 + c], (unsigned long long)L.stg[c]);
        }
    }
    if (STATS && tid == 0 && blockIdx.x < PSXHIP_MDEC_TRACE_GROUPS) {
        unsigned long long* t = job.stats + PSXHIP_MDEC_STATS + 4 * blockIdx.x;
        t[0] = t_entry;
        t[1] = t_end;
        t[2] = (unsigned long long)n_done | ((t_start - t_entry) & 0xFFFFFFull) << 8 | ((t_first ? t_first - t_entry : 0ull) & 0xFFFFFFull) << 32;
        // HW_REG_HW_ID (wave slot, SIMD, CU, shader array / engine) | HW_REG_XCC_ID << 32: which piece of the chip ran the group
        t[3] = (unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4) |
               (unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20) << 32;
    }
    // ---- the last workgroup to leave re-arms the ticket counters for the next launch (launches on one context are
    //      stream-ordered, see psxav_hip.h)
    if (tid < 64) {       // (the group's first wavefront: lane 0 counts the group out, all 64 lanes re-arm the slots)
        unsigned lo = 0, hi = 0;
        if (tid == 0) {
            unsigned tried = (unsigned)L.scalars[S_F_TRIED], wrong = (unsigned)L.scalars[S_F_WRONG];
            if (tried > kTrustCap) { wrong = (wrong * kTrustCap + tried / 2) / tried; tried = kTrustCap; }
            // (the pilot's counts go in first, and the count-out below is made to depend on that atomic's RETURN: whoever sees this
            //  group gone has its pilot counts in the word too -- no fence)
            // (a group that ran no pilot -- every group of a launch on content whose hints hold -- has nothing to hand in and does not
            //  pay the atomic's round trip on its way out: 2 us at the end of a 127 us launch)
            unsigned zero = 0;
            if (L.scalars[S_P_TRIED]) {
                const unsigned long long pr = atomicAdd((unsigned long long*)&job.ticket[kPilotWord],
                                                        (unsigned long long)(unsigned)L.scalars[S_P_WRONG] | (unsigned long long)(unsigned)L.scalars[S_P_TRIED] << 32);
                asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"((unsigned)pr));
            }
            const unsigned long long mine = (1ull + zero) | (unsigned long long)wrong << kLeaveWrongShift | (unsigned long long)tried << kLeaveTriedShift;
            const unsigned long long lw = atomicAdd(leave_word(job), mine) + mine;       // groups gone | abandoned queue slots << 16 | wrong << 32 | tried << 48
            lo = (unsigned)lw;
            hi = (unsigned)(lw >> 32);
        }
        lo = (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
        hi = (unsigned)__builtin_amdgcn_readfirstlane((int)hi);
        const unsigned left = lo - 1u;
        if ((left & 0xFFFFu) == gridDim.x - 1u) {
            if (job.retry && (left >> 16)) {      // queue slots that were given up and never reserved still say so
                // (all 64 lanes: with two launches in flight a group that finishes before every group of its launch has started may
                //  not wait, so whole launches' worth of pop tickets are given up -- one thread storing 500 words one after the
                //  other held the launch's end up by 12 us)
                unsigned long long w = 0;
                if (tid == 0) w = atomicAdd(queue_state(job), 0ull);
                const unsigned wh = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w >> 32));
                unsigned head = (wh >> (kQueueHeadShift - 32)) & kQueueMask;
                const unsigned cap = (unsigned)PSX_JOB_INT(retry_cap);
                if (head > cap) head = cap;
                for (unsigned i = ((wh >> (kQueueReservedShift - 32)) & kQueueMask) + (unsigned)tid; i < head; i += 64u) retry_slots(job)[i] = kRetryEmpty;
            }
            if (tid == 0) {
                // the launch's verdict on foreign hints, for the launches after it (a launch that judged fewer than eight leaves it alone)
                // Foreign hints are distrusted when more than one in four was wrong AND the pilot does better (a launch that ran no
                // pilots has no say on the second: it goes by the first alone, and the launch after it -- which pilots -- decides
                // whether that was a good idea: at 640x480 a dozen macroblocks are a poor sample, the pilot is right 60 % of the time
                // on content whose hints are right 87 %).
                const unsigned all_wrong = hi & 0xFFFFu, all_tried = hi >> 16;
                if (all_tried >= 8u) {
                    // (without a pilot record the bar is one in THREE: content whose answer flips between two scales one frame in eight
                    //  has its hints -- the answer of a frame 512 positions back -- wrong 2 x 1/8 x 7/8 = 22 % of the time, and with the
                    //  bar at 25 % one launch in ten of 640x480 v3 crossed it by chance and was followed by a launch of pilots)
                    bool distrust_next = false;
                    if (4u * all_wrong > all_tried) {          // (only then is the pilot's record looked at: a round trip at the very end of the launch)
                        const unsigned long long pw = atomicAdd((unsigned long long*)&job.ticket[kPilotWord], 0ull);
                        const unsigned p_wrong = (unsigned)pw, p_tried = (unsigned)(pw >> 32);
                        distrust_next = p_tried >= 8u ? (unsigned long long)p_wrong * all_tried < (unsigned long long)all_wrong * p_tried : 3u * all_wrong > all_tried;
                    }
                    job.hint[kDistrustWord] = distrust_next ? 1u : 0u;
                }
                *(unsigned long long*)&job.ticket[kPilotWord] = 0ull;
                *leave_word(job) = 0ull;
                job.ticket[kStartedWord] = 0u;
                *queue_state(job) = 0ull;       // tickets and queue counters (every slot that was filled has been vacated by the group that took it)
            }
            __threadfence();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The 8x8 forward DCT on its own, exactly as the frame kernel runs it (same fdct8_pk, same lane mapping, same LDS
// transposes): 6 blocks per wavefront.  in: level-shifted samples (-128..127, what mdec.c:619-633 hands to
// AVDCT.fdct), out: the 64 coefficients in raster order, like the in-place result of mdec.c:640.  This is the surface
// tools/check_fdct_vs_ffmpeg.c diffs against a real libavcodec, and tests/test_gpu_mdec.py against the oracle.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mdec_fdct_probe_kernel(const int16_t* in, int16_t* out, int n_blocks) {
    __shared__ __attribute__((aligned(16))) int16_t tile[6 * kTileStride];
    const int lane = (int)threadIdx.x;
    const int blk = lane >> 3, r8 = lane & 7;
    const int b = (int)blockIdx.x * 6 + blk;
    const bool live = lane < 48 && b < n_blocks;
    int d[8];
    {
        // row pass: the lane's 8 raw samples 0..255 as bytes, through the matrix pipe like the frame kernel
        uint32_t x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live) {
            const int16_t* p = in + (size_t)b * 64 + r8 * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = (uint32_t)((int)p[i] + 128) & 0xFFu;
        }
        fdct8_rows_mfma(x[0] | x[1] << 8 | x[2] << 16 | x[3] << 24, x[4] | x[5] << 8 | x[6] << 16 | x[7] << 24, dct_row_a_operand(lane), d);
        const int n = lane & 31;
        int16_t* t0 = tile + (n >> 3) * kTileStride + (n & 7) + (lane >> 5) * 32;
        store_row_outputs(t0, d);
        if (n < 16) store_row_outputs(t0 + 4 * kTileStride, d + 4);
    }
    __syncthreads();
    if (live) {
        const uint4 q = *(const uint4*)&tile[blk * kTileStride + r8 * 8];
        fdct8_col_acc(q.x, q.y, __builtin_amdgcn_alignbit(q.w, q.w, 16), __builtin_amdgcn_alignbit(q.z, q.z, 16), d);
        const uint32_t o[4] = {pack_sh17(d[1], d[0]), pack_sh17(d[3], d[2]), pack_sh17(d[5], d[4]), pack_sh17(d[7], d[6])};      // as the frame kernel packs them
#pragma unroll
        for (int v = 0; v < 8; v++) out[(size_t)b * 64 + v * 8 + r8] = (int16_t)(o[v >> 1] >> (16 * (v & 1)));
    }
}

}  // namespace

// The drop-in's one-frame-per-call pattern (encode_frame_bs, mdec.c:580; filefmt.c:643): the frame is copied from page-locked
// host memory the device can see into HBM by a kernel of its own -- every lane one 16-byte load, all in flight at once -- instead
// of a DMA-engine copy, whose start-up costs more than moving 115 KB does.  bytes is a multiple of 16.
namespace {
__global__ __launch_bounds__(256) void mdec_stage_in_kernel(const uint4* src, uint4* dst, int n16) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < n16) dst[i] = src[i];
}
}  // namespace
extern "C" hipError_t psxhip_mdec_stage_in_launch(const void* src_mapped, void* d_dst, size_t bytes, void* stream) {
    const int n16 = (int)(bytes >> 4);
    if (n16 <= 0) return hipSuccess;
    hipLaunchKernelGGL(mdec_stage_in_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)src_mapped,
                       (uint4*)d_dst, n16);
    return hipGetLastError();
}

#include "mdec_split.inc"

extern "C" hipError_t psxhip_mdec_fdct_launch(const int16_t* d_in, int16_t* d_out, int n_blocks, void* stream) {
    if (n_blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(mdec_fdct_probe_kernel, dim3((unsigned)((n_blocks + 5) / 6)), dim3(64), 0, (hipStream_t)stream, d_in, d_out, n_blocks);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Host side of the kernel (called from psxhip_api.cpp through psxhip_internal.h)
// ---------------------------------------------------------------------------------------------
extern "C" size_t psxhip_mdec_lds_bytes(int nmb, int out_words, int stg_words, int large) {
    return lds_bytes(nmb, out_words, stg_words, large ? kWavesLarge : kWavesSmall);
}
extern "C" int psxhip_mdec_threads_per_group(int large) { return (large ? kWavesLarge : kWavesSmall) * 64; }

extern "C" hipError_t psxhip_mdec_upload_tables(void) {
    hipError_t e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_ac_len16), bs_ac_len16_lut, sizeof(bs_ac_len16_lut))) != hipSuccess) return e;
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

// iteration visiting stride: coprime with `trips`, near 0.38 * trips, so that any prefix of a pass samples the frame evenly
static int pick_it_step(int trips) {
    if (trips <= 2) return 1;
    const int want = (trips * 382 + 500) / 1000;
    for (int d = 0; d < trips; d++) {
        const int cand[2] = {want + d, want - d};
        for (int i = 0; i < 2; i++) {
            const int c = cand[i];
            if (c < 1 || c >= trips) continue;
            int x = c, y = trips;
            while (y) { const int t = x % y; x = y; y = t; }
            if (x == 1) return c;
        }
    }
    return 1;
}

// The order in which a pass's tickets visit the macroblocks: ticket t = (round r = t / waves, slot w = t % waves) visits
// raster index w + seq[r] * waves.  For the rounds of the first quarter -- what the checkpoint looks at -- seq[r] =
// (r * step) % trips with step coprime to trips and close to 0.382 trips: they are spread evenly over the frame.  Entries
// past the last macroblock (the final round may be partial) hold kNoMb.  Returns the number of tickets (trips * waves).
extern "C" int psxhip_mdec_pass_order(int width, int height, int large, uint32_t* out, int cap) {
    const int waves = large ? kWavesLarge : kWavesSmall;
    const int nx = width / 16, ny = height / 16, nmb = nx * ny;
    const int trips = (nmb + waves - 1) / waves, step = pick_it_step(trips);
    const int n = trips * waves;
    if (!out) return n;
    // The rounds before the checkpoint's mark (trips / 4 of them) are spread evenly over the frame, the rest follow in raster
    // order: a round's twelve macroblocks read 192-byte pieces of pixel rows, memory is fetched in 128-byte granules, and with
    // EVERY round scattered the half-used granules at a round's ends were gone from L2 before the neighbouring round came
    // (fetch 141 MB per 1000 frames of 320x240 against 115 MB of pixels; 123 MB with three quarters of the rounds in raster order).  Scattering all rounds in runs
    // of 3 or 5 neighbours instead (tools/gpu_pass_run_sweep.sh) saves as much but changes what the checkpoint samples -- three
    // adjacent macroblock rows are no sample of a picture -- and moved noisy content by -12 .. +14 %.
    // (experiment, PSXHIP_MDEC_SAMPLE_CLUSTER=c: the quarter's sample as clusters of c neighbouring macroblocks spread over the frame
    //  instead of whole rounds of `waves` neighbours -- 18 clusters of 4 where the default has 6 of 12 at 320x240)
    int cluster = 0;
    if (const char* e = getenv("PSXHIP_MDEC_SAMPLE_CLUSTER")) cluster = atoi(e);
    if (cluster > 0 && cluster < waves && trips >= 8) {
        const int K = (nmb + cluster - 1) / cluster, Q = (trips >> 2) * waves;
        int stepk = (K * 382 + 500) / 1000;
        for (;; stepk++) { int x = stepk, y = K; while (y) { const int t = x % y; x = y; y = t; } if (x == 1) break; }
        char* usedmb = (char*)calloc((size_t)nmb, 1);
        if (!usedmb) return -1;
        int t = 0;
        for (int j = 0; j < K && t < Q; j++) {
            const int k = (int)(((long long)j * stepk) % K);
            for (int m = k * cluster; m < (k + 1) * cluster && m < nmb && t < Q; m++) {
                if (t < cap) out[t] = (uint32_t)(m % nx) | (uint32_t)(m / nx) << 8;
                usedmb[m] = 1;
                t++;
            }
        }
        for (int m = 0; m < nmb; m++)
            if (!usedmb[m]) { if (t < cap) out[t] = (uint32_t)(m % nx) | (uint32_t)(m / nx) << 8; t++; }
        for (; t < n; t++) if (t < cap) out[t] = kNoMb;
        free(usedmb);
        return n;
    }
    int* seq = (int*)malloc((size_t)trips * sizeof(int));      // ticket round -> raster round (a permutation)
    char* used = (char*)calloc((size_t)trips, 1);
    if (!seq || !used) { free(seq); free(used); return -1; }
    const int spread = trips >= 8 ? trips >> 2 : trips;       // (the kernel's check_t: no checkpoint below 8 rounds)
    int k = 0;
    for (int r = 0; r < spread; r++) { seq[k] = (r * step) % trips; used[seq[k++]] = 1; }
    for (int rr = 0; rr < trips; rr++) if (!used[rr]) seq[k++] = rr;
    free(used);
    for (int t = 0; t < n && t < cap; t++) {
        const int r = t / waves, w = t % waves;
        const int rr = seq[r];
        const int m = w + rr * waves;
        out[t] = m < nmb ? (uint32_t)(m % nx) | (uint32_t)(m / nx) << 8 : kNoMb;
    }
    free(seq);
    return n;
}

// ... and the same order as the kernel reads it: per ticket {fy * 8 W | valid << 31, fx * 16 | 4 * encode-order index << 16} (an entry
// without a macroblock is all zero), followed by one all-zero entry (cap > n).  Returns the number of tickets n.
extern "C" int psxhip_mdec_pass_table(int width, int height, int large, uint32_t* out /* [2 * (n + 1)] */, int cap) {
    const int n = psxhip_mdec_pass_order(width, height, large, nullptr, 0);
    if (!out) return n;
    if (cap > n) { out[2 * n] = 0u; out[2 * n + 1] = 0u; }        // the entry tickets past the end are clamped onto
    uint32_t* o = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    if (!o) return -1;
    (void)psxhip_mdec_pass_order(width, height, large, o, n);
    const int ny = height / 16;
    for (int t = 0; t < n && t < cap; t++) {
        if (o[t] == kNoMb) { out[2 * t] = 0u; out[2 * t + 1] = 0u; continue; }
        const uint32_t fx = o[t] & 0xFFu, fy = o[t] >> 8;
        out[2 * t] = (fy * 8u * (uint32_t)width) | 0x80000000u;
        out[2 * t + 1] = (fx * 16u) | ((fx * (uint32_t)ny + fy) * 4u << 16);
    }
    free(o);
    return n;
}

// Frame tickets of a launch of n_frames frames on `groups` persistent workgroups (ticket_run() in the kernel reads the result):
// whole rounds of the grid in runs of 4 while at least four rounds are left, then whole rounds in runs of 2, and what remains --
// less than two rounds -- as one round of runs of 2 when it is more than one frame per group, else as single frames.  Every
// round of the grid is whole, so runs cost no balance: 1000 frames on 512 groups are 500 runs of 2 (as many frames per group as
// single tickets give), 1250 are 512 runs of 2 and 226 single frames, 4000 are 512 runs of 4 and 976 of 2.
// max_run: 1 = single frames only (the hand-out of the kernels before mdec-k3.7), 2 = no runs of 4.
extern "C" void psxhip_mdec_ticket_plan(int n_frames, int groups, int max_run, int* t4, int* t2, int* n_tickets) {
    int r = n_frames, a4 = 0, a2 = 0;
    if (groups < 1) groups = 1;
    if (max_run >= 4) { a4 = groups * (r / (4 * groups)); r -= 4 * a4; }
    if (max_run >= 2) {
        const int whole = groups * (r / (2 * groups));
        a2 = whole;
        r -= 2 * whole;
        if (r > groups) { a2 += r / 2; r &= 1; }
    }
    *t4 = a4;
    *t2 = a2;
    *n_tickets = a4 + a2 + r;
}

extern "C" hipError_t psxhip_mdec_launch(const psxhip_mdec_launch_t* a) {
    const int waves_ = a->large ? kWavesLarge : kWavesSmall;
    FrameJob job;
    memset(&job, 0, sizeof job);
    job.n_batches = a->n_batches;
    for (int i = 0, first = 0; i < a->n_batches && i < kMaxBatches; i++) {
        job.batch[i].frames = a->batches[i].d_frames;
        job.batch[i].out = a->batches[i].d_out;
        job.batch[i].results = a->batches[i].d_results;
        job.batch[i].max_sizes = a->batches[i].d_frame_max_sizes;
        job.first[i] = first;
        first += a->batches[i].n_frames;
    }
    for (int i = a->n_batches; i < kMaxBatches; i++) job.first[i] = 0x7FFFFFFF;
    job.frame_stride = a->frame_stride;
    job.width = a->width;
    job.height = a->height;
    job.nx = a->width / 16;
    job.ny = a->height / 16;
    job.nmb = job.nx * job.ny;
    job.n_frames = a->n_frames;
    job.n_tickets = a->n_tickets;
    job.t4 = a->t4;
    job.t2 = a->t2;
    job.uniform_max_size = a->uniform_max_size;
    job.out_stride = a->out_stride;
    job.out_words = a->out_words;
    job.out_tile = a->out_tile;
    job.max_frame_size = a->max_frame_size;
    job.stg_words = a->stg_words;
    job.ticket = a->d_ticket;
    job.hint = a->d_hint;
    job.retry = a->d_retry;
    job.retry_cap = a->retry_cap;
    job.retry_patience = a->retry_patience;
    job.stats = a->d_stats;
    job.prio_pattern = a->prio_pattern;
    job.ck_margin = a->ck_margin > 0 ? a->ck_margin : 1000;
    job.trust_mode = a->trust_mode;
    { constexpr ColCoeffs ck = col_coeffs(); for (int i = 0; i < 16; i++) job.col_k[i] = ck.k[i]; }
    job.trips = (job.nmb + waves_ - 1) / waves_;
    job.it_step = pick_it_step(job.trips);
    job.order = a->d_order;
    const int waves = waves_;
    const size_t lds = lds_bytes(job.nmb, job.out_words, job.stg_words, waves);
    const dim3 grid((unsigned)a->grid), block((unsigned)waves * 64u);
    hipStream_t st = (hipStream_t)a->stream;
#define PSX_LAUNCH(CODEC)                                                                                             \
    do {                                                                                                              \
        if (job.stats) {                                                                                              \
            if (a->large) hipLaunchKernelGGL((mdec_encode_frames_kernel<CODEC, kWavesLarge, kOccLarge, true>), grid, block, lds, st, job); \
            else hipLaunchKernelGGL((mdec_encode_frames_kernel<CODEC, kWavesSmall, kOccSmall, true>), grid, block, lds, st, job);          \
        } else {                                                                                                      \
            if (a->large) hipLaunchKernelGGL((mdec_encode_frames_kernel<CODEC, kWavesLarge, kOccLarge, false>), grid, block, lds, st, job); \
            else hipLaunchKernelGGL((mdec_encode_frames_kernel<CODEC, kWavesSmall, kOccSmall, false>), grid, block, lds, st, job);          \
        }                                                                                                             \
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
        e = hipFuncSetAttribute((const void*)mdec_encode_frames_kernel<CODEC, kWavesSmall, kOccSmall, false>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                                        \
        if (e == hipSuccess)                                                                                                    \
            e = hipFuncSetAttribute((const void*)mdec_encode_frames_kernel<CODEC, kWavesLarge, kOccLarge, false>,                \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                                    \
        if (e == hipSuccess)                                                                                                    \
            e = hipFuncSetAttribute((const void*)mdec_encode_frames_kernel<CODEC, kWavesSmall, kOccSmall, true>,                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                                    \
        if (e == hipSuccess)                                                                                                    \
            e = hipFuncSetAttribute((const void*)mdec_encode_frames_kernel<CODEC, kWavesLarge, kOccLarge, true>,                 \
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
