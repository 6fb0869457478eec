// psxhip_api.cpp -- the C-ABI layer of libpsxav_hip.so (include/psxav_hip.h): contexts, stream/
// buffer plumbing and kernel launches.  No algorithmic work happens on the host here, and there is
// no CPU fallback: without a gfx950 device every entry point fails with PSXHIP_EDEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <ctime>
#include <mutex>
#include <thread>
#include <vector>

#include "psxhip_internal.h"

namespace {

thread_local char g_err[512] = "";
std::mutex g_tables_mu;
// Split launches (one frame across many workgroups: a frame's groups wait for each other, so all of them have to be resident) are
// ordered one behind the other per device, whatever context, thread or stream they come from: a single launch always fits the
// device (the geometry sees to that), but the groups of three or four launches dispatched side by side could fill every CU slot with
// groups that each wait for siblings still in the queue -- nobody would finish until the watchdog lets go.  Asynchronous launches
// chain on an event; a one-frame call holds the gate until its flag is up.  (Processes that share a GPU are not ordered against
// each other: the watchdog is what is left there -- a released frame is flagged, counted, and the one-frame call takes it again
// through the frame kernel.)
struct SplitGate {
    std::mutex mu;
    hipEvent_t last = nullptr;
    hipStream_t last_stream = nullptr;      // the stream the last asynchronous split launch went to (launches on ONE stream are ordered as they are)
    bool pending = false;
};
SplitGate g_split_gate[64];
bool g_tables_ready[64] = {false};

#define HIP_TRY(expr, code)                                                                   \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            psxhip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (code);                                                                    \
        }                                                                                     \
    } while (0)

int ensure_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        psxhip_set_error("no HIP device visible (libpsxav_hip has no CPU fallback)");
        return PSXHIP_EDEVICE;
    }
    if (device < 0 || device >= n || device >= 64) {
        psxhip_set_error("device %d out of range (%d visible)", device, n);
        return PSXHIP_EINVAL;
    }
    HIP_TRY(hipSetDevice(device), PSXHIP_EDEVICE);
    std::lock_guard<std::mutex> lk(g_tables_mu);
    if (!g_tables_ready[device]) {
        HIP_TRY(psxhip_mdec_upload_tables(), PSXHIP_EDEVICE);
        g_tables_ready[device] = true;
    }
    return PSXHIP_OK;
}

}  // namespace

extern "C" void psxhip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* psxhip_last_error(void) { return g_err; }
extern "C" const char* psxhip_version(void) { return "psxav_hip 0.4 (gfx950, " PSXHIP_MDEC_KERNEL_REV ")"; }

extern "C" int psxhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int psxhip_ensure_device(int device) { return ensure_device(device); }

// ------------------------------------------------------------------------------------------ MDEC

constexpr int kLanes = 2;
struct psxhip_mdec_ctx {
    int device, codec, width, height, nmb;
    int max_frame_size, out_words, stg_words;
    int groups_max;            // persistent grid size: compute units x resident groups per CU
    int large;                 // 1: one 16-wavefront group per CU (two 12-wavefront groups do not fit the LDS)
    size_t lds_bytes;
    unsigned int* d_ticket;         // [kLanes][128] per launch lane: frame hand-out counters, retry queue pop tickets / state (a cache line each) (the kernel re-arms them when it ends); word [2] of lane 0 is the hint all lanes share
    unsigned int* d_retry;          // [kLanes][retry_cap] retry queue slots (frames a group hands on instead of running another pass over them)
    int retry_cap;
    // launch lanes (psxhip_mdec_set_lanes): launches that own different counters may overlap
    int lanes, next_lane;
    hipStream_t lane_stream[2];
    hipEvent_t lane_in[2], lane_done[2];
    bool lane_pending[2];           // lane_done[l] has been recorded and no caller stream has been ordered behind it yet
    int retry_patience;
    int trust_mode;                 // experiments (PSXHIP_MDEC_TRUST): 1 = foreign hints always trusted, 2 = never
    bool spare_groups;              // PSXHIP_MDEC_SPARE (experiments, measured and not used)
    int max_run;                    // longest run of consecutive frames a frame ticket may be (4; PSXHIP_MDEC_RUN: experiments)
    unsigned long long* d_stats;    // diagnostics (PSXHIP_MDEC_STATS=1)
    unsigned prio_pattern;
    int ck_margin;
    uint32_t* d_order;              // the order a pass's tickets visit the macroblocks (psxhip_mdec_pass_order)
    uint32_t* d_order_large;        // ... for the 16-wavefront shape, when small batches may use it (see encode_frames_device)
    int n_cu;
    // host-path staging: two chunk-sized sets of device buffers and pinned host buffers (double buffering)
    hipStream_t stream;
    uint8_t* d_frames[2];
    uint8_t* d_out[2];
    psxhip_mdec_result_t* d_res[2];
    int32_t* d_sizes[2];
    uint8_t* h_in[2];                 // pinned
    uint8_t* h_out[2];                // pinned: output rows, then the chunk's results
    hipEvent_t chunk_done[2];
    hipStream_t stream2;             // chunks alternate between the two streams (and the two launch lanes): copies and kernels of neighbouring chunks overlap
    int cap_frames;                   // frames per chunk the buffers hold
    size_t cap_out_stride;
    // the one-frame-per-call pattern (encode_frame_bs): one page-locked, device-visible block [frame | output row | result] and a
    // frame-sized device buffer; a call is a CPU copy in, two launches (stage-in, encode -- which writes the host block itself), a
    // wait, a CPU copy out
    uint8_t* h_call;                  // host address
    uint8_t* d_call;                  // the same block as the device addresses it
    uint8_t* d_call_frame;
    size_t call_out_off, call_res_off;
    bool call_disabled;
    bool call_no_flush;               // PSXHIP_NO_HDP_FLUSH (measurements only: what the flush costs)
    volatile uint32_t* hdp_flush;     // the device's HDP_MEM_FLUSH_CNTL register: CPU writes through the BAR are in the device's memory after a write and a read-back of it
    bool call_bar;                    // d_call_frame is device memory the CPU writes straight into (large BAR): no stage-in launch
    int call_hint;                    // the previous one-frame call's answer: where the split kernel builds its streams before it knows the answer
    unsigned call_seq;                // the split kernel's last group stores it into the host block when row and result are there
    // launches of a few frames: one frame across many workgroups (mdec_split.inc)
    unsigned char* d_split_ws;        // [kLanes][split_max] per-frame workspaces, all zero between launches
    size_t split_ws_stride;
    int split_max;                    // launches of at most this many frames take the split kernel (0: never)
    unsigned long long* d_split_dbg;  // diagnostics (PSXHIP_MDEC_SPLIT_DBG=1): phase stamps of the last split launch
    int split_dbg_groups;
    // diagnostics (PSXHIP_PERCALL_TRACE=1): where a one-frame call's host time goes, printed when the context is destroyed
    bool call_trace;
    double call_ns[6];                // copy in, stage-in launch, encode launch, wait, copy out, calls
};

namespace {
// LDS working set of one frame for a geometry; *large = 1 when only the one-group-per-CU shape fits
int mdec_geometry(int width, int height, int max_frame_size, size_t lds_cu, int* large, int* out_words, int* stg_words,
                  size_t* lds_bytes) {
    const int nmb = (width / 16) * (height / 16);
    const int image = (max_frame_size + 3) / 4;     // dwords of the frame image
    const int sw = image + nmb + 2;
    if (sw > 0xFFFF) return 0;                      // staging offsets are 16-bit
    // The frame image is assembled in LDS whole, or one tile of 8 / 4 / 2 KiB at a time (+2: tile slack; at most 16 tiles).
    // Whole is a little faster (one merge sweep); a smaller tile is taken when that is what lets two 12-wavefront groups
    // share a CU (17-21 % faster than one 16-wavefront group: 640x480 at 8 KiB budgets needs the 4 KiB tile for it), or
    // what makes the geometry fit at all.
    const int tiles[4] = {image, 2048, 1024, 512};
    int lg = 1, ow = 0;
    for (int shape = 0; shape < 2 && !ow; shape++) {             // 0: two small groups per CU, 1: one large group
        for (int i = 0; i < 4 && !ow; i++) {
            const int t = tiles[i] < image ? tiles[i] : image;
            if ((image + t - 1) / t > 16) continue;
            if ((shape ? 1 : 2) * psxhip_mdec_lds_bytes(nmb, t + 2, sw, shape) <= lds_cu) {
                lg = shape;
                ow = t + 2;
            }
        }
    }
    if (const char* e = getenv("PSXHIP_MDEC_LARGE")) {           // experiments: force the large shape
        if (atoi(e) && !lg) {
            lg = 1;
            ow = 0;
            for (int i = 0; i < 4 && !ow; i++) {
                const int t = tiles[i] < image ? tiles[i] : image;
                if ((image + t - 1) / t <= 16 && psxhip_mdec_lds_bytes(nmb, t + 2, sw, 1) <= lds_cu) ow = t + 2;
            }
        }
    }
    if (!ow) {                                                   // nothing fits: report what the smallest working set would need
        lg = 1;
        ow = (image < 512 ? image : 512) + 2;
    }
    const size_t need = psxhip_mdec_lds_bytes(nmb, ow, sw, lg);
    if (large) *large = lg;
    if (out_words) *out_words = ow;
    if (stg_words) *stg_words = sw;
    if (lds_bytes) *lds_bytes = need;
    return need <= lds_cu;
}
bool mdec_args_ok(int codec, int width, int height, int max_frame_size) {
    return !(codec < 0 || codec > 2 || width <= 0 || height <= 0 || (width % 16) || (height % 16) || width > 1024 ||
             height > 1024 || max_frame_size < 8);
}
}  // namespace

extern "C" int psxhip_mdec_query_geometry(int device, int codec, int width, int height, int max_frame_size,
                                          psxhip_mdec_geometry_t* out) {
    if (out) memset(out, 0, sizeof(*out));
    if (!mdec_args_ok(codec, width, height, max_frame_size)) {
        psxhip_set_error("bad MDEC geometry: codec %d, %dx%d, budget %d", codec, width, height, max_frame_size);
        return PSXHIP_EINVAL;
    }
    int rc = ensure_device(device);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device), PSXHIP_EDEVICE);
    const size_t lds_cu = (size_t)prop.maxSharedMemoryPerMultiProcessor;
    int large = 0, ow = 0, sw = 0;
    size_t need = 0;
    const int fits = mdec_geometry(width, height, max_frame_size, lds_cu, &large, &ow, &sw, &need);
    if (out) {
        out->fits = fits;
        out->groups_per_cu = fits ? (large ? 1 : 2) : 0;
        out->wavefronts_per_group = psxhip_mdec_threads_per_group(large) / 64;
        out->lds_bytes_per_group = (int64_t)need;
        out->lds_bytes_per_cu = (int64_t)lds_cu;
        out->frames_in_flight = fits ? prop.multiProcessorCount * (large ? 1 : 2) : 0;
        out->image_tile_bytes = fits ? (ow - 2) * 4 : 0;
        // largest budget that still fits this frame size (the LDS need grows by 8 bytes per budget dword)
        int lo = 8, hi = 1 << 20;
        while (lo < hi) {
            const int mid = lo + (hi - lo + 1) / 2;
            if (mdec_geometry(width, height, mid, lds_cu, nullptr, nullptr, nullptr, nullptr)) lo = mid;
            else hi = mid - 1;
        }
        out->max_frame_size_limit = mdec_geometry(width, height, lo, lds_cu, nullptr, nullptr, nullptr, nullptr) ? lo : 0;
    }
    return PSXHIP_OK;
}

extern "C" int psxhip_mdec_create(psxhip_mdec_ctx_t** out, int device, int codec, int width, int height,
                                  int max_frame_size) {
    if (!out) return PSXHIP_EINVAL;
    *out = nullptr;
    if (!mdec_args_ok(codec, width, height, max_frame_size)) {
        psxhip_set_error("bad MDEC geometry: codec %d, %dx%d, budget %d", codec, width, height, max_frame_size);
        return PSXHIP_EINVAL;
    }
    int rc = ensure_device(device);
    if (rc) return rc;

    psxhip_mdec_ctx* c = (psxhip_mdec_ctx*)calloc(1, sizeof(*c));
    if (!c) return PSXHIP_ENOMEM;
    struct Guard {                       // releases the half-built context on every early return
        psxhip_mdec_ctx* p;
        ~Guard() { if (p) psxhip_mdec_destroy(p); }
    } guard{c};
    c->device = device;
    c->codec = codec;
    c->width = width;
    c->height = height;
    c->nmb = (width / 16) * (height / 16);
    c->max_frame_size = max_frame_size;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device), PSXHIP_EDEVICE);
    const size_t lds_cu = (size_t)prop.maxSharedMemoryPerMultiProcessor;
    // shape: two 12-wavefront groups per CU when their LDS fits, else one 16-wavefront group
    if (!mdec_geometry(width, height, max_frame_size, lds_cu, &c->large, &c->out_words, &c->stg_words, &c->lds_bytes)) {
        psxhip_set_error("frame budget %d with %d macroblocks needs %zu B of LDS (> %zu); see psxhip_mdec_query_geometry",
                         max_frame_size, c->nmb, c->lds_bytes, lds_cu);
        return PSXHIP_EINVAL;
    }
    // opt the kernels into the whole LDS once (contexts with different geometries share the kernel attribute)
    HIP_TRY(psxhip_mdec_set_max_lds(codec, lds_cu), PSXHIP_EDEVICE);
    c->groups_max = prop.multiProcessorCount * (c->large ? 1 : 2);
    if (const char* e = getenv("PSXHIP_MDEC_GRID")) { const int g = atoi(e); if (g > 0 && g < c->groups_max) c->groups_max = g; }   // experiments
    // the kernel's leave word packs four 16-bit sums, to each of which a group adds at most 63 (kTrustCap): the grid stays below
    // 65535 / 63 groups so that none carries into its neighbour (1040; MI355X: 512)
    if (c->groups_max > 1040) c->groups_max = 1040;
    c->spare_groups = getenv("PSXHIP_MDEC_SPARE") != nullptr;      // experiments; read once
    c->prio_pattern = 0x2EE01u;      // younger group raised 6 steps in 8, older 1 (re-swept on mdec-k2.23: tools/gpu_prio_sweep.py)
    if (const char* e = getenv("PSXHIP_MDEC_PRIO")) c->prio_pattern = (unsigned)strtoul(e, nullptr, 0);
    if (const char* e = getenv("PSXHIP_MDEC_CKMARGIN")) c->ck_margin = atoi(e);      // experiments (tools/gpu_ckmargin_sweep.py)
    // Runs of consecutive frames per ticket are built and measured, and NOT the default: what made
    // scene-structured content fast is the trust policy (pilot when foreign hints fail); on top of it runs of 2 / 4 changed mixed
    // content by -2 .. +5 % and cost uniform content 10 % with two launch lanes (groups that finish while their launch's other
    // groups have yet to start may not wait for handed-on frames).  PSXHIP_MDEC_RUN=2 / 4 turns them on.
    c->max_run = 1;
    if (const char* e = getenv("PSXHIP_MDEC_TRUST")) c->trust_mode = atoi(e);
    if (const char* e = getenv("PSXHIP_MDEC_RUN")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) c->max_run = v; }      // experiments: 1 = single-frame tickets

    c->lanes = 1;
    HIP_TRY(hipMalloc((void**)&c->d_ticket, kLanes * 128 * sizeof(unsigned int)), PSXHIP_ENOMEM);
    HIP_TRY(hipMemset(c->d_ticket, 0, kLanes * 128 * sizeof(unsigned int)), PSXHIP_EDEVICE);
    if (!getenv("PSXHIP_MDEC_NO_RETRY_QUEUE")) {      // experiments: frames are never handed on
        c->retry_cap = 1 << 16;
        c->retry_patience = 256;
        if (const char* e = getenv("PSXHIP_MDEC_QUEUE_PATIENCE")) c->retry_patience = atoi(e);      // tests: 0 exercises the give-up path
        HIP_TRY(hipMalloc((void**)&c->d_retry, (size_t)kLanes * c->retry_cap * sizeof(unsigned int)), PSXHIP_ENOMEM);
        HIP_TRY(hipMemset(c->d_retry, 0xFF, (size_t)kLanes * c->retry_cap * sizeof(unsigned int)), PSXHIP_EDEVICE);
    }
    // (the host path's two streams are created when it is first used: a context that only ever launches on its caller's streams
    //  would otherwise take two of the device's few hardware queues -- streams are dealt onto them round-robin -- and with three
    //  contexts alive two callers' streams shared one queue: their launches ran one after the other)
    c->n_cu = prop.multiProcessorCount;
    c->call_trace = getenv("PSXHIP_PERCALL_TRACE") != nullptr;
    c->call_no_flush = getenv("PSXHIP_NO_HDP_FLUSH") != nullptr;
    {
        // one frame across many workgroups, for launches of at most split_max frames (PSXHIP_MDEC_SPLIT_MAX: experiments, 0 = off).
        // 12: tools/gpu_r06_split_sweep.py -- 320x240: 14.7 us for one frame against the frame kernel's 40, 30 against 41 at 12 frames,
        // 54 against 41 at 16; 640x480 v3: 34 against 171 for one, 152 against 172 at 12, 216 against 172 at 16
        c->split_max = 12;
        if (const char* e = getenv("PSXHIP_MDEC_SPLIT_MAX")) c->split_max = atoi(e);
        if (c->split_max > 64) c->split_max = 64;
        psxhip_mdec_split_geo_t g;
        if (c->split_max > 0 && psxhip_mdec_split_geometry(codec, width, height, max_frame_size, 1, c->n_cu, &g)) {
            c->split_ws_stride = g.ws_stride;
            const size_t bytes = (size_t)(kLanes + 1) * c->split_max * g.ws_stride;
            HIP_TRY(hipMalloc((void**)&c->d_split_ws, bytes), PSXHIP_ENOMEM);
            HIP_TRY(hipMemset(c->d_split_ws, 0, bytes), PSXHIP_EDEVICE);
            if (getenv("PSXHIP_MDEC_SPLIT_DBG")) HIP_TRY(hipMalloc((void**)&c->d_split_dbg, (size_t)c->split_max * c->n_cu * 8 * sizeof(unsigned long long)), PSXHIP_ENOMEM);
        } else {
            c->split_max = 0;
        }
    }
    // A batch of at most one frame per CU gains nothing from the two-group shape (its point is two frames per CU): such
    // launches use the 16-wavefront shape, which finishes a lone frame sooner -- the drop-in one-frame-per-call pattern most
    // of all.  Needs that shape's pass order too, and its (larger) LDS working set to fit.
    const bool both = !c->large && psxhip_mdec_lds_bytes(c->nmb, c->out_words, c->stg_words, 1) <= lds_cu && !getenv("PSXHIP_MDEC_NO_SMALL_BATCH_SHAPE");
    for (int shape = c->large; shape <= (both ? 1 : c->large); shape++) {
        const int n = psxhip_mdec_pass_table(width, height, shape, nullptr, 0);
        const size_t tab_bytes = ((size_t)n + 1) * 2 * sizeof(uint32_t);          // n tickets + the all-zero entry behind them
        uint32_t* h = (uint32_t*)malloc(tab_bytes);
        if (!h) { psxhip_set_error("out of host memory"); return PSXHIP_ENOMEM; }
        if (psxhip_mdec_pass_table(width, height, shape, h, n + 1) != n) { free(h); psxhip_set_error("out of host memory"); return PSXHIP_ENOMEM; }
        uint32_t** dst = shape == c->large ? &c->d_order : &c->d_order_large;
        hipError_t e = hipMalloc((void**)dst, tab_bytes);
        if (e == hipSuccess) e = hipMemcpy(*dst, h, tab_bytes, hipMemcpyHostToDevice);
        free(h);
        if (e != hipSuccess) { psxhip_set_error("pass order table: %s", hipGetErrorString(e)); return PSXHIP_ENOMEM; }
    }
    if (const char* e = getenv("PSXHIP_MDEC_STATS")) {
        if (atoi(e)) {
            HIP_TRY(hipMalloc((void**)&c->d_stats, PSXHIP_MDEC_STATS_TOTAL * sizeof(unsigned long long)), PSXHIP_ENOMEM);
            HIP_TRY(hipMemset(c->d_stats, 0, PSXHIP_MDEC_STATS_TOTAL * sizeof(unsigned long long)), PSXHIP_EDEVICE);
        }
    }
    // the fills above ran on the null stream; the context's launches go to streams that do not wait for it
    HIP_TRY(hipStreamSynchronize(nullptr), PSXHIP_EDEVICE);
    guard.p = nullptr;                   // ownership passes to the caller
    *out = c;
    return PSXHIP_OK;
}

static void psxhip_mdec_free_staging(psxhip_mdec_ctx* c);

extern "C" void psxhip_mdec_destroy(psxhip_mdec_ctx_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    {
        SplitGate& gate = g_split_gate[c->device & 63];          // (the gate must not keep a stream that is about to go)
        std::lock_guard<std::mutex> lk(gate.mu);
        if (gate.last_stream && (gate.last_stream == c->stream || gate.last_stream == c->stream2 || gate.last_stream == c->lane_stream[0] ||
                                 gate.last_stream == c->lane_stream[1])) {
            (void)hipStreamSynchronize(gate.last_stream);
            gate.last_stream = nullptr;
            gate.pending = false;
        }
    }
    for (int l = 0; l < kLanes; l++) {
        if (c->lane_stream[l]) { (void)hipStreamSynchronize(c->lane_stream[l]); (void)hipStreamDestroy(c->lane_stream[l]); }
        if (c->lane_in[l]) (void)hipEventDestroy(c->lane_in[l]);
        if (c->lane_done[l]) (void)hipEventDestroy(c->lane_done[l]);
    }
    if (c->d_ticket) (void)hipFree(c->d_ticket);
    if (c->d_retry) (void)hipFree(c->d_retry);
    if (c->d_order) (void)hipFree(c->d_order);
    if (c->d_order_large) (void)hipFree(c->d_order_large);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->call_trace && c->call_ns[5] > 0)
        fprintf(stderr, "per-call trace (%.0f calls, us per call): copy in %.2f, stage-in launch %.2f, encode launch %.2f, wait %.2f, copy out %.2f\n", c->call_ns[5],
                c->call_ns[0] / c->call_ns[5] * 1e-3, c->call_ns[1] / c->call_ns[5] * 1e-3, c->call_ns[2] / c->call_ns[5] * 1e-3, c->call_ns[3] / c->call_ns[5] * 1e-3, c->call_ns[4] / c->call_ns[5] * 1e-3);
    if (c->d_split_ws) (void)hipFree(c->d_split_ws);
    if (c->d_split_dbg) (void)hipFree(c->d_split_dbg);
    psxhip_mdec_free_staging(c);
    if (c->h_call) (void)hipHostFree(c->h_call);
    if (c->d_call_frame) (void)hipFree(c->d_call_frame);
    for (int b = 0; b < 2; b++)
        if (c->chunk_done[b]) (void)hipEventDestroy(c->chunk_done[b]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    free(c);
}

// One launch over the frames of `nb` batches (1 .. PSXHIP_MDEC_MAX_BATCHES, all vetted by the caller) on `stream`, with the
// hand-out counters and the retry queue of launch lane `lane`.  Launches of one lane must be ordered (same stream, or events).
static int mdec_launch_lane(psxhip_mdec_ctx* c, int lane, const psxhip_mdec_batch_t* batches, int nb, size_t frame_stride,
                            int uniform_max_size, size_t out_stride, hipStream_t stream, unsigned* d_done_flag = nullptr,
                            unsigned done_seq = 0, bool* flagged = nullptr, std::unique_lock<std::mutex>* gate_hold = nullptr, bool no_split = false) {
    int n_frames = 0;
    for (int i = 0; i < nb; i++) n_frames += batches[i].n_frames;
    // A launch of a few frames: every frame across many workgroups (the reference's own pattern, one frame per call, most of all:
    // one workgroup would encode it on ONE compute unit while 255 idle)
    if (nb == 1 && n_frames <= c->split_max && !c->d_stats && !no_split) {
        psxhip_mdec_split_t sp;
        memset(&sp, 0, sizeof sp);
        if (psxhip_mdec_split_geometry(c->codec, c->width, c->height, c->max_frame_size, n_frames, c->n_cu, &sp.geo)) {
            sp.d_frames = batches[0].d_frames; sp.d_out = batches[0].d_out; sp.d_results = batches[0].d_results;
            sp.d_frame_max_sizes = batches[0].d_frame_max_sizes;
            sp.frame_stride = frame_stride; sp.out_stride = out_stride;
            sp.width = c->width; sp.height = c->height; sp.codec = c->codec; sp.n_frames = n_frames;
            sp.uniform_max_size = uniform_max_size; sp.max_frame_size = c->max_frame_size;
            // (a one-frame call returns when its flag is up, while the kernel still tidies its workspace: it has one of its own, so that
            //  a launch on a caller's stream right after cannot find lane 0's in use)
            sp.d_ws = c->d_split_ws + (size_t)(d_done_flag ? kLanes : lane) * c->split_max * c->split_ws_stride;
            sp.d_done_flag = d_done_flag; sp.done_seq = done_seq;
            sp.hint = d_done_flag ? c->call_hint : 0;
            if (flagged) *flagged = d_done_flag != nullptr;
            sp.d_lost = c->d_ticket + 128 * lane + 3;
            sp.d_dbg = c->d_split_dbg;
            c->split_dbg_groups = sp.geo.segs * n_frames;
            sp.stream = stream;
            static const bool no_gate = getenv("PSXHIP_NO_SPLIT_GATE") != nullptr;      // experiments only: what happens without the gate (tests/test_gpu_split_threads.py)
            if (no_gate) {
                HIP_TRY(psxhip_mdec_split_launch(&sp), PSXHIP_EDEVICE);
                return PSXHIP_OK;
            }
            SplitGate& gate = g_split_gate[c->device & 63];
            std::unique_lock<std::mutex> lk(gate.mu);
            if (!gate.last) HIP_TRY(hipEventCreateWithFlags(&gate.last, hipEventDisableTiming), PSXHIP_EDEVICE);
            if (gate.pending && gate.last_stream != stream) {
                // another stream than the last split launch's: an event recorded on THAT stream now stands behind everything it was
                // given, the launch included.  (Launches that stay on one stream pay nothing.  A stream its owner has destroyed
                // meanwhile is refused by the runtime: its work is done.)
                if (hipEventRecord(gate.last, gate.last_stream) == hipSuccess) HIP_TRY(hipStreamWaitEvent(stream, gate.last, 0), PSXHIP_EDEVICE);
                else (void)hipGetLastError();
            }
            HIP_TRY(psxhip_mdec_split_launch(&sp), PSXHIP_EDEVICE);
            gate.last_stream = stream;
            gate.pending = true;
            if (d_done_flag && gate_hold) *gate_hold = std::move(lk);          // the synchronous caller lets go when its flag is up
            return PSXHIP_OK;
        }
    }
    // A batch of at most one frame per CU gains nothing from the two-group shape (its point is two frames per CU): such
    // launches use the 16-wavefront shape, which finishes a lone frame sooner -- the drop-in one-frame-per-call pattern most
    // of all.  (Tried and dropped: sending the REMAINDER of a large batch -- the frames past the last full round of
    // groups_max, when they are at most one per CU -- through that shape as a second launch.  1250 frames of 640x480 are
    // 2.44 rounds and the mean group is resident 76 % of the launch, but the frame tickets already let early finishers
    // start the third round while others are in their second; a second launch puts a barrier there instead: 1.176 ms
    // against 1.144 ms.)
    const bool small_batch = c->d_order_large && n_frames <= c->n_cu;
    psxhip_mdec_launch_t a;
    a.batches = batches;
    a.n_batches = nb;
    a.frame_stride = frame_stride;
    a.width = c->width;
    a.height = c->height;
    a.codec = c->codec;
    a.n_frames = n_frames;
    a.uniform_max_size = uniform_max_size;
    a.out_stride = out_stride;
    a.out_words = c->out_words;
    a.out_tile = c->out_words - 2;
    a.max_frame_size = c->max_frame_size;
    a.stg_words = c->stg_words;
    // frame tickets: runs of consecutive frames (a group encodes neighbours in time one after the other: the hint that is worth
    // trusting), long runs first (psxhip_mdec_ticket_plan)
    psxhip_mdec_ticket_plan(n_frames, c->groups_max, c->max_run, &a.t4, &a.t2, &a.n_tickets);
    a.grid = a.n_tickets < c->groups_max ? a.n_tickets : c->groups_max;
    a.large = c->large || small_batch;
    a.stream = stream;
    a.d_ticket = c->d_ticket + 128 * lane;
    a.d_hint = c->d_ticket + 2;
    // frames are handed on only when a group can hold more than one (else nobody is left to take them), the queue has a slot per
    // frame, and a group holds FEW: from about eight frames per group on the fresh-frame tickets level the groups by themselves,
    // and a frame restarted on another XCD is read from HBM again (10 000 x 640x480: -1 % time, +8 % traffic with the queue)
    const bool queue = c->d_retry && n_frames > a.grid && n_frames <= 8 * a.grid && n_frames < c->retry_cap;
    // (experiments, PSXHIP_MDEC_SPARE: a launch of at most one run per group has CU slots to spare when runs left some empty -- 1000 frames:
    //  500 runs on 512 slots -- and the kernel lets groups start without a ticket, to take handed-on frames only.  Measured: they
    //  never get any, because a group that finds not every group of its launch started may not wait and gives its place up at
    //  once; not used.)
    if (queue && a.n_tickets < c->groups_max && n_frames > a.n_tickets && c->spare_groups) a.grid = c->groups_max;
    a.d_retry = queue ? c->d_retry + (size_t)lane * c->retry_cap : nullptr;
    a.retry_cap = queue ? c->retry_cap : 0;
    a.retry_patience = c->retry_patience;
    a.d_order = small_batch ? c->d_order_large : c->d_order;
    a.d_stats = c->d_stats;
    a.prio_pattern = c->prio_pattern;
    a.ck_margin = c->ck_margin;
    a.trust_mode = c->trust_mode;
    HIP_TRY(psxhip_mdec_launch(&a), PSXHIP_EDEVICE);
    return PSXHIP_OK;
}

static int mdec_ensure_lanes(psxhip_mdec_ctx* c) {
    for (int l = 0; l < kLanes; l++) {
        if (!c->lane_stream[l]) HIP_TRY(hipStreamCreateWithFlags(&c->lane_stream[l], hipStreamNonBlocking), PSXHIP_EDEVICE);
        if (!c->lane_in[l]) HIP_TRY(hipEventCreateWithFlags(&c->lane_in[l], hipEventDisableTiming), PSXHIP_EDEVICE);
        if (!c->lane_done[l]) HIP_TRY(hipEventCreateWithFlags(&c->lane_done[l], hipEventDisableTiming), PSXHIP_EDEVICE);
    }
    return PSXHIP_OK;
}

extern "C" int psxhip_mdec_fence(psxhip_mdec_ctx_t* c, void* stream) {
    if (!c) return PSXHIP_EINVAL;
    HIP_TRY(hipSetDevice(c->device), PSXHIP_EDEVICE);
    for (int l = 0; l < kLanes; l++)
        if (c->lane_pending[l]) {
            HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c->lane_done[l], 0), PSXHIP_EDEVICE);
            c->lane_pending[l] = false;
        }
    return PSXHIP_OK;
}

extern "C" int psxhip_mdec_set_lanes(psxhip_mdec_ctx_t* c, int lanes) {
    if (!c || lanes < 1 || lanes > kLanes) {
        psxhip_set_error("psxhip_mdec_set_lanes: 1 or %d lanes", kLanes);
        return PSXHIP_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device), PSXHIP_EDEVICE);
    for (int l = 0; l < kLanes; l++) {          // nothing of the old regime is left in flight
        if (c->lane_stream[l]) HIP_TRY(hipStreamSynchronize(c->lane_stream[l]), PSXHIP_EDEVICE);
        c->lane_pending[l] = false;
    }
    if (lanes > 1) { const int rc = mdec_ensure_lanes(c); if (rc) return rc; }
    c->lanes = lanes;
    c->next_lane = 0;
    return PSXHIP_OK;
}

extern "C" int psxhip_mdec_watchdog(psxhip_mdec_ctx_t* c, unsigned* lost) {
    if (!c || !lost) return PSXHIP_EINVAL;
    *lost = 0;
    HIP_TRY(hipSetDevice(c->device), PSXHIP_EDEVICE);
    HIP_TRY(hipDeviceSynchronize(), PSXHIP_EDEVICE);
    unsigned w[kLanes * 128];
    HIP_TRY(hipMemcpy(w, c->d_ticket, sizeof w, hipMemcpyDeviceToHost), PSXHIP_EDEVICE);
    for (int l = 0; l < kLanes; l++) *lost += w[128 * l + 3];
    return PSXHIP_OK;
}

extern "C" int psxhip_mdec_encode_batches_device(psxhip_mdec_ctx_t* c, const psxhip_mdec_batch_t* batches, int n_batches,
                                                 size_t frame_stride, int uniform_max_size, size_t out_stride, void* stream) {
    if (!c || !batches || n_batches < 0) {
        psxhip_set_error("encode_batches_device: NULL argument");
        return PSXHIP_EINVAL;
    }
    const size_t fsz = (size_t)c->width * c->height * 3 / 2;
    if (frame_stride < fsz || (frame_stride & 3) || (out_stride & 3)) {
        psxhip_set_error("encode_frames_device: strides / pointers must be 4-byte aligned and frame_stride >= w*h*3/2");
        return PSXHIP_EINVAL;
    }
    bool any_uniform = false;
    long long total = 0;
    for (int i = 0; i < n_batches; i++) {
        const psxhip_mdec_batch_t& b = batches[i];
        if (b.n_frames < 0 || (b.n_frames > 0 && (!b.d_frames || !b.d_out || !b.d_results))) {
            psxhip_set_error("encode_frames_device: NULL argument (batch %d)", i);
            return PSXHIP_EINVAL;
        }
        if (b.n_frames > 0 && (((uintptr_t)b.d_frames & 3) || ((uintptr_t)b.d_out & 3))) {
            psxhip_set_error("encode_frames_device: strides / pointers must be 4-byte aligned and frame_stride >= w*h*3/2");
            return PSXHIP_EINVAL;
        }
        if (b.n_frames > 0 && !b.d_frame_max_sizes) any_uniform = true;
        total += b.n_frames;
    }
    if (any_uniform && (uniform_max_size < 8 || uniform_max_size > c->max_frame_size || (size_t)uniform_max_size > out_stride)) {
        psxhip_set_error("encode_frames_device: frame_max_size %d outside [8, %d] or larger than out_stride",
                         uniform_max_size, c->max_frame_size);
        return PSXHIP_EINVAL;
    }
    if (total == 0) return PSXHIP_OK;
    if (total > 0xFFFFFF) {          // (the retry queue's entries carry 24 bits of frame index)
        psxhip_set_error("encode_frames_device: more than 16 777 215 frames in one call");
        return PSXHIP_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device), PSXHIP_EDEVICE);
    hipStream_t S = (hipStream_t)stream;
    // groups of up to PSXHIP_MDEC_MAX_BATCHES non-empty batches, one launch each
    psxhip_mdec_batch_t grp[PSXHIP_MDEC_MAX_BATCHES];
    int ng = 0;
    for (int i = 0; i <= n_batches; i++) {
        if (i < n_batches && batches[i].n_frames > 0) grp[ng++] = batches[i];
        if (ng == 0 || (ng < PSXHIP_MDEC_MAX_BATCHES && i < n_batches)) continue;
        int rc;
        if (c->lanes <= 1) {
            rc = mdec_launch_lane(c, 0, grp, ng, frame_stride, uniform_max_size, out_stride, S);
        } else {
            // launch k on lane k % lanes, on the lane's own stream: it waits for the caller's stream as it stands now (inputs), and
            // the caller's stream is ordered behind the launch BEFORE this one -- launch k itself stays outstanding, so that
            // launch k + 1 (the other lane: other counters) can start while launch k's last frames finish
            const int lane = c->next_lane, prev = (lane + kLanes - 1) % kLanes;
            c->next_lane = (lane + 1) % c->lanes;
            HIP_TRY(hipEventRecord(c->lane_in[lane], S), PSXHIP_EDEVICE);
            HIP_TRY(hipStreamWaitEvent(c->lane_stream[lane], c->lane_in[lane], 0), PSXHIP_EDEVICE);
            rc = mdec_launch_lane(c, lane, grp, ng, frame_stride, uniform_max_size, out_stride, c->lane_stream[lane]);
            if (rc == PSXHIP_OK) {
                HIP_TRY(hipEventRecord(c->lane_done[lane], c->lane_stream[lane]), PSXHIP_EDEVICE);
                c->lane_pending[lane] = true;
                if (prev != lane && c->lane_pending[prev]) {
                    HIP_TRY(hipStreamWaitEvent(S, c->lane_done[prev], 0), PSXHIP_EDEVICE);
                    c->lane_pending[prev] = false;
                }
            }
        }
        if (rc) return rc;
        ng = 0;
    }
    return PSXHIP_OK;
}

extern "C" int psxhip_mdec_encode_frames_device(psxhip_mdec_ctx_t* c, const uint8_t* d_frames, size_t frame_stride,
                                                int n_frames, const int32_t* d_frame_max_sizes,
                                                int uniform_max_size, uint8_t* d_out, size_t out_stride,
                                                psxhip_mdec_result_t* d_results, void* stream) {
    if (!c || !d_frames || !d_out || !d_results || n_frames < 0) {
        psxhip_set_error("encode_frames_device: NULL argument");
        return PSXHIP_EINVAL;
    }
    psxhip_mdec_batch_t b;
    b.d_frames = d_frames;
    b.n_frames = n_frames;
    b.reserved = 0;
    b.d_frame_max_sizes = d_frame_max_sizes;
    b.d_out = d_out;
    b.d_results = d_results;
    return psxhip_mdec_encode_batches_device(c, &b, 1, frame_stride, uniform_max_size, out_stride, stream);
}

static void psxhip_mdec_free_staging(psxhip_mdec_ctx* c) {
    for (int b = 0; b < 2; b++) {
        if (c->d_frames[b]) (void)hipFree(c->d_frames[b]);
        if (c->d_out[b]) (void)hipFree(c->d_out[b]);
        if (c->d_res[b]) (void)hipFree(c->d_res[b]);
        if (c->d_sizes[b]) (void)hipFree(c->d_sizes[b]);
        if (c->h_in[b]) (void)hipHostFree(c->h_in[b]);
        if (c->h_out[b]) (void)hipHostFree(c->h_out[b]);
        c->d_frames[b] = nullptr; c->d_out[b] = nullptr; c->d_res[b] = nullptr; c->d_sizes[b] = nullptr;
        c->h_in[b] = nullptr; c->h_out[b] = nullptr;
    }
    c->cap_frames = 0;
    c->cap_out_stride = 0;
}

namespace {
// The frame of a one-frame call in device memory.  With a large BAR the CPU writes it there itself -- 115 KB in 2.3 us of
// write-combined stores (tools/microbench/bar_probe.hip: 50 GB/s) -- where the copy kernel from the page-locked block took 6 us and
// a launch of its own.  Fine-grained device memory: what the CPU wrote is what the next kernel reads.
// The HDP flush register of a HIP device, from the HSA runtime HIP itself runs on (HSA_AMD_AGENT_INFO_HDP_FLUSH; the agent is matched
// by PCI bus / device / function).  The symbols are looked up in the running process: no link-time dependency.  NULL: not available.
volatile uint32_t* hdp_flush_register(int device) {
    typedef int (*iterate_fn)(int (*)(uint64_t, void*), void*);
    typedef int (*info_fn)(uint64_t, int, void*);
    static iterate_fn iterate = (iterate_fn)dlsym(RTLD_DEFAULT, "hsa_iterate_agents");
    static info_fn info = (info_fn)dlsym(RTLD_DEFAULT, "hsa_agent_get_info");
    if (!iterate || !info) return nullptr;
    int dom = 0, bus = 0, dev = 0, fn = 0;
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, sizeof id, device) != hipSuccess || sscanf(id, "%x:%x:%x.%x", &dom, &bus, &dev, &fn) != 4) return nullptr;
    struct Ctx { info_fn info; uint32_t bdf, dom; volatile uint32_t* reg; } ctx = {info, (uint32_t)(bus << 8 | dev << 3 | fn), (uint32_t)dom, nullptr};
    iterate([](uint64_t agent, void* p) -> int {
        Ctx* x = (Ctx*)p;
        int type = 0;
        uint32_t bdf = 0, dom = 0;
        struct { uint32_t* mem; uint32_t* reg; } hdp = {nullptr, nullptr};
        if (x->info(agent, 17 /* HSA_AGENT_INFO_DEVICE */, &type) != 0 || type != 1 /* GPU */) return 0;
        if (x->info(agent, 0xA006 /* HSA_AMD_AGENT_INFO_BDFID */, &bdf) != 0 || bdf != x->bdf) return 0;
        if (x->info(agent, 0xA00F /* HSA_AMD_AGENT_INFO_DOMAIN */, &dom) == 0 && dom != x->dom) return 0;
        if (x->info(agent, 0xA00E /* HSA_AMD_AGENT_INFO_HDP_FLUSH */, &hdp) == 0) x->reg = hdp.mem;
        return 0;
    }, &ctx);
    return ctx.reg;
}
hipError_t call_frame_alloc(psxhip_mdec_ctx* c, size_t bytes) {
    int large_bar = 0;
    if (!getenv("PSXHIP_NO_BAR_WRITE") && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device) == hipSuccess && large_bar &&
        (c->hdp_flush = hdp_flush_register(c->device)) != nullptr &&
        hipExtMallocWithFlags((void**)&c->d_call_frame, bytes, hipDeviceMallocFinegrained) == hipSuccess) {
        c->call_bar = true;
        return hipSuccess;
    }
    (void)hipGetLastError();
    c->call_bar = false;
    return hipMalloc((void**)&c->d_call_frame, bytes);
}
// copy `bytes` with a few threads: one core moves ~10 GB/s, a pinned staging buffer can take several times that
// is this host pointer page-locked memory the runtime knows about (hipHostMalloc / hipHostRegister)?
bool host_pinned(const void* p) {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();      // pageable memory is reported as an error by some runtimes: not one of ours
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

void parallel_copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
    static const unsigned hw = std::thread::hardware_concurrency();
    unsigned t = bytes >= (8u << 20) ? (hw >= 16 ? 8u : (hw >= 4 ? hw / 2 : 1u)) : 1u;
    if (t <= 1) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t per = ((bytes + t - 1) / t + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (size_t o = per; o < bytes; o += per) th.emplace_back([=]() { memcpy(dst + o, src + o, (o + per <= bytes ? per : bytes - o)); });
    memcpy(dst, src, per <= bytes ? per : bytes);
    for (auto& x : th) x.join();
}
}  // namespace

// Host buffers in, host buffers out.  The batch moves through in chunks on the context's stream: while the GPU works on
// chunk k (H2D from a pinned buffer, kernel, D2H into a pinned buffer) the CPU stages chunk k+1 into the other pinned
// buffer and hands chunk k-1's output to the caller -- pageable caller memory never waits on the DMA engine and the DMA
// engine never reads pageable memory.
extern "C" int psxhip_mdec_encode_frames_host(psxhip_mdec_ctx_t* c, const uint8_t* frames, int n_frames,
                                              const int32_t* frame_max_sizes, int uniform_max_size, uint8_t* out,
                                              size_t out_stride, psxhip_mdec_result_t* results) {
    return psxhip_mdec_encode_frames_host_rows(c, frames, n_frames, frame_max_sizes, uniform_max_size, out, out_stride, results, 0);
}

// row_bytes > 0: the bytes written per output row (>= every budget of the batch) -- a caller that splits one batch with
// per-frame budgets over several calls (psxhip_multi.cpp) passes the whole batch's largest budget, so that every row is
// written exactly as wide as the unsplit call would have written it
extern "C" int psxhip_mdec_encode_frames_host_rows(psxhip_mdec_ctx_t* c, const uint8_t* frames, int n_frames,
                                                   const int32_t* frame_max_sizes, int uniform_max_size, uint8_t* out,
                                                   size_t out_stride, psxhip_mdec_result_t* results, int row_bytes) {
    if (!c || !frames || !out || !results || n_frames < 0) {
        psxhip_set_error("encode_frames_host: NULL argument");
        return PSXHIP_EINVAL;
    }
    if (n_frames == 0) return PSXHIP_OK;
    HIP_TRY(hipSetDevice(c->device), PSXHIP_EDEVICE);
    // the host path uses both launch lanes on its own streams: whatever the caller's device-path launches left running on the
    // lanes' streams is waited for first -- a lane's launches must be ordered.  (lane_pending says "no caller stream has been
    // ordered behind this launch yet", not "still running": after psxhip_mdec_fence, or after the next device call took the
    // dependency over, the launch may still be in flight with the flag down -- so every lane stream that exists is drained.)
    for (int l = 0; l < kLanes; l++)
        if (c->lane_stream[l]) {
            HIP_TRY(hipStreamSynchronize(c->lane_stream[l]), PSXHIP_EDEVICE);
            c->lane_pending[l] = false;
        }
    const size_t fsz = (size_t)c->width * c->height * 3 / 2;
    int max_size = uniform_max_size;
    if (frame_max_sizes) {
        max_size = 0;
        for (int i = 0; i < n_frames; i++) {
            if (frame_max_sizes[i] < 8 || frame_max_sizes[i] > c->max_frame_size) {
                psxhip_set_error("encode_frames_host: frame %d budget %d outside [8, %d]", i, frame_max_sizes[i],
                                 c->max_frame_size);
                return PSXHIP_EINVAL;
            }
            if (frame_max_sizes[i] > max_size) max_size = frame_max_sizes[i];
        }
    } else if (uniform_max_size < 8 || uniform_max_size > c->max_frame_size) {
        psxhip_set_error("encode_frames_host: frame_max_size %d outside [8, %d]", uniform_max_size, c->max_frame_size);
        return PSXHIP_EINVAL;
    }
    if (row_bytes > 0) {
        if (row_bytes < max_size || row_bytes > c->max_frame_size) {
            psxhip_set_error("encode_frames_host: row width %d outside [%d, %d]", row_bytes, max_size, c->max_frame_size);
            return PSXHIP_EINVAL;
        }
        max_size = row_bytes;
    }
    if ((size_t)max_size > out_stride) {
        psxhip_set_error("encode_frames_host: out_stride %zu smaller than the largest budget %d", out_stride, max_size);
        return PSXHIP_EINVAL;
    }
    const size_t dstride = ((size_t)max_size + 3) & ~(size_t)3;
    if (n_frames == 1 && !c->call_disabled) {
        // ---- one frame per call, the reference's own pattern (filefmt.c:643): see the context's h_call
        if (!c->h_call) {
            const size_t fpad = (fsz + 15) & ~(size_t)15, opad = ((size_t)c->max_frame_size + 15) & ~(size_t)15;
            void* dp = nullptr;
            if (getenv("PSXHIP_NO_PERCALL_PATH") ||
                hipHostMalloc((void**)&c->h_call, fpad + opad + 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess ||
                hipHostGetDevicePointer(&dp, c->h_call, 0) != hipSuccess || call_frame_alloc(c, fpad) != hipSuccess) {
                (void)hipGetLastError();
                if (c->h_call) (void)hipHostFree(c->h_call);
                c->h_call = nullptr;
                c->call_disabled = true;
            } else {
                c->d_call = (uint8_t*)dp;
                c->call_out_off = fpad;
                c->call_res_off = fpad + opad;
            }
        }
        if (c->h_call) {
            if (!c->stream) HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), PSXHIP_EDEVICE);
            const size_t fpad = c->call_out_off;
            struct timespec ts[6];
            auto tick = [&](int k) { if (c->call_trace) clock_gettime(CLOCK_MONOTONIC, &ts[k]); };
            tick(0);
            if (c->call_bar) {
                // write-combined stores through the BAR, then the HDP flush that puts them in the device's memory: a register write,
                // posted like the stores in front of it and like the doorbell behind it -- the device takes them in that order
                memcpy(c->d_call_frame, frames, fsz);
                __builtin_ia32_sfence();
                if (!c->call_no_flush) {
                    *c->hdp_flush = 1u;
                    __builtin_ia32_sfence();
                }
                tick(1);
            } else {
                memcpy(c->h_call, frames, fsz);
                tick(1);
                HIP_TRY(psxhip_mdec_stage_in_launch(c->d_call, c->d_call_frame, fpad, c->stream), PSXHIP_EDEVICE);
            }
            tick(2);
            psxhip_mdec_batch_t bd;
            bd.d_frames = c->d_call_frame; bd.n_frames = 1; bd.reserved = 0; bd.d_frame_max_sizes = nullptr;
            bd.d_out = c->d_call + c->call_out_off; bd.d_results = (psxhip_mdec_result_t*)(c->d_call + c->call_res_off);
            const int one = frame_max_sizes ? frame_max_sizes[0] : uniform_max_size;
            if (dstride > (size_t)one) memset(c->h_call + c->call_out_off + one, 0, dstride - (size_t)one);   // (a row wider than the frame's own budget reads as zero there)
            bool flagged = false;
            volatile unsigned* h_flag = (volatile unsigned*)(c->h_call + c->call_res_off + 32);
            const unsigned seq = ++c->call_seq ? c->call_seq : ++c->call_seq;
            std::unique_lock<std::mutex> gate_hold;          // (split launches of a device go one behind the other: see SplitGate)
            int rc = mdec_launch_lane(c, 0, &bd, 1, (fsz + 3) & ~(size_t)3, one, dstride, c->stream, (unsigned*)(c->d_call + c->call_res_off + 32), seq, &flagged, &gate_hold);
            if (rc) return rc;
            tick(3);
            if (flagged) {
                // the kernel's last group raises the flag when row and result are in this block: a look at our own memory instead of
                // the stream's completion signal.  (A flag that does not come within 2 ms: the stream is waited for.)
                struct timespec w0, w1;
                clock_gettime(CLOCK_MONOTONIC, &w0);
                for (unsigned spins = 0;; spins++) {
                    if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == seq) break;
                    __builtin_ia32_pause();
                    if ((spins & 1023u) == 1023u) {
                        clock_gettime(CLOCK_MONOTONIC, &w1);
                        if ((w1.tv_sec - w0.tv_sec) * 1000000000ll + (w1.tv_nsec - w0.tv_nsec) > 2000000ll) {
                            HIP_TRY(hipStreamSynchronize(c->stream), PSXHIP_EDEVICE);
                            break;
                        }
                    }
                }
            } else {
                HIP_TRY(hipStreamSynchronize(c->stream), PSXHIP_EDEVICE);
            }
            if (gate_hold.owns_lock()) gate_hold.unlock();
            tick(4);
            if (c->d_split_dbg) {          // diagnostics: where a split launch's time goes (one line per call on stderr)
                static int shown = 0;
                const int g = c->split_dbg_groups;
                std::vector<unsigned long long> t((size_t)g * 8);
                if (g > 0 && shown++ % 100 == 50 && hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(t.data(), c->d_split_dbg, t.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                    unsigned long long t0 = ~0ull;
                    for (int i = 0; i < g; i++) if (t[(size_t)i * 8] < t0) t0 = t[(size_t)i * 8];
                    fprintf(stderr, "split dbg (%d groups; ticks of 10 ns since the first group's start: min / max over groups)", g);
                    static const char* nm[8] = {"start", "dct", "counted", "met", "emitted", "left", "flag up", "last done"};
                    for (int k = 0; k < 8; k++) {
                        unsigned long long lo = ~0ull, hi = 0;
                        for (int i = 0; i < g; i++) { const unsigned long long v = t[(size_t)i * 8 + k]; if (v < t0) continue; if (v - t0 < lo) lo = v - t0; if (v - t0 > hi) hi = v - t0; }
                        fprintf(stderr, "  %s %llu/%llu", nm[k], lo == ~0ull ? 0 : lo, hi);
                    }
                    fprintf(stderr, "\n");
                }
            }
            memcpy(results, c->h_call + c->call_res_off, sizeof(psxhip_mdec_result_t));
            if (flagged && results[0].quant_scale >= 64) {
                // "nothing fits" from the split kernel is also what a frame says whose groups the watchdog released (another process
                // holding the CUs for 0.2 s): the frame kernel has the last word -- the reference aborts here, nobody waits for this
                HIP_TRY(hipStreamSynchronize(c->stream), PSXHIP_EDEVICE);
                rc = mdec_launch_lane(c, 0, &bd, 1, (fsz + 3) & ~(size_t)3, one, dstride, c->stream, nullptr, 0, nullptr, nullptr, true);
                if (rc) return rc;
                HIP_TRY(hipStreamSynchronize(c->stream), PSXHIP_EDEVICE);
                memcpy(results, c->h_call + c->call_res_off, sizeof(psxhip_mdec_result_t));
            }
            memcpy(out, c->h_call + c->call_out_off, (size_t)max_size);
            if (c->call_trace) {
                tick(5);
                for (int k = 0; k < 5; k++) c->call_ns[k] += (double)(ts[k + 1].tv_sec - ts[k].tv_sec) * 1e9 + (double)(ts[k + 1].tv_nsec - ts[k].tv_nsec);
                c->call_ns[5] += 1.0;
            }
            c->call_hint = results[0].quant_scale < 64 ? results[0].quant_scale : 0;
            if (results[0].quant_scale >= 64) {
                psxhip_set_error("frame %d does not fit %d bytes at any quant scale", 0, one);
                return PSXHIP_ENOFIT;
            }
            return PSXHIP_OK;
        }
    }
    // chunk: most of a GPU-load of frames -- small enough that a 1000-frame call already pipelines staging, DMA and kernel
    // over three chunks (353 k frames/s against 269 k with 1024-frame chunks), large enough for launches to stay efficient
    // (tools/gpu_chunk_sweep.py)
    int chunk = c->groups_max * 3 / 4;
    if (const char* e = getenv("PSXHIP_MDEC_CHUNK")) { const int v = atoi(e); if (v > 0) chunk = v; }      // experiments
    const size_t staging_cap = (size_t)96 << 20;                  // pinned bytes per staging buffer
    if ((size_t)chunk * fsz > staging_cap) chunk = (int)(staging_cap / fsz);
    if (chunk < 1) chunk = 1;
    if (chunk > n_frames) chunk = n_frames;
    if (!c->stream) HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), PSXHIP_EDEVICE);
    if (!c->stream2) HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking), PSXHIP_EDEVICE);
    if (chunk > c->cap_frames || dstride > c->cap_out_stride) {
        const int cap = chunk > c->cap_frames ? chunk : c->cap_frames;
        const size_t os = dstride > c->cap_out_stride ? dstride : c->cap_out_stride;
        psxhip_mdec_free_staging(c);
        for (int b = 0; b < 2; b++) {
            HIP_TRY(hipMalloc((void**)&c->d_frames[b], fsz * cap), PSXHIP_ENOMEM);
            HIP_TRY(hipMalloc((void**)&c->d_out[b], os * cap), PSXHIP_ENOMEM);
            HIP_TRY(hipMalloc((void**)&c->d_res[b], sizeof(psxhip_mdec_result_t) * cap), PSXHIP_ENOMEM);
            HIP_TRY(hipMalloc((void**)&c->d_sizes[b], sizeof(int32_t) * cap), PSXHIP_ENOMEM);
            HIP_TRY(hipHostMalloc((void**)&c->h_in[b], fsz * cap + sizeof(int32_t) * cap, hipHostMallocDefault), PSXHIP_ENOMEM);
            HIP_TRY(hipHostMalloc((void**)&c->h_out[b], os * cap + sizeof(psxhip_mdec_result_t) * cap, hipHostMallocDefault), PSXHIP_ENOMEM);
            if (!c->chunk_done[b]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_done[b], hipEventDisableTiming), PSXHIP_EDEVICE);
        }
        c->cap_frames = cap;
        c->cap_out_stride = os;
    }
    const size_t os = c->cap_out_stride;
    const int n_chunks = (n_frames + chunk - 1) / chunk;
    // Callers whose buffers are page-locked (hipHostMalloc, hipHostRegister / psxhip_host_register) are served by DMA
    // straight from / to their memory; pageable buffers go through the pinned staging buffers, copied by the CPU while the
    // GPU works on the neighbouring chunks.
    const bool in_pinned = host_pinned(frames), out_pinned = host_pinned(out);
    // hand a finished chunk to the caller
    auto deliver = [&](int k) {
        const int b = k & 1, first = k * chunk, cnt = (first + chunk <= n_frames) ? chunk : n_frames - first;
        const uint8_t* src = c->h_out[b];
        if (out_pinned) {
            // (already there)
        } else if (os == out_stride) {
            parallel_copy(out + (size_t)first * out_stride, src, os * (size_t)cnt);
        } else {
            for (int i = 0; i < cnt; i++) memcpy(out + (size_t)(first + i) * out_stride, src + (size_t)i * os, (size_t)max_size);
        }
        memcpy(results + first, src + os * (size_t)c->cap_frames, sizeof(psxhip_mdec_result_t) * (size_t)cnt);
    };
    for (int k = 0; k < n_chunks; k++) {
        const int b = k & 1, first = k * chunk, cnt = (first + chunk <= n_frames) ? chunk : n_frames - first;
        hipStream_t st = b ? c->stream2 : c->stream;
        if (k >= 2) {
            HIP_TRY(hipEventSynchronize(c->chunk_done[b]), PSXHIP_EDEVICE);      // chunk k-2 left the GPU: its buffers are free
            deliver(k - 2);
        }
        const uint8_t* src = frames + (size_t)first * fsz;
        if (!in_pinned) {
            parallel_copy(c->h_in[b], src, fsz * (size_t)cnt);
            src = c->h_in[b];
        }
        HIP_TRY(hipMemcpyAsync(c->d_frames[b], src, fsz * (size_t)cnt, hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
        if (frame_max_sizes) {
            int32_t* hs = (int32_t*)(c->h_in[b] + fsz * (size_t)c->cap_frames);
            memcpy(hs, frame_max_sizes + first, sizeof(int32_t) * (size_t)cnt);
            HIP_TRY(hipMemcpyAsync(c->d_sizes[b], hs, sizeof(int32_t) * (size_t)cnt, hipMemcpyHostToDevice, st), PSXHIP_EDEVICE);
            // rows are handed back max_size wide; bytes past a frame's own (smaller) budget read as zero
            HIP_TRY(hipMemsetAsync(c->d_out[b], 0, os * (size_t)cnt, st), PSXHIP_EDEVICE);
        }
        // chunk k runs on launch lane k & 1 = its stream's own set of frame tickets: neighbouring chunks' kernels overlap like
        // their copies do (the tail of one launch is filled by the head of the next); chunks k and k + 2 share stream and lane
        psxhip_mdec_batch_t bd;
        bd.d_frames = c->d_frames[b]; bd.n_frames = cnt; bd.reserved = 0; bd.d_frame_max_sizes = frame_max_sizes ? c->d_sizes[b] : nullptr;
        bd.d_out = c->d_out[b]; bd.d_results = c->d_res[b];
        int rc = mdec_launch_lane(c, b, &bd, 1, fsz, uniform_max_size, os, st);
        if (rc) return rc;
        if (out_pinned) {
            HIP_TRY(hipMemcpy2DAsync(out + (size_t)first * out_stride, out_stride, c->d_out[b], os, (size_t)max_size, (size_t)cnt,
                                     hipMemcpyDeviceToHost, st), PSXHIP_EDEVICE);
        } else {
            HIP_TRY(hipMemcpyAsync(c->h_out[b], c->d_out[b], os * (size_t)cnt, hipMemcpyDeviceToHost, st), PSXHIP_EDEVICE);
        }
        HIP_TRY(hipMemcpyAsync(c->h_out[b] + os * (size_t)c->cap_frames, c->d_res[b], sizeof(psxhip_mdec_result_t) * (size_t)cnt,
                               hipMemcpyDeviceToHost, st), PSXHIP_EDEVICE);
        HIP_TRY(hipEventRecord(c->chunk_done[b], st), PSXHIP_EDEVICE);
    }
    for (int k = n_chunks >= 2 ? n_chunks - 2 : 0; k < n_chunks; k++) {
        HIP_TRY(hipEventSynchronize(c->chunk_done[k & 1]), PSXHIP_EDEVICE);
        deliver(k);
    }
    for (int i = 0; i < n_frames; i++)
        if (results[i].quant_scale >= 64) {
            psxhip_set_error("frame %d does not fit %d bytes at any quant scale", i,
                             frame_max_sizes ? frame_max_sizes[i] : uniform_max_size);
            return PSXHIP_ENOFIT;
        }
    return PSXHIP_OK;
}

extern "C" int psxhip_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return PSXHIP_EINVAL;
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault), PSXHIP_EDEVICE);
    return PSXHIP_OK;
}

extern "C" int psxhip_host_unregister(void* p) {
    if (!p) return PSXHIP_EINVAL;
    HIP_TRY(hipHostUnregister(p), PSXHIP_EDEVICE);
    return PSXHIP_OK;
}

extern "C" int psxhip_mdec_read_stats(psxhip_mdec_ctx_t* c, unsigned long long* out, int n, int reset) {
    if (!c || !out || n < 0) return PSXHIP_EINVAL;
    if (n > PSXHIP_MDEC_STATS_TOTAL) n = PSXHIP_MDEC_STATS_TOTAL;
    memset(out, 0, (size_t)n * sizeof(unsigned long long));
    if (!c->d_stats) return PSXHIP_OK;
    HIP_TRY(hipSetDevice(c->device), PSXHIP_EDEVICE);
    HIP_TRY(hipDeviceSynchronize(), PSXHIP_EDEVICE);
    HIP_TRY(hipMemcpy(out, c->d_stats, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost), PSXHIP_EDEVICE);
    if (reset) HIP_TRY(hipMemset(c->d_stats, 0, PSXHIP_MDEC_STATS_TOTAL * sizeof(unsigned long long)), PSXHIP_EDEVICE);
    return PSXHIP_OK;
}

extern "C" int psxhip_mdec_fdct_host(int device, const int16_t* blocks, int n_blocks, int16_t* coefs) {
    if (!blocks || !coefs || n_blocks < 0) return PSXHIP_EINVAL;
    for (size_t i = 0; i < (size_t)n_blocks * 64; i++)
        if (blocks[i] < -128 || blocks[i] > 127) {
            psxhip_set_error("psxhip_mdec_fdct_host: sample %zu = %d outside -128..127", i, blocks[i]);
            return PSXHIP_EINVAL;
        }
    int rc = ensure_device(device);
    if (rc) return rc;
    if (n_blocks == 0) return PSXHIP_OK;
    int16_t *d_in = nullptr, *d_out = nullptr;
    const size_t bytes = (size_t)n_blocks * 64 * sizeof(int16_t);
    HIP_TRY(hipMalloc((void**)&d_in, bytes), PSXHIP_ENOMEM);
    if (hipMalloc((void**)&d_out, bytes) != hipSuccess) { (void)hipFree(d_in); return PSXHIP_ENOMEM; }
    hipError_t e = hipMemcpy(d_in, blocks, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = psxhip_mdec_fdct_launch(d_in, d_out, n_blocks, nullptr);
    if (e == hipSuccess) e = hipMemcpy(coefs, d_out, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) {
        psxhip_set_error("psxhip_mdec_fdct_host: %s", hipGetErrorString(e));
        return PSXHIP_EDEVICE;
    }
    return PSXHIP_OK;
}

extern "C" const char* psxhip_mdec_kernel_name(void) { return "mdec_encode_frames_kernel"; }
