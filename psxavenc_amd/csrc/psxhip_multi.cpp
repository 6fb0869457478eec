// psxhip_multi.cpp -- several devices behind one C call (include/psxav_hip.h, "several devices").
//
// The reference's host side is one C loop over frames (psxavenc/filefmt.c:633-662) or sectors (:450-503): a C caller has
// no ranks to shard over.  These entry points take a device LIST and do the sharding inside the call: one host thread,
// one encoder context and one pair of pinned staging buffers per list entry; frames are independent units
// (encode_frame_bs resets all bit / DC state per attempt, mdec.c:678-686), so every schedule produces the bytes of the
// single-device call.  A device may be listed more than once (two contexts on one GPU overlap each other's copies).
//
// Schedules: contiguous ranges (psxhip_shard_range, the same partition psxavenc_amd/parallel.py uses across ranks), or
// a host-side ticket queue of frame-range chunks -- a worker that finishes its range draws the next one, so a device
// that drew expensive frames (content at a scale boundary costs up to 1.5x) does not hold the call up.  No device-side
// collective is involved: the only shared state is one atomic counter in host memory.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "psxhip_internal.h"

extern "C" void psxhip_shard_range(int64_t n_units, int rank, int world, int64_t* first, int64_t* count) {
    int64_t f = 0, c = 0;
    if (world > 0 && rank >= 0 && rank < world && n_units >= 0) {
        const int64_t base = n_units / world, extra = n_units % world;
        f = (int64_t)rank * base + (rank < extra ? rank : extra);
        c = base + (rank < extra ? 1 : 0);
    }
    if (first) *first = f;
    if (count) *count = c;
}

// ---- ticket queue: [0, n_units) handed out in ranges of ticket_units, in order, each exactly once
struct psxhip_ticket_queue {
    std::atomic<int64_t> next;
    int64_t n_units, ticket_units;
};

extern "C" psxhip_ticket_queue_t* psxhip_ticket_queue_create(int64_t n_units, int64_t ticket_units) {
    if (n_units < 0 || ticket_units <= 0) return nullptr;
    psxhip_ticket_queue* q = new (std::nothrow) psxhip_ticket_queue;
    if (!q) return nullptr;
    q->next.store(0);
    q->n_units = n_units;
    q->ticket_units = ticket_units;
    return q;
}

extern "C" int psxhip_ticket_queue_next(psxhip_ticket_queue_t* q, int64_t* first, int64_t* count) {
    if (!q) return 0;
    const int64_t f = q->next.fetch_add(q->ticket_units, std::memory_order_relaxed);
    if (f >= q->n_units) return 0;
    if (first) *first = f;
    if (count) *count = f + q->ticket_units <= q->n_units ? q->ticket_units : q->n_units - f;
    return 1;
}

extern "C" void psxhip_ticket_queue_destroy(psxhip_ticket_queue_t* q) { delete q; }

// ------------------------------------------------------------------------------------------ MDEC

struct psxhip_mdec_multi {
    std::vector<int> devices;
    std::vector<psxhip_mdec_ctx_t*> ctx;
    int codec, width, height, max_frame_size;
};

extern "C" int psxhip_mdec_multi_create(psxhip_mdec_multi_t** out, const int* devices, int n_devices, int codec, int width,
                                        int height, int max_frame_size) {
    if (!out) return PSXHIP_EINVAL;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) {
        psxhip_set_error("psxhip_mdec_multi_create: need 1..64 devices");
        return PSXHIP_EINVAL;
    }
    psxhip_mdec_multi* m = new (std::nothrow) psxhip_mdec_multi;
    if (!m) return PSXHIP_ENOMEM;
    m->codec = codec; m->width = width; m->height = height; m->max_frame_size = max_frame_size;
    for (int i = 0; i < n_devices; i++) {
        psxhip_mdec_ctx_t* c = nullptr;
        const int rc = psxhip_mdec_create(&c, devices[i], codec, width, height, max_frame_size);
        if (rc) {
            psxhip_mdec_multi_destroy(m);
            return rc;
        }
        m->devices.push_back(devices[i]);
        m->ctx.push_back(c);
    }
    *out = m;
    return PSXHIP_OK;
}

extern "C" void psxhip_mdec_multi_destroy(psxhip_mdec_multi_t* m) {
    if (!m) return;
    for (psxhip_mdec_ctx_t* c : m->ctx) psxhip_mdec_destroy(c);
    delete m;
}

extern "C" int psxhip_mdec_multi_device_count(const psxhip_mdec_multi_t* m) { return m ? (int)m->ctx.size() : 0; }

extern "C" int psxhip_mdec_multi_encode_frames_host(psxhip_mdec_multi_t* m, const uint8_t* frames, int n_frames,
                                                    const int32_t* frame_max_sizes, int uniform_max_size, uint8_t* out,
                                                    size_t out_stride, psxhip_mdec_result_t* results, int schedule,
                                                    int ticket_frames, psxhip_multi_report_t* report) {
    if (!m || !frames || !out || !results || n_frames < 0 || (schedule != PSXHIP_SCHED_STATIC && schedule != PSXHIP_SCHED_TICKETS)) {
        psxhip_set_error("psxhip_mdec_multi_encode_frames_host: bad argument");
        return PSXHIP_EINVAL;
    }
    const int nd = (int)m->ctx.size();
    if (report)
        for (int d = 0; d < nd && d < PSXHIP_MULTI_MAX_REPORT; d++) report[d] = psxhip_multi_report_t{m->devices[(size_t)d], 0, 0, 0.0};
    if (n_frames == 0) return PSXHIP_OK;
    const size_t fsz = (size_t)m->width * m->height * 3 / 2;
    if (schedule == PSXHIP_SCHED_TICKETS && ticket_frames <= 0) {
        // default ticket: a few GPU-loads of frames (the single-device path pipelines staging, DMA and kernel over chunks
        // of 3/4 of a GPU-load; a ticket of 1536 frames is four of those at 320x240) but at least 8 tickets per device
        ticket_frames = 1536;
        while (ticket_frames > 64 && (int64_t)ticket_frames * nd * 8 > n_frames) ticket_frames /= 2;
    }
    // the whole batch is vetted before any shard runs, like the single-device call does (a bad budget must not leave the other
    // shards' rows written, and the index reported is the call's, not a shard's)
    const int cap = m->max_frame_size;
    int row_bytes = 0;                     // per-frame budgets: every row as wide as the unsplit call writes it
    if (frame_max_sizes) {
        for (int i = 0; i < n_frames; i++) {
            if (frame_max_sizes[i] < 8 || frame_max_sizes[i] > cap) {
                psxhip_set_error("psxhip_mdec_multi_encode_frames_host: frame %d budget %d outside [8, %d]", i, frame_max_sizes[i], cap);
                return PSXHIP_EINVAL;
            }
            if (frame_max_sizes[i] > row_bytes) row_bytes = frame_max_sizes[i];
        }
    } else if (uniform_max_size < 8 || uniform_max_size > cap) {
        psxhip_set_error("psxhip_mdec_multi_encode_frames_host: frame_max_size %d outside [8, %d]", uniform_max_size, cap);
        return PSXHIP_EINVAL;
    }
    if ((size_t)(frame_max_sizes ? row_bytes : uniform_max_size) > out_stride) {
        psxhip_set_error("psxhip_mdec_multi_encode_frames_host: out_stride %zu smaller than the largest budget %d", out_stride,
                         frame_max_sizes ? row_bytes : uniform_max_size);
        return PSXHIP_EINVAL;
    }
    psxhip_ticket_queue_t* q = schedule == PSXHIP_SCHED_TICKETS ? psxhip_ticket_queue_create(n_frames, ticket_frames) : nullptr;
    if (schedule == PSXHIP_SCHED_TICKETS && !q) return PSXHIP_ENOMEM;

    std::vector<int> rcs((size_t)nd, PSXHIP_OK);
    std::vector<std::string> errs((size_t)nd);
    std::vector<psxhip_multi_report_t> rep((size_t)nd);
    auto worker = [&](int d) {
        const auto t0 = std::chrono::steady_clock::now();
        rep[(size_t)d] = psxhip_multi_report_t{m->devices[(size_t)d], 0, 0, 0.0};
        auto run = [&](int64_t first, int64_t count) {
            if (count <= 0) return;
            const int rc = psxhip_mdec_encode_frames_host_rows(m->ctx[(size_t)d], frames + (size_t)first * fsz, (int)count,
                                                               frame_max_sizes ? frame_max_sizes + first : nullptr, uniform_max_size,
                                                               out + (size_t)first * out_stride, out_stride, results + first, row_bytes);
            if (rc && rcs[(size_t)d] == PSXHIP_OK) {          // keep the first failure (ENOFIT is reported after all frames ran)
                rcs[(size_t)d] = rc;
                errs[(size_t)d] = psxhip_last_error();
                if (rc == PSXHIP_ENOFIT) {                    // frame index inside this range -> index inside the call
                    for (int64_t i = 0; i < count; i++)
                        if (results[first + i].quant_scale >= 64) {
                            char b[160];
                            snprintf(b, sizeof b, "frame %lld does not fit its budget at any quant scale", (long long)(first + i));
                            errs[(size_t)d] = b;
                            break;
                        }
                }
            }
            rep[(size_t)d].units += count;
            rep[(size_t)d].tickets += 1;
        };
        if (q) {
            int64_t first, count;
            while ((rcs[(size_t)d] == PSXHIP_OK || rcs[(size_t)d] == PSXHIP_ENOFIT) && psxhip_ticket_queue_next(q, &first, &count)) run(first, count);
        } else {
            int64_t first, count;
            psxhip_shard_range(n_frames, d, nd, &first, &count);
            run(first, count);
        }
        rep[(size_t)d].seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    std::vector<std::thread> th;
    for (int d = 1; d < nd; d++) th.emplace_back(worker, d);
    worker(0);
    for (auto& t : th) t.join();
    psxhip_ticket_queue_destroy(q);
    if (report)
        for (int d = 0; d < nd && d < PSXHIP_MULTI_MAX_REPORT; d++) report[d] = rep[(size_t)d];
    // a hard failure outranks "some frame does not fit"
    int rc = PSXHIP_OK;
    for (int d = 0; d < nd; d++)
        if (rcs[(size_t)d] && (rc == PSXHIP_OK || (rc == PSXHIP_ENOFIT && rcs[(size_t)d] != PSXHIP_ENOFIT))) {
            rc = rcs[(size_t)d];
            psxhip_set_error("device %d: %s", m->devices[(size_t)d], errs[(size_t)d].c_str());
        }
    return rc;
}

// ------------------------------------------------------------------------------------------ XA streams

// independent XA streams over several devices: contiguous stream ranges, one host thread per device
extern "C" int psxhip_xa_encode_streams_host_multi(const int* devices, int n_devices, int format, int stereo, int frequency,
                                                   int bits, int file_number, int channel_number, const int16_t* samples,
                                                   int n_streams, int64_t stream_stride, int samples_per_stream,
                                                   const int32_t* lbas, psxhip_adpcm_state_t* states, uint8_t* out,
                                                   int64_t out_stride, int finalize, psxhip_multi_report_t* report) {
    if (!devices || n_devices < 1 || n_devices > 64 || n_streams < 0 || !samples || !states || !out) {
        psxhip_set_error("psxhip_xa_encode_streams_host_multi: bad argument");
        return PSXHIP_EINVAL;
    }
    {   // (checked here, before any pointer is offset by a shard's first stream and before any thread is started)
        const int have = psxhip_device_count();
        if (have <= 0) {
            psxhip_set_error("no HIP device visible (libpsxav_hip has no CPU fallback)");
            return PSXHIP_EDEVICE;
        }
        for (int i = 0; i < n_devices; i++)
            if (devices[i] < 0 || devices[i] >= have) {
                psxhip_set_error("psxhip_xa_encode_streams_host_multi: device %d out of range (%d visible)", devices[i], have);
                return PSXHIP_EINVAL;
            }
    }
    const int nd = n_devices < n_streams ? n_devices : (n_streams > 0 ? n_streams : 1);
    if (report)
        for (int d = 0; d < n_devices && d < PSXHIP_MULTI_MAX_REPORT; d++) report[d] = psxhip_multi_report_t{devices[d], 0, 0, 0.0};
    if (n_streams <= 1 || nd == 1) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = psxhip_xa_encode_streams_host(devices[0], format, stereo, frequency, bits, file_number, channel_number, samples,
                                                     n_streams, stream_stride, samples_per_stream, lbas, states, out, out_stride, finalize);
        if (report) {
            report[0].units = n_streams;
            report[0].tickets = 1;
            report[0].seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        return rc;
    }
    const int ch = stereo ? 2 : 1;
    std::vector<int> rcs((size_t)nd, 0);
    std::vector<std::string> errs((size_t)nd);
    std::vector<psxhip_multi_report_t> rep((size_t)nd);
    auto worker = [&](int d, bool own_thread) {
        const auto t0 = std::chrono::steady_clock::now();
        int64_t first, count;
        psxhip_shard_range(n_streams, d, nd, &first, &count);
        // a sub-batch of one stream would drop the caller's strides (the single-stream call ignores them): hand it two-stream
        // geometry by keeping the strides explicit -- the callee only overrides them when n_streams == 1, which is right then
        rcs[(size_t)d] = psxhip_xa_encode_streams_host(devices[d], format, stereo, frequency, bits, file_number, channel_number,
                                                       samples + first * stream_stride, (int)count, stream_stride, samples_per_stream,
                                                       lbas ? lbas + first : nullptr, states + first * ch, out + first * out_stride,
                                                       out_stride, finalize);
        if (rcs[(size_t)d] < 0) errs[(size_t)d] = psxhip_last_error();
        if (own_thread) psxhip_release_scratch();            // the scratch pool is per thread, and this thread ends here
        rep[(size_t)d] = psxhip_multi_report_t{devices[d], count, 1, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()};
    };
    std::vector<std::thread> th;
    for (int d = 1; d < nd; d++) th.emplace_back(worker, d, true);
    worker(0, false);
    for (auto& t : th) t.join();
    if (report)
        for (int d = 0; d < nd && d < PSXHIP_MULTI_MAX_REPORT; d++) report[d] = rep[(size_t)d];
    for (int d = 0; d < nd; d++)
        if (rcs[(size_t)d] < 0) {
            psxhip_set_error("device %d: %s", devices[d], errs[(size_t)d].c_str());
            return rcs[(size_t)d];
        }
    return rcs[0];
}
