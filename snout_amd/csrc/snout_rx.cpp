// snout_rx.cpp — the C ABI of libsnout_rx.so (include/snout_rx.h): handle management, argument
// checking, the three-slot submit/collect pipeline (record D2H of segment i overlaps the kernels of
// segment i+1 on a copy stream), H->D staging for the host-pointer entry point, profiling.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <new>
#include <algorithm>

namespace snout {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int DevBuf::ensure(size_t bytes)
{
    if (bytes <= cap && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        set_last_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        p = nullptr;
        return SNOUT_ENOMEM;
    }
    cap = want;
    return 0;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

bool host_is_pinned(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

int ResultSlot::init()
{
    if (int rc = d_totals.ensure(kTotalsBytes)) return rc;
    SNOUT_HIP(hipHostMalloc((void**)&h_totals, 64, hipHostMallocDefault));
    memset(h_totals, 0, 64);
    SNOUT_HIP(hipEventCreate(&ev_k0));
    SNOUT_HIP(hipEventCreate(&ev_k1));
    SNOUT_HIP(hipEventCreate(&ev_front));
    SNOUT_HIP(hipEventCreate(&ev_compute));
    SNOUT_HIP(hipEventCreate(&ev_copy));
    return 0;
}

void ResultSlot::destroy()
{
    d_out.release();
    d_totals.release();
    if (h_totals) (void)hipHostFree(h_totals);
    if (h_recs) (void)hipHostFree(h_recs);
    h_totals = nullptr; h_recs = nullptr; h_cap = 0;
    if (ev_k0) {
        (void)hipEventDestroy(ev_k0); (void)hipEventDestroy(ev_k1);
        (void)hipEventDestroy(ev_front); (void)hipEventDestroy(ev_compute); (void)hipEventDestroy(ev_copy);
        ev_k0 = nullptr;
    }
}

int ResultSlot::ensure_host(uint64_t recs)
{
    if (recs <= h_cap) return 0;
    if (h_recs) (void)hipHostFree(h_recs);
    h_recs = nullptr;
    h_cap = recs + recs / 4 + 1024;
    SNOUT_HIP(hipHostMalloc((void**)&h_recs, h_cap * sizeof(snout_pkt), hipHostMallocDefault));
    return 0;
}

}  // namespace snout

using namespace snout;

struct snout_rx {
    snout_rx_cfg cfg;
    int device = 0;
    bool wide = false;
    // Work sets: the tail of segment i overlaps the front end of i+1.  BTLE has three, so that with
    // three segments in flight the set a new segment takes was released by a segment already
    // collected (no wait to enqueue).  Zigbee has three as well since round 5: its tail holds two latency-bound
    // kernels (zb_walk, zb_repair: a few waves that run for one to two milliseconds), and with a third set the front
    // end of segment i + 2 does not wait for them.  Every work set has its own tail stream, so the tails of
    // consecutive segments overlap each other too.
    BtleCtx btle, btle2, btle3;
    ZbCtx zb2, zb3;           // (second and third Zigbee work set; the first is `zb`)
    hipStream_t tail_streams[3] = {nullptr, nullptr, nullptr};
    bool ext_launch = true;   // SNOUT_EXT_LAUNCH=0 (A/B): events around the front-end kernels as recorded barrier packets
    int n_tails = 1;          // tail streams CREATED (SNOUT_TAIL_STREAMS overrides): work set k's tail runs on stream k % n_tails
    bool sync_call = false;   // inside snout_rx_process*: nothing to overlap, the tail stays on the caller's stream
    ZbCtx zb;
    PfbCtx pfb;
    DevBuf d_iq;              // staging for snout_rx_process (host input)
    static constexpr int kSlots = 3;    // segments in flight: front end i+1 is queued before copy i-1 lands
    ResultSlot slots[kSlots];
    hipEvent_t ws_free[3] = {nullptr, nullptr, nullptr};   // the last tail that used work set k has finished
    uint64_t n_submitted = 0;
    hipStream_t copy_stream = nullptr;
    int head = 0, pending = 0;          // ring of submitted, not yet collected segments
    uint64_t spec = 0;                  // records copied speculatively with the totals
    // pool of event pairs around the dominant kernel: durations are read back after the fact
    // (snout_rx_profile_history) so that timing does not perturb a pipelined run
    static constexpr int kHist = 64;
    hipEvent_t hist_k0[kHist] = {}, hist_k1[kHist] = {};
    uint64_t hist_n = 0;                // segments submitted with timing
    // last collected segment (profile / soft taps)
    ResultSlot* last = nullptr;
    uint64_t last_n = 0, last_nch = 0, last_pkts = 0;
};

static uint32_t sample_bytes(uint32_t fmt)
{
    return fmt == SNOUT_FMT_SC8 ? 2u : (fmt == SNOUT_FMT_SC16 ? 4u : 8u);
}

static BtleCtx& btle_of(snout_rx* h, const ResultSlot& s)
{
    return s.work_set == 0 ? h->btle : (s.work_set == 1 ? h->btle2 : h->btle3);
}
static ZbCtx& zb_of(snout_rx* h, const ResultSlot& s) { return s.work_set == 0 ? h->zb : (s.work_set == 1 ? h->zb2 : h->zb3); }

// Enqueue every kernel of one segment, results into slot s.  No host synchronisation.
// BTLE: the front end (demod+correlate, or channelizer+correlate) runs on the caller's stream `st`;
// the O(candidates) tail runs on an internal stream behind an event, on the slot's own work set, so
// the next segment's front end does not wait for it.
static int enqueue_segment(snout_rx* h, ResultSlot& s, hipStream_t st)
{
    const void* ch_iq = s.iq;
    int ch_fmt = (int)h->cfg.sample_format;          // the channelizer always emits cf32
    uint64_t n_ch = s.n_in, ch_stride = s.n_in;
    // Every event on the caller's stream is a barrier packet between consecutive front-end kernels
    // (~8 us each): the narrowband BTLE path, whose kernel is the whole front end, reuses the pair
    // around the kernel as start-of-segment and front-end-done events.
    const bool nb_btle = !h->wide && h->cfg.proto == SNOUT_PROTO_BTLE;
    // A one-segment-at-a-time call has nothing to overlap the tail with: it stays on the caller's
    // stream, in order behind the front end (no event wait between the two, 2-4 % faster).
    // So does the (tiny) BTLE tail of a small pipelined segment: the two cross-stream hand-overs cost
    // more than the ~50 us of kernels they would overlap (48 wideband segments of 2^24 samples: 12.3
    // -> 9.9 ms); behind a 1e9-sample front end the separate stream is worth 5 %.
    const uint64_t ch_samples = (h->wide ? h->pfb.n_out_for(s.n_in) * h->cfg.n_channels : s.n_in) * s.segs.count;
    const bool inline_tail = h->sync_call || (h->cfg.proto == SNOUT_PROTO_BTLE && ch_samples < (1ull << 26));
    hipStream_t tail = inline_tail ? st : h->tail_streams[s.work_set % h->n_tails];
    // (the first kernel's start event ev_k0 also marks the start of the segment: every event on the
    //  caller's stream is a barrier packet, so there is no separate one)
    // the tail that last used this work set must be done; usually it is, and a wait that is not
    // enqueued is one barrier packet less between two front-end kernels
    if (hipEventQuery(h->ws_free[s.work_set]) != hipSuccess)
        SNOUT_HIP(hipStreamWaitEvent(st, h->ws_free[s.work_set], 0));
    // fused wideband modes (unless the caller keeps channel IQ for the CHAN_IQ tap): BTLE hard bits
    // straight into the bit planes, 802.15.4 discriminator output straight into the Zigbee context
    const bool fused = h->wide && h->cfg.proto == SNOUT_PROTO_BTLE && !(h->cfg.flags & SNOUT_CFG_KEEP_CHANNEL_IQ);
    const bool fused_zb = h->wide && h->cfg.proto == SNOUT_PROTO_ZIGBEE && !(h->cfg.flags & SNOUT_CFG_KEEP_CHANNEL_IQ);
    if (h->wide) {
        n_ch = h->pfb.n_out_for(s.n_in);
        BtleCtx& bw = btle_of(h, s);
        if (fused) { if (int rc = bw.reserve(n_ch, s.segs.count)) return rc; }
        PfbZbTarget zt{};
        if (fused_zb) {
            ZbCtx& z = zb_of(h, s);
            if (int rc = z.reserve(n_ch, s.segs.count)) return rc;
            zt = z.pfb_target();
        }
        // ext_launch: the channelizer's stop event and the correlator's "front end done" event ride on the kernels' own
        // dispatches (hipExtLaunchKernel) -- every hipEventRecord on the caller's stream is a barrier packet of its own
        // between two front-end kernels (SNOUT_EXT_LAUNCH=0: recorded events, rounds 1-5)
        if (h->ext_launch) { h->pfb.ev_start = s.ev_k0; h->pfb.ev_stop = s.ev_k1; }
        else SNOUT_HIP(hipEventRecord(s.ev_k0, st));
        if (s.segs.count > 1) {
            // a batch: every segment's channelizer pass in one launch, outputs one segment stride apart
            if (fused) {
                if (int rc = h->pfb.run_batch(s.iq_more, s.segs.count, s.n_in, st, bw.d_planes.as<uint16_t>(), bw.plane_stride,
                                              (uint64_t)bw.seg_slots * bw.plane_stride * 4u, nullptr, 0, 0, ch_fmt)) return rc;
            } else {
                const ZbCtx& z = zb_of(h, s);
                if (int rc = h->pfb.run_batch(s.iq_more, s.segs.count, s.n_in, st, nullptr, 0, 0, n_ch >= 9u ? &zt : nullptr,
                                              (uint64_t)z.seg_slots * z.d_stride, (uint64_t)z.seg_slots * z.nsb, ch_fmt)) return rc;
            }
        } else if (int rc = h->pfb.run(s.iq, s.n_in, st, fused ? bw.d_planes.as<uint16_t>() : nullptr,
                                       bw.plane_stride, (fused_zb && n_ch >= 9u) ? &zt : nullptr, ch_fmt)) return rc;
        ch_fmt = 0;
        if (!h->ext_launch) SNOUT_HIP(hipEventRecord(s.ev_k1, st));
        ch_iq = (fused_zb && n_ch >= 9u) ? nullptr : h->pfb.d_y.as<float>();
        ch_stride = h->pfb.y_stride;
    }
    if (h->cfg.proto == SNOUT_PROTO_BTLE) {
        BtleCtx& b = btle_of(h, s);
        if (int rc = b.reserve(n_ch, s.segs.count)) return rc;
        bool front_bound = false;
        if (fused) {
            front_bound = h->ext_launch;
            if (int rc = b.launch_corr_planes(n_ch, st, front_bound ? s.ev_front : nullptr)) return rc;       // bits are already in the planes
        } else {
            if (int rc = b.launch_demod_corr(ch_iq, n_ch, ch_stride, st, h->wide ? nullptr : &s, ch_fmt)) return rc;
        }
        if (!nb_btle && !front_bound) SNOUT_HIP(hipEventRecord(s.ev_front, st));
        if (!inline_tail) SNOUT_HIP(hipStreamWaitEvent(tail, nb_btle ? s.ev_k1 : s.ev_front, 0));
        if (int rc = b.enqueue_tail(n_ch, s.segs, tail, s)) return rc;
        SNOUT_HIP(hipEventRecord(s.ev_compute, tail));
        SNOUT_HIP(hipEventRecord(h->ws_free[s.work_set], tail));
        return 0;
    } else {
        ZbCtx& z = zb_of(h, s);
        if (int rc = z.reserve(n_ch, s.segs.count)) return rc;
        if (int rc = z.enqueue_front(ch_iq, n_ch, ch_stride, st, s, !h->wide, ch_fmt)) return rc;
        // the lanes: a narrowband handle's on the work set's stream (the next segment's discriminator overlaps them),
        // a wideband handle's behind its channelizer (see ZbCtx::enqueue_lanes)
        // (measured in round 6 and dropped, profiles/r6_split.md: the wideband lanes on the work set's stream beside the NEXT
        //  segment's channelizer on CUs its grid leaves free -- the chip is busy either way, the step does not move)
        if (h->wide) { if (int rc = z.enqueue_lanes(n_ch, st)) return rc; }
        SNOUT_HIP(hipEventRecord(s.ev_front, st));
        if (!inline_tail) SNOUT_HIP(hipStreamWaitEvent(tail, s.ev_front, 0));
        if (!h->wide) { if (int rc = z.enqueue_lanes(n_ch, tail)) return rc; }
        if (int rc = z.enqueue_tail(n_ch, s.segs, tail, s, !h->wide)) return rc;
        SNOUT_HIP(hipEventRecord(s.ev_compute, tail));
        SNOUT_HIP(hipEventRecord(h->ws_free[s.work_set], tail));
        return 0;
    }
}

// Totals (and `spec` records, speculatively) -> pinned host memory on the copy stream.
static int enqueue_copy(snout_rx* h, ResultSlot& s, uint64_t spec)
{
    // one-segment-at-a-time calls stay on the caller's stream throughout (see enqueue_segment)
    hipStream_t cs = h->sync_call ? s.stream : h->copy_stream;
    if (!h->sync_call) SNOUT_HIP(hipStreamWaitEvent(cs, s.ev_compute, 0));
    SNOUT_HIP(hipMemcpyAsync(s.h_totals, s.d_totals.p, 16, hipMemcpyDeviceToHost, cs));
    s.spec_copied = 0;
    if (h->cfg.flags & SNOUT_CFG_RECORDS_ON_DEVICE) spec = 0;       // the caller takes them from the device copy
    if (spec) {
        const uint64_t room = s.d_out.cap / sizeof(snout_pkt);
        spec = spec < room ? spec : room;
        if (int rc = s.ensure_host(spec)) return rc;
        SNOUT_HIP(hipMemcpyAsync(s.h_recs, s.d_out.p, spec * sizeof(snout_pkt), hipMemcpyDeviceToHost, cs));
        s.spec_copied = spec;
    }
    SNOUT_HIP(hipEventRecord(s.ev_copy, cs));
    return 0;
}

// Wait for a slot; rerun with grown capacity on overflow.  On return the record count is known.
static int finish_slot(snout_rx* h, ResultSlot& s)
{
    for (int attempt = 0; attempt < 12; attempt++) {
        SNOUT_HIP(hipEventSynchronize(s.ev_copy));
        const bool over = h->cfg.proto == SNOUT_PROTO_BTLE ? btle_of(h, s).check_overflow(s)
                                                           : zb_of(h, s).check_overflow(s);
        if (!over) { s.n_pkts = s.h_totals[1]; return SNOUT_OK; }
        // other segments in flight share this slot's work set every second submit: let them finish
        // (their results are already on their way to their own slots) before reusing it
        SNOUT_HIP(hipDeviceSynchronize());
        std::swap(s.ev_k0, h->hist_k0[s.hist_idx]);     // time the rerun with the same pool pair
        std::swap(s.ev_k1, h->hist_k1[s.hist_idx]);
        int rc = enqueue_segment(h, s, s.stream);
        if (!rc) rc = enqueue_copy(h, s, 0);
        std::swap(s.ev_k0, h->hist_k0[s.hist_idx]);
        std::swap(s.ev_k1, h->hist_k1[s.hist_idx]);
        if (rc) return rc;
    }
    return SNOUT_EOVERFLOW;
}

static void note_last(snout_rx* h, ResultSlot& s)
{
    h->last = &s;
    h->last_n = s.n_in;
    h->last_nch = h->wide ? h->pfb.n_out_for(s.n_in) : s.n_in;
    h->last_pkts = s.n_pkts;
}

extern "C" {

uint32_t snout_abi_version(void) { return SNOUT_ABI_VERSION; }

const char* snout_last_error(void) { return g_err; }

const char* snout_strerror(int code)
{
    switch (code) {
        case SNOUT_OK: return "ok";
        case SNOUT_EINVAL: return "invalid argument";
        case SNOUT_ENODEV: return "no usable HIP device (libsnout_rx has no CPU fallback)";
        case SNOUT_ENOMEM: return "out of memory";
        case SNOUT_EHIP: return "HIP runtime error";
        case SNOUT_EOVERFLOW: return "capacity exceeded";
        case SNOUT_ERANGE: return "segment too long";
        default: return "unknown error";
    }
}

int snout_rx_create(const snout_rx_cfg* cfg, snout_rx** out)
{
    if (!cfg || !out) return SNOUT_EINVAL;
    *out = nullptr;
    if (cfg->abi_version != SNOUT_ABI_VERSION) {
        set_last_error("abi_version %u != %u", cfg->abi_version, SNOUT_ABI_VERSION);
        return SNOUT_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_last_error("no HIP device visible");
        return SNOUT_ENODEV;
    }
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) return SNOUT_ENODEV; }
    if (dev >= ndev) { set_last_error("device %d of %d", dev, ndev); return SNOUT_EINVAL; }
    SNOUT_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    SNOUT_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error("device %d is %s; libsnout_rx is built for gfx950 only", dev, prop.gcnArchName);
        return SNOUT_ENODEV;
    }
    snout_rx* h = new (std::nothrow) snout_rx();
    if (!h) return SNOUT_ENOMEM;
    h->cfg = *cfg;
    h->device = dev;
    snout_rx_cfg& c = h->cfg;
    if (c.access_addr == 0) c.access_addr = 0x8E89BED6u;
    if (c.crc_init == 0) c.crc_init = 0x555555u;
    if (c.chip_threshold == 0) c.chip_threshold = 10;
    if (c.taps_per_branch == 0) c.taps_per_branch = 16;
    if (c.zb_core == 0 && c.zb_warmup == 0) snout_zigbee_lane_shape(c.n_channels, &c.zb_core, &c.zb_warmup);   // the default shape, fixed per handle (by its kind: n_channels is defaulted below, 0 = narrowband)
    if (c.zb_core == 0) c.zb_core = 2048;
    if (c.zb_warmup == 0) c.zb_warmup = 512;
    if (c.n_channels == 0) c.n_channels = 1;
    if (c.batch_segments == 0) c.batch_segments = 1;
    int rc = SNOUT_EINVAL;
    if (c.batch_segments > kMaxBatch || (c.batch_segments > 1 && (c.n_channels == 1 || (c.flags & SNOUT_CFG_KEEP_CHANNEL_IQ)))) {
        set_last_error("batch_segments %u: 1..%u, wideband handles without SNOUT_CFG_KEEP_CHANNEL_IQ only",
                       c.batch_segments, kMaxBatch);
        goto fail;
    }
    if (c.sample_format > SNOUT_FMT_SC16) { set_last_error("sample format %u", c.sample_format); goto fail; }
    if (c.proto == SNOUT_PROTO_ZIGBEE &&
        (c.zb_core < 1024 || c.zb_core > (1u << 24) || c.zb_warmup > (1u << 20))) {
        set_last_error("zb_core %u / zb_warmup %u out of range", c.zb_core, c.zb_warmup);
        goto fail;
    }
    if (c.proto == SNOUT_PROTO_BTLE && c.n_channels == 1) {
        if (c.channel > 39) { set_last_error("BTLE channel %u", c.channel); goto fail; }
        uint16_t ch = (uint16_t)c.channel;
        rc = h->btle.init(1, &ch, c.access_addr, c.crc_init, c.max_hits);
        if (!rc) rc = h->btle2.init(1, &ch, c.access_addr, c.crc_init, c.max_hits);
        if (!rc) rc = h->btle3.init(1, &ch, c.access_addr, c.crc_init, c.max_hits);
        if (rc) goto fail;
    } else if (c.proto == SNOUT_PROTO_ZIGBEE && c.n_channels == 1) {
        if (c.channel < 11 || c.channel > 26) { set_last_error("Zigbee channel %u", c.channel); goto fail; }
        uint16_t ch = (uint16_t)c.channel;
        rc = h->zb.init(1, &ch, c.chip_threshold, c.zb_core, c.zb_warmup);
        if (!rc) rc = h->zb2.init(1, &ch, c.chip_threshold, c.zb_core, c.zb_warmup);
        if (!rc) rc = h->zb3.init(1, &ch, c.chip_threshold, c.zb_core, c.zb_warmup);
        if (rc) goto fail;
    } else if ((c.proto == SNOUT_PROTO_BTLE && c.n_channels == 40) ||
               (c.proto == SNOUT_PROTO_ZIGBEE && c.n_channels == 16)) {
        // wideband: M-branch channelizer, every bin is a channel slot
        h->wide = true;
        const uint32_t M = c.n_channels;
        uint16_t chs[40];
        for (uint32_t b = 0; b < M; b++)
            chs[b] = c.proto == SNOUT_PROTO_BTLE ? (uint16_t)snout_btle_rf_to_channel((b + 20u) % 40u)
                                                 : (uint16_t)(11u + (b + 8u) % 16u);
        if (c.taps_per_branch != 16) { set_last_error("taps_per_branch must be 16"); goto fail; }
        rc = h->pfb.init(M, (uint32_t)prop.multiProcessorCount, c.reserved_cus);
        if (rc) goto fail;
        rc = c.proto == SNOUT_PROTO_BTLE ? h->btle.init(M, chs, c.access_addr, c.crc_init, c.max_hits, c.batch_segments)
                                         : h->zb.init(M, chs, c.chip_threshold, c.zb_core, c.zb_warmup, c.batch_segments);
        if (!rc && c.proto == SNOUT_PROTO_BTLE) rc = h->btle2.init(M, chs, c.access_addr, c.crc_init, c.max_hits, c.batch_segments);
        if (!rc && c.proto == SNOUT_PROTO_BTLE) rc = h->btle3.init(M, chs, c.access_addr, c.crc_init, c.max_hits, c.batch_segments);
        if (!rc && c.proto == SNOUT_PROTO_ZIGBEE) rc = h->zb2.init(M, chs, c.chip_threshold, c.zb_core, c.zb_warmup, c.batch_segments);
        if (!rc && c.proto == SNOUT_PROTO_ZIGBEE) rc = h->zb3.init(M, chs, c.chip_threshold, c.zb_core, c.zb_warmup, c.batch_segments);
        if (rc) goto fail;
    } else {
        set_last_error("configuration proto=%u n_channels=%u not supported (BTLE: 1 or 40, "
                       "Zigbee: 1 or 16)", c.proto, c.n_channels);
        goto fail;
    }
    // Tail streams.  The runtime maps streams onto a handful of hardware queues (four of normal priority): every stream
    // that exists beyond that -- used or not -- shares a queue with another one, and round 5's three tail streams per handle
    // put one work set's tail into the FRONT stream's hardware queue (rocprofv3: Queue_Id of stream 7 = that of stream 0):
    // its kernels then run in queue order in front of the next channelizer launch instead of beside it, and the record
    // exchange of N > 1 no longer hid under the next step (8-block rehearsal + 41 % instead of + 6 %; profiles/r6_streams.md).
    // One tail stream, as in rounds 1-4 (BTLE, and wideband 802.15.4: cfg #4 3.47 -> 3.42 ms, cfg #5 6.56 -> 6.02 ms per
    // step).  Narrowband 802.15.4 keeps three: its lanes (zb_mm) run on the tail stream beside the next segment's
    // discriminator, and consecutive segments' lanes overlap each other only on separate streams (1e9 samples: 4.05 ms
    // per step with three, 5.2 with one).
    if (const char* e = getenv("SNOUT_EXT_LAUNCH")) h->ext_launch = atoi(e) != 0;
    h->n_tails = (c.proto == SNOUT_PROTO_ZIGBEE && !h->wide) ? 3 : 1;
    if (const char* e = getenv("SNOUT_TAIL_STREAMS")) { const int v = atoi(e); if (v >= 1 && v <= 3) h->n_tails = v; }
    for (auto& s : h->slots) { rc = s.init(); if (rc) goto fail; }
    for (int k = 0; k < 3; k++) {
        if (hipEventCreate(&h->ws_free[k]) != hipSuccess) { rc = SNOUT_EHIP; goto fail; }
    }
    for (int i = 0; i < snout_rx::kHist; i++) {
        if (hipEventCreate(&h->hist_k0[i]) != hipSuccess || hipEventCreate(&h->hist_k1[i]) != hipSuccess) {
            rc = SNOUT_EHIP;
            goto fail;
        }
    }
    for (int k = 0; k < h->n_tails; k++) {
        if (hipStreamCreateWithFlags(&h->tail_streams[k], hipStreamNonBlocking) != hipSuccess) {
            set_last_error("hipStreamCreate failed");
            rc = SNOUT_EHIP;
            goto fail;
        }
    }
    {   // highest priority: the short record copy must not queue behind the next segment's blocks
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&h->copy_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
            set_last_error("hipStreamCreate failed");
            rc = SNOUT_EHIP;
            goto fail;
        }
    }
    *out = h;
    return SNOUT_OK;
fail:
    snout_rx_destroy(h);
    return rc;
}

void snout_rx_destroy(snout_rx* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    h->btle.destroy();
    h->btle2.destroy();
    h->btle3.destroy();
    for (int k = 0; k < 3; k++) if (h->tail_streams[k]) (void)hipStreamDestroy(h->tail_streams[k]);
    h->zb.destroy();
    h->zb2.destroy();
    h->zb3.destroy();
    h->pfb.destroy();
    h->d_iq.release();
    for (auto& s : h->slots) s.destroy();
    for (int k = 0; k < 3; k++) if (h->ws_free[k]) (void)hipEventDestroy(h->ws_free[k]);
    for (int i = 0; i < snout_rx::kHist; i++) {
        if (h->hist_k0[i]) (void)hipEventDestroy(h->hist_k0[i]);
        if (h->hist_k1[i]) (void)hipEventDestroy(h->hist_k1[i]);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    delete h;
}

static int check_segment(snout_rx* h, const void* iq_dev, uint64_t n_samples)
{
    if (!h || (!iq_dev && n_samples)) return SNOUT_EINVAL;
    if (n_samples >= 0xFFFF0000ull) {
        set_last_error("segment of %llu samples", (unsigned long long)n_samples);
        return SNOUT_ERANGE;
    }
    return SNOUT_OK;
}

// true if the segment is too short to hold anything (nothing enqueued)
static bool too_short(snout_rx* h, uint64_t n_samples)
{
    const uint64_t n_ch = h->wide ? h->pfb.n_out_for(n_samples) : n_samples;
    return n_ch < (h->cfg.proto == SNOUT_PROTO_BTLE ? 5u : 9u);
}

int snout_rx_submit_dev(snout_rx* h, const void* iq_dev, uint64_t n_samples,
                        uint64_t first_sample_index, void* hip_stream)
{
    return snout_rx_submit_batch_dev(h, &iq_dev, 1, n_samples, &first_sample_index, nullptr, hip_stream);
}

int snout_rx_submit_batch_dev(snout_rx* h, const void* const* iq_devs, uint32_t count, uint64_t n_samples,
                              const uint64_t* first_sample_index, const uint64_t* min_sample_index,
                              void* hip_stream)
{
    if (!h || !iq_devs || !first_sample_index || count == 0) return SNOUT_EINVAL;
    if (count > h->cfg.batch_segments) {
        set_last_error("batch of %u segments: the handle was created with batch_segments = %u", count, h->cfg.batch_segments);
        return SNOUT_EINVAL;
    }
    for (uint32_t k = 0; k < count; k++)
        if (int rc = check_segment(h, iq_devs[k], n_samples)) return rc;
    const void* iq_dev = iq_devs[0];
    const uint64_t first0 = first_sample_index[0];
    if (h->pending >= snout_rx::kSlots) {
        set_last_error("%d segments in flight: collect one first", snout_rx::kSlots);
        return SNOUT_EINVAL;
    }
    SNOUT_HIP(hipSetDevice(h->device));
    ResultSlot& s = h->slots[(h->head + h->pending) % snout_rx::kSlots];
    s.work_set = (int)(h->n_submitted++ % 3u);
    s.iq = iq_dev;
    s.n_in = n_samples;
    s.first_index = first0;
    s.segs = SegBatch{};
    s.segs.count = count;
    for (uint32_t k = 0; k < count; k++) {
        s.iq_more[k] = iq_devs[k];
        s.segs.first[k] = first_sample_index[k];
        s.segs.min_index[k] = min_sample_index ? min_sample_index[k] : 0u;
    }
    s.stream = (hipStream_t)hip_stream;
    s.n_pkts = 0;
    s.timed = false;
    if (too_short(h, n_samples)) {
        // nothing to demodulate; still channelize so the soft tap of a wideband handle is defined
        if (h->wide) {
            h->pfb.n_out = 0;
            if (n_samples) {
                if (int rc = h->pfb.run(iq_dev, n_samples, s.stream, nullptr, 0, nullptr, (int)h->cfg.sample_format)) {
                    h->n_submitted--;           // nothing was submitted: keep the work-set rotation in phase with the slot ring
                    return rc;
                }
            }
        }
        s.h_totals[0] = s.h_totals[1] = s.h_totals[2] = 0;
        s.spec_copied = 0;
        SNOUT_HIP(hipEventRecord(s.ev_copy, s.stream));
    } else {
        const int hi = (int)(h->hist_n % snout_rx::kHist);      // rotate the event pair of this slot
        std::swap(s.ev_k0, h->hist_k0[hi]);
        std::swap(s.ev_k1, h->hist_k1[hi]);
        int rc = enqueue_segment(h, s, s.stream);
        if (!rc) rc = enqueue_copy(h, s, h->spec);
        std::swap(s.ev_k0, h->hist_k0[hi]);                     // pool[hi] now holds this segment's pair
        std::swap(s.ev_k1, h->hist_k1[hi]);
        if (rc) {
            // nothing was submitted: the work-set rotation must stay in phase with the slot ring, and
            // whatever was enqueued before the failure has to be off the work set before it is reused
            h->n_submitted--;
            (void)hipDeviceSynchronize();
            return rc;
        }
        s.hist_idx = hi;
        h->hist_n++;
        s.timed = true;
    }
    h->pending++;
    return SNOUT_OK;
}

int snout_rx_poll(snout_rx* h)
{
    if (!h) return SNOUT_EINVAL;
    if (h->pending == 0) return 0;
    const hipError_t e = hipEventQuery(h->slots[h->head].ev_copy);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    set_last_error("hipEventQuery: %s", hipGetErrorString(e));
    return SNOUT_EHIP;
}

int snout_rx_collect_view(snout_rx* h, const snout_pkt** recs, uint64_t* n_out)
{
    if (!h || !recs || !n_out) return SNOUT_EINVAL;
    *recs = nullptr;
    *n_out = 0;
    if (h->pending == 0) { set_last_error("nothing submitted"); return SNOUT_EINVAL; }
    SNOUT_HIP(hipSetDevice(h->device));
    ResultSlot& s = h->slots[h->head];
    h->head = (h->head + 1) % snout_rx::kSlots;
    h->pending--;
    if (int rc = finish_slot(h, s)) return rc;
    const uint64_t np = s.n_pkts;
    if (h->cfg.flags & SNOUT_CFG_RECORDS_ON_DEVICE) {
        // the records stay in device memory (snout_rx_last_records_dev / snout_rx_pack_last_records): only the count is handed out
        note_last(h, s);
        *n_out = np;
        return SNOUT_OK;
    }
    if (np > s.spec_copied) {
        // the speculative copy was short: fetch the rest (d_out of this slot is still intact)
        std::vector<snout_pkt> keep;
        if (s.spec_copied && np > s.h_cap) keep.assign(s.h_recs, s.h_recs + s.spec_copied);
        if (int rc = s.ensure_host(np)) return rc;
        if (!keep.empty()) memcpy(s.h_recs, keep.data(), keep.size() * sizeof(snout_pkt));
        SNOUT_HIP(hipMemcpyAsync(s.h_recs + s.spec_copied, s.d_out.as<snout_pkt>() + s.spec_copied,
                                 (np - s.spec_copied) * sizeof(snout_pkt), hipMemcpyDeviceToHost,
                                 h->copy_stream));
        SNOUT_HIP(hipEventRecord(s.ev_copy, h->copy_stream));
        SNOUT_HIP(hipEventSynchronize(s.ev_copy));
    }
    h->spec = np + np / 8 + 64;        // next segment: copy this many with the totals
    note_last(h, s);
    *recs = s.h_recs;
    *n_out = np;
    return SNOUT_OK;
}

int snout_rx_last_records_dev(snout_rx* h, const snout_pkt** recs_dev, uint64_t* n_out)
{
    if (!h || !recs_dev || !n_out) return SNOUT_EINVAL;
    *recs_dev = nullptr;
    *n_out = 0;
    if (!h->last) { set_last_error("no collected segment"); return SNOUT_EINVAL; }
    *recs_dev = h->last->n_pkts ? h->last->d_out.as<snout_pkt>() : nullptr;
    *n_out = h->last->n_pkts;
    return SNOUT_OK;
}

int snout_rx_pack_last_records(snout_rx* h, void* dst_dev, uint64_t dst_cap, uint32_t width, uint64_t own_from,
                               uint64_t skip, void* longest_dev, void* hip_stream, uint64_t* n_packed)
{
    if (!h || !n_packed || (!dst_dev && dst_cap) || width < 32u || width > sizeof(snout_pkt) || (width & 15u)) return SNOUT_EINVAL;
    *n_packed = 0;
    if (!h->last) { set_last_error("no collected segment"); return SNOUT_EINVAL; }
    const uint64_t np = h->last->n_pkts > skip ? h->last->n_pkts - skip : 0u;
    const uint64_t m = np < dst_cap ? np : dst_cap;
    SNOUT_HIP(hipSetDevice(h->device));
    if (m) {
        if (int rc = launch_pack_records(h->last->d_out.as<snout_pkt>() + skip, m, dst_dev, width, own_from, longest_dev, (hipStream_t)hip_stream))
            return rc;
    }
    *n_packed = m;
    return np > dst_cap ? SNOUT_EOVERFLOW : SNOUT_OK;
}

int snout_rx_collect(snout_rx* h, snout_pkt* out, uint64_t cap, uint64_t* n_out)
{
    if (!n_out || (!out && cap)) return SNOUT_EINVAL;
    const snout_pkt* recs = nullptr;
    uint64_t np = 0;
    if (int rc = snout_rx_collect_view(h, &recs, &np)) return rc;
    *n_out = np;
    const uint64_t m = np < cap ? np : cap;
    if (m && !recs) {                   // SNOUT_CFG_RECORDS_ON_DEVICE: downloaded on request only
        SNOUT_HIP(hipMemcpy(out, h->last->d_out.p, m * sizeof(snout_pkt), hipMemcpyDeviceToHost));
    } else if (m) memcpy(out, recs, m * sizeof(snout_pkt));
    if (np > cap) { set_last_error("output capacity %llu < %llu packets", (unsigned long long)cap,
                                   (unsigned long long)np); return SNOUT_EOVERFLOW; }
    return SNOUT_OK;
}

int snout_rx_process_dev(snout_rx* h, const void* iq_dev, uint64_t n_samples,
                         uint64_t first_sample_index, void* hip_stream, snout_pkt* out, uint64_t cap,
                         uint64_t* n_out)
{
    if (!n_out || (!out && cap)) return SNOUT_EINVAL;
    *n_out = 0;
    if (int rc = check_segment(h, iq_dev, n_samples)) return rc;
    if (h->pending) { set_last_error("segments in flight: collect them first"); return SNOUT_EINVAL; }
    // synchronous form: no speculation; records are DMA'd straight into `out` when it is pinned
    const uint64_t spec_save = h->spec;
    h->spec = 0;
    struct SyncGuard { snout_rx* h; ~SyncGuard() { h->sync_call = false; } } guard{h};
    h->sync_call = true;      // also covers a rerun after a capacity overflow (finish_slot)
    int rc = snout_rx_submit_dev(h, iq_dev, n_samples, first_sample_index, hip_stream);
    h->spec = spec_save;
    if (rc) return rc;
    ResultSlot& s = h->slots[h->head];
    h->head = (h->head + 1) % snout_rx::kSlots;
    h->pending--;
    if ((rc = finish_slot(h, s))) return rc;
    uint64_t np = s.n_pkts;
    *n_out = np;
    note_last(h, s);
    if (np > cap) { set_last_error("output capacity %llu < %llu packets", (unsigned long long)cap,
                                   (unsigned long long)np); rc = SNOUT_EOVERFLOW; np = cap; }
    if (np) {
        snout_pkt* dst = out;
        const bool direct = host_is_pinned(out);
        if (!direct) { if (int r2 = s.ensure_host(np)) return r2; dst = s.h_recs; }
        SNOUT_HIP(hipMemcpyAsync(dst, s.d_out.p, np * sizeof(snout_pkt), hipMemcpyDeviceToHost,
                                 h->copy_stream));
        SNOUT_HIP(hipEventRecord(s.ev_copy, h->copy_stream));
        SNOUT_HIP(hipEventSynchronize(s.ev_copy));
        if (!direct) memcpy(out, s.h_recs, np * sizeof(snout_pkt));
    }
    return rc;
}

int snout_rx_process(snout_rx* h, const void* iq_host, uint64_t n_samples,
                     uint64_t first_sample_index, snout_pkt* out, uint64_t cap, uint64_t* n_out)
{
    if (!h || !n_out || (!iq_host && n_samples)) return SNOUT_EINVAL;
    *n_out = 0;
    SNOUT_HIP(hipSetDevice(h->device));
    if (n_samples) {
        const uint64_t bytes = n_samples * sample_bytes(h->cfg.sample_format);
        if (int rc = h->d_iq.ensure(bytes)) return rc;
        SNOUT_HIP(hipMemcpy(h->d_iq.p, iq_host, bytes, hipMemcpyHostToDevice));
    }
    return snout_rx_process_dev(h, n_samples ? h->d_iq.p : nullptr, n_samples,
                                first_sample_index, nullptr, out, cap, n_out);
}

void* snout_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        set_last_error("hipHostMalloc(%zu) failed", bytes);
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void snout_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int snout_rx_profile(snout_rx* h, snout_rx_prof* out)
{
    if (!h || !out) return SNOUT_EINVAL;
    memset(out, 0, sizeof(*out));
    if (!h->last || !h->last->timed) { set_last_error("no processed segment to profile"); return SNOUT_EINVAL; }
    ResultSlot& s = *h->last;
    SNOUT_HIP(hipEventElapsedTime(&out->ms_total, h->hist_k0[s.hist_idx], s.ev_copy));
    SNOUT_HIP(hipEventElapsedTime(&out->ms_dominant, h->hist_k0[s.hist_idx], h->hist_k1[s.hist_idx]));
    out->bytes_algorithmic = (uint64_t)sample_bytes(h->cfg.sample_format) * h->last_n + 160ull * h->last_pkts;
    out->n_hits = s.h_totals[0];
    out->dominant_launches = 1;
    if (h->wide) {
        // the kernel PfbCtx::run_batch actually launched
        switch (h->pfb.last_kernel) {
            case PfbCtx::kKernelMfma: snprintf(out->dominant_name, sizeof(out->dominant_name), "pfb_mfma<%u>", h->pfb.M); break;
            case PfbCtx::kKernelValu: snprintf(out->dominant_name, sizeof(out->dominant_name), "pfb_channelize<%u>", h->pfb.M); break;
            default: snprintf(out->dominant_name, sizeof(out->dominant_name), "pfb_spec%u", h->pfb.M); break;
        }
    } else if (h->cfg.proto == SNOUT_PROTO_ZIGBEE) {
        out->dominant_launches = 2;
        snprintf(out->dominant_name, sizeof(out->dominant_name), "zb_discrim..zb_walk");
    } else {
        snprintf(out->dominant_name, sizeof(out->dominant_name), "btle_demod_corr");
    }
    return SNOUT_OK;
}

int snout_rx_profile_history(snout_rx* h, float* ms, uint32_t cap, uint32_t* n_out)
{
    if (!h || !n_out || (!ms && cap)) return SNOUT_EINVAL;
    *n_out = 0;
    if (h->pending) { set_last_error("segments in flight"); return SNOUT_EINVAL; }
    const uint64_t have = h->hist_n < (uint64_t)snout_rx::kHist ? h->hist_n : (uint64_t)snout_rx::kHist;
    uint32_t k = 0;
    for (uint64_t i = h->hist_n - have; i < h->hist_n && k < cap; i++, k++) {   // oldest first
        const int hi = (int)(i % snout_rx::kHist);
        SNOUT_HIP(hipEventElapsedTime(&ms[k], h->hist_k0[hi], h->hist_k1[hi]));
    }
    *n_out = k;
    return SNOUT_OK;
}

int snout_rx_soft(snout_rx* h, uint32_t stage, uint32_t channel_slot, float* out, uint64_t cap,
                  uint64_t* n_out)
{
    if (!h || !n_out) return SNOUT_EINVAL;
    *n_out = 0;
    if (h->pending) { set_last_error("segments in flight"); return SNOUT_EINVAL; }
    SNOUT_HIP(hipSetDevice(h->device));
    SNOUT_HIP(hipDeviceSynchronize());
    if (stage == SNOUT_STAGE_CHAN_IQ && h->wide) {
        if (channel_slot >= h->pfb.M) return SNOUT_EINVAL;
        const uint64_t nf = 2ull * h->pfb.n_out;
        const uint64_t m = nf < cap ? nf : cap;
        if (m) SNOUT_HIP(hipMemcpy(out, h->pfb.d_y.as<float>() + 2ull * channel_slot * h->pfb.y_stride,
                                   m * 4u, hipMemcpyDeviceToHost));
        *n_out = nf;
        return nf > cap ? SNOUT_EOVERFLOW : SNOUT_OK;
    }
    if (!h->last) { set_last_error("no processed segment: nothing to tap"); return SNOUT_EINVAL; }
    if (stage == SNOUT_STAGE_BTLE_BITS && h->cfg.proto == SNOUT_PROTO_BTLE) {
        BtleCtx& b = btle_of(h, *h->last);
        if (channel_slot >= b.n_slots || h->last_nch < 5) return SNOUT_EINVAL;
        const uint64_t nb = h->last_nch - 4;
        const uint64_t words = (uint64_t)b.n_chunks * kChunkIters * 4u;
        std::vector<uint64_t> pl(words);
        SNOUT_HIP(hipMemcpy(pl.data(), b.d_planes.as<uint64_t>() + channel_slot * b.plane_stride,
                            words * 8u, hipMemcpyDeviceToHost));
        *n_out = nb;
        const uint64_t m = nb < cap ? nb : cap;
        for (uint64_t n = 0; n < m; n++) {
            const uint64_t g = n >> 8, l = (n & 255u) >> 2, j = n & 3u;
            out[n] = (float)((pl[g * 4u + j] >> l) & 1ull);
        }
        return nb > cap ? SNOUT_EOVERFLOW : SNOUT_OK;
    }
    if (h->cfg.proto == SNOUT_PROTO_ZIGBEE && stage >= SNOUT_STAGE_ZB_DISCRIM &&
        stage <= SNOUT_STAGE_ZB_CHIPS) {
        if (h->last_nch < 9) return SNOUT_EINVAL;
        return (h->last ? zb_of(h, *h->last) : h->zb).soft(stage, channel_slot, h->last_nch, out, cap, n_out);
    }
    set_last_error("stage %u not available for this configuration", stage);
    return SNOUT_EINVAL;
}

}  // extern "C"
