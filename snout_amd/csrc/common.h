// common.h — shared declarations of libsnout_rx.so internals (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <algorithm>
#include <string.h>
#include "../../include/snout_rx.h"

namespace snout {

// ---- error plumbing -----------------------------------------------------------------------
void set_last_error(const char* fmt, ...);
#define SNOUT_HIP(expr)                                                                   \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::snout::set_last_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                    __FILE__, __LINE__);                                  \
            return SNOUT_EHIP;                                                            \
        }                                                                                 \
    } while (0)

// ---- geometry of the BTLE streaming kernel ----------------------------------------------------
constexpr int kWave = 64;
constexpr int kIterSamples = 256;   // samples one wave consumes per iteration (4 per lane)
constexpr int kChunkIters = 64;     // iterations per chunk -> 16384 samples per wave
constexpr int kChunkSamples = kIterSamples * kChunkIters;
constexpr uint32_t kMaxTiles = 65536;                     // tile-sum table entries (1024 items each)
constexpr size_t kTotalsBytes = (16 + 3 * kMaxTiles) * 4;
constexpr uint32_t kScanBlock = 256, kScanItems = 4, kScanTile = kScanBlock * kScanItems;  // 1024
constexpr int kSoftCap = 1 << 17;   // floats per soft-tap buffer (tests)
constexpr int kBtleMaxSpan = 128 + 32 * (2 + 37 + 3);   // AA start -> end of CRC, samples

// Candidate produced by the decode kernel, consumed by resolve/emit.
struct BtleCand {
    uint32_t n_hit;      // in-segment sample index of the last access-address bit
    uint32_t next;       // sample index at which the sequential search would resume
    uint16_t slot;       // channel slot
    uint8_t  status;     // 0 ok, 1 header truncated, 2 bad length, 3 payload truncated
    uint8_t  accept;     // set by resolve
};
static_assert(sizeof(BtleCand) == 12, "BtleCand layout");

bool host_is_pinned(const void* p);

// A growable device buffer.
struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};



// One in-flight result: device-side record buffer + totals, pinned host mirrors, events.
// Two of them let the record D2H of segment i overlap the kernels of segment i+1.
// A batch of capture segments of equal length handled as ONE call (snout_rx_submit_batch_dev): segment k
// owns the channel slots [k slots_per_seg, (k + 1) slots_per_seg).  By value in kernel arguments.
constexpr uint32_t kMaxBatch = 64;
struct SegBatch {
    uint64_t first[kMaxBatch];          // channel-sample index of the segment's first sample
    uint64_t min_index[kMaxBatch];      // records that start before it are dropped (0: keep all)
    uint32_t slots_per_seg, count;
};

struct ResultSlot {
    DevBuf d_out, d_totals;              // records; totals: [0] candidates/lanes [1] packets [2] overflow
    uint32_t* h_totals = nullptr;        // pinned, 16 u32
    snout_pkt* h_recs = nullptr;         // pinned staging for records
    uint64_t h_cap = 0;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_front = nullptr,
               ev_compute = nullptr, ev_copy = nullptr;
    // bookkeeping of the submitted segment
    const void* iq = nullptr;           // cf32 / sc8 / sc16 as the handle is configured (iq_fmt.h)
    const void* iq_more[kMaxBatch] = {};   // segments 1.. of a batch (iq is segment 0)
    SegBatch segs = {};
    uint64_t n_in = 0, first_index = 0, spec_copied = 0, n_pkts = 0;
    hipStream_t stream = nullptr;
    bool timed = false;
    int hist_idx = 0;                    // event pair of the handle's pool that timed this segment
    int work_set = 0;                    // which BTLE work set (planes, lists, candidates) it runs on
    int init();
    void destroy();
    int ensure_host(uint64_t recs);
};

// BTLE pipeline state shared by the narrowband and the channelized front ends (btle.hip).
struct BtleCtx {
    uint32_t n_slots = 0, aa = 0, crc_init = 0, max_hits_cfg = 0, max_cand_grown = 0;     // n_slots: slots of the current call
    uint32_t seg_slots = 0, batch_cap = 1;          // slots per segment; segments a call may hold
    uint32_t n_chunks = 0, max_cand = 0, last_n_cand = 0;
    uint32_t hit_cap = 64;          // candidate hits one 16384-sample chunk can hold (grows on overflow)
    bool overflow_chunk = false, overflow_cand = false;
    uint32_t variant = 1;           // btle_demod_corr prefetch depth (rows in flight per wave; 1 measured best)
    uint64_t plane_stride = 0;
    DevBuf d_planes, d_chunk_cnt, d_chunk_hits, d_hit_n, d_hit_slot, d_cand, d_stage, d_accept,
        d_whiten, d_slot_channel;

    int init(uint32_t seg_slots, const uint16_t* slot_channel, uint32_t aa, uint32_t crc_init,
             uint32_t max_hits, uint32_t batch_cap = 1);
    void destroy();
    int reserve(uint64_t n_channel_samples, uint32_t segs = 1);
    int launch_demod_corr(const void* d_iq, uint64_t n, uint64_t iq_stride, hipStream_t st,
                          ResultSlot* timing, int fmt = 0);
    int launch_corr_planes(uint64_t n, hipStream_t st, hipEvent_t ev_stop = nullptr);     // ev_stop: bound to the kernel's dispatch
    // hit lists -> ordered records in s.d_out, totals in s.d_totals (no host sync)
    int enqueue_tail(uint64_t n, const SegBatch& segs, hipStream_t st, ResultSlot& s);
    // inspect the totals of a finished slot; true if capacity was exceeded (and grows it)
    bool check_overflow(const ResultSlot& s);
};


// records.hip: the first `width` bytes of n 160-byte records -> dst (n x width bytes); records whose
// sample_index is below own_from get sample_index = 2^62 ("disowned": the gather's sort drops them)
// longest_dev (optional): a device uint64 that receives, by atomicMax, the largest snout_pkt.len packed
int launch_pack_records(const snout_pkt* src, uint64_t n, void* dst, uint32_t width, uint64_t own_from, void* longest_dev, hipStream_t st);

void launch_tile_reduce(const uint32_t* in, const uint32_t* n_ptr, uint32_t n_fixed, uint32_t n_limit,
                        uint32_t clamp, uint32_t* tile_sums, uint32_t* tile_over, uint32_t n_tiles,
                        hipStream_t st);

// Polyphase channelizer (pfb.hip): wideband cf32 -> [M][n_out] channel IQ at 2 fs/M.
// Fused 802.15.4 mode of the M = 16 channelizer: where the discriminator rows and IIR sub-block sums go.
struct PfbZbTarget {
    float* d;
    uint64_t d_stride;
    double* S;
    uint64_t nsb;
    const float* atan_tab;
    const double* iir_w;
    // Host-side bookkeeping of the rows' tails (ZbCtx::tails_*): the lanes read whole tiles, so what lies behind the last
    // channelizer tile must be zero.  *dirty_to = the index from which every row of the buffer is known to be zero
    // (~0: unknown); a launch that writes [0, done) with done >= *dirty_to leaves the tails as they are -- no fill
    // kernel in front of the channelizer of every segment (round 5).  zero_rows: all rows of the buffer.
    uint64_t* dirty_to = nullptr;
    uint32_t zero_rows = 0;
};

// The segments of one channelizer launch (a batch of equal-length capture segments,
// snout_rx_submit_batch_dev): the grid is `wgs_per_seg` workgroups per segment; segment k reads x[k] and
// writes k "seg" strides further.
struct PfbSegs {
    const void* x[kMaxBatch];
    uint32_t wgs_per_seg;
    uint64_t planes_seg;        // uint16 elements between the bit planes of consecutive segments
    uint64_t d_seg, S_seg;      // floats / doubles between their discriminator rows / sub-block sums
};

// Where the fused M = 16 (802.15.4) epilogue writes: discriminator rows and IIR sub-block sums of the Zigbee
// context, plus the tables its arithmetic needs.
struct PfbZbOut {
    float* d;
    uint64_t d_stride;
    double* S;
    uint64_t nsb;
    const float* atan_tab;
    const double* iir_w;
};

// Arguments of the specialised-wave channelizer kernels (pfb_spec.hip, pfb_mfma.hip), by value.
struct PfbMfArgs {
    PfbSegs segs;
    uint64_t n, n_out;
    uint32_t n_tiles, tiles_per_wg;
    const float* proto;
    float2* y;
    uint64_t y_stride;
    uint16_t* planes16;
    uint64_t plane_stride;
    PfbZbOut zb;
};
// M = 40; btle: hard bits into the planes, else channel IQ into y
int pfb_mfma_launch(uint32_t M, bool btle, int fmt, int impl, uint32_t grid, hipStream_t st, const PfbMfArgs& a);   // impl: 0 MFMA FIR, 1 VALU FIR

// pfb_spec.hip: one workgroup of specialised waves per CU (FIR + staging | FFT in registers), M = 40 and 16
uint32_t pfb_spec_tile(uint32_t M);      // output times per tile of the pfb_spec kernel for M channels
int pfb_spec_launch(uint32_t M, int mode, int fmt, int waves, uint32_t grid, hipStream_t st, const PfbMfArgs& a, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);   // mode: 0 channel IQ, 1 BTLE planes, 2 802.15.4 rows; waves: 12 or 16 per workgroup (M = 40)

struct PfbCtx {
    uint32_t M = 0;
    uint64_t n_out = 0, y_stride = 0;
    uint32_t grid_blocks = 0;        // persistent grid; 0 = what is RESIDENT at once (see PfbCtx::run), else SNOUT_PFB_BLOCKS
    uint32_t n_cus = 256, reserved_cus = 0;     // the device's compute units; those the grid leaves free (cfg.reserved_cus)
    enum { kKernelSpec = 0, kKernelSpec12, kKernelMfma, kKernelValu };
    int last_kernel = kKernelSpec;   // the kernel the last launch actually ran (profile names)
    uint32_t small_tiles = 0;        // SNOUT_PFB_SMALL40 / SNOUT_PFB_SMALL16: launches with fewer tiles than this use pfb.hip's kernels (several workgroups per CU
                                     // leave room for other streams' kernels; pfb_spec.hip's one 16-wave workgroup per CU does not)
    int impl = 4;                    // SNOUT_PFB_IMPL, M = 40 kernel: 0 valu = pfb.hip, 1 mfma / 2 spec16 = pfb_mfma.hip with its FIR on the matrix / vector pipe, 3 spec12 / 4 spec = pfb_spec.hip with 12 / 16 waves
    DevBuf d_proto, d_tw, d_tw5, d_y;
    // events of the NEXT launch (cleared by it): bound to the kernel's own dispatch (hipExtLaunchKernel) instead of being
    // recorded as barrier packets of their own between two front-end kernels
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    uint32_t min_item_tiles = 48;    // SNOUT_PFB_MIN_ITEM: fewest tiles of one workgroup's range when a batch is cut finer than one range per CU
    int init(uint32_t M, uint32_t n_cus = 256, uint32_t reserved_cus = 0);
    void destroy();
    uint64_t n_out_for(uint64_t n) const;
    // planes16 != null (M = 40): fused BTLE mode, hard bits go straight into the bit planes
    // zbt != null (M = 16): fused 802.15.4 mode, discriminator output goes straight to the Zigbee context
    void zero_tails(const PfbZbTarget& zt, uint64_t done, uint32_t count, uint64_t d_seg, hipStream_t st);
    int run(const void* d_iq, uint64_t n, hipStream_t st, uint16_t* planes16 = nullptr,
            uint64_t plane_stride = 0, const PfbZbTarget* zbt = nullptr, int fmt = 0);
    // several equal-length segments in one launch (fused modes only), outputs k "seg" strides apart
    int run_batch(const void* const* iqs, uint32_t count, uint64_t n, hipStream_t st, uint16_t* planes16,
                  uint64_t plane_stride, uint64_t planes_seg, const PfbZbTarget* zbt, uint64_t d_seg,
                  uint64_t S_seg, int fmt);
};

// Zigbee / IEEE 802.15.4 pipeline state (zigbee.hip).
struct ZbCtx {
    uint32_t n_slots = 0, threshold = 10, core = 2048, warmup = 512;      // n_slots: slots of the current call
    uint32_t seg_slots = 0, batch_cap = 1;                                 // slots per segment; segments a call may hold
    uint32_t lanes_per_slot = 0, total_lanes = 0, max_out = 0;
    uint32_t pkts_per_lane = 8;     // record slots per lane (grows on overflow)
    uint32_t n_waves = 0, nt = 0, tiles_per_slot = 0;   // waves of 64 lanes, 64-sample tiles per lane
    uint64_t stream_words = 0;                          // u64 words of one channel's chip stream
    uint64_t d_stride = 0;                              // floats per channel row of the discriminator output
    bool overflow = false;
    void set_shape(uint32_t core, uint32_t warmup);
    DevBuf d_atan, d_mmse, d_slot_channel, d_stage, d_lane_cnt, d_soft;
    // discriminator output rows, tile records, per-lane stitch inputs, candidate keys,
    // first_owned|owned|offs|tsum|slot_total, chip streams
    DevBuf d_d, d_TR, d_lane_out, d_cand, d_lane_u32, d_stream;
    uint64_t tails_dirty_to = ~0ull, tails_stride = 0; uint32_t tails_rows = 0; void* tails_ptr = nullptr;     // see PfbZbTarget
    DevBuf d_iirw, d_S, d_Lblk;               // IIR carry-in: weights, sub-block sums, block sums (folded per lane in zb_mm)
    DevBuf d_lane_end, d_snap, d_req;         // frame repair: every lane's loop at its core end, sinks busy at a seam, requests
    bool repair = true;                       // SNOUT_ZB_REPAIR=0: the lanes alone (rounds 1-4; A/B and tests only)
    uint32_t tail_prio = 3;                   // SNOUT_ZB_TAIL_PRIO: bit 0 zb_walk, bit 1 zb_repair run at s_setprio 3 (A/B)
    double d64 = 0, dcore = 0, dfirst = 0;
    uint64_t nsb = 0;

    int init(uint32_t n_slots, const uint16_t* slot_channel, uint32_t threshold, uint32_t core,
             uint32_t warmup, uint32_t batch_cap = 1);
    void destroy();
    int reserve(uint64_t n_channel_samples, uint32_t segs = 1);
    int launch_sinks(uint64_t n, const SegBatch& segs, hipStream_t st);
    // front end (discriminator, carry-in, lanes) and tail (stitch, sinks, ordered compaction into
    // s.d_out / s.d_totals); no host sync
    // d_iq == nullptr: the fused channelizer has already written d and S (see pfb_target)
    int enqueue_front(const void* d_iq, uint64_t n, uint64_t iq_stride, hipStream_t st, ResultSlot& s,
                      bool time_front, int fmt = 0);
    int enqueue_lanes(uint64_t n_channel_samples, hipStream_t st);     // IIR carry-in + zb_mm (behind enqueue_front's part)
    PfbZbTarget pfb_target(uint32_t seg = 0);
    unsigned long long* seam_masks() const;         // per lane: XOR of the two timing loops' last 48 chips before its seam
    int enqueue_tail(uint64_t n, const SegBatch& segs, hipStream_t st, ResultSlot& s, bool time_front);
    bool check_overflow(const ResultSlot& s);
    int soft(uint32_t stage_id, uint32_t lane, uint64_t n, float* out, uint64_t cap, uint64_t* n_out);
};

}  // namespace snout
