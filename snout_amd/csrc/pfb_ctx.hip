// pfb_ctx.hip — host side of the polyphase channelizer (PfbCtx): sizes a launch, deals the tiles of its segments into work
// items and launches the shipped kernels of pfb_spec.hip.  The channelizer replaces the one-channel-at-a-time hop of the
// reference (snout/core/radio.py:415, snout/util/btle.py:62; SURVEY.md §8d cfg #3 / #4).
//
// The kernels kept as A/B partners (pfb.hip: round 2's lock-step kernel; pfb_mfma.hip: the FIR on the f32 matrix pipe) are
// compiled into libsnout_rx_ab.so only (make ab, -DSNOUT_AB_KERNELS): there SNOUT_PFB_IMPL / SNOUT_PFB_SMALL40 /
// SNOUT_PFB_SMALL16 select them; the product library carries ONE channelizer and ignores nothing silently (an unknown
// SNOUT_PFB_IMPL value, or any value in the product library, fails the handle's creation).
#include "common.h"
#include "iq_fmt.h"
#include "pfb_tables.inc"

namespace snout {

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

#ifdef SNOUT_AB_KERNELS
uint32_t pfb_valu_tile(uint32_t M);
int pfb_valu_launch(uint32_t M, int mode, int fmt, uint32_t grid, hipStream_t st, const PfbMfArgs& a, const float* twM, const float* tw5);
#endif

// Zero of the discriminator rows behind the last channelizer tile (zb_mm reads whole lane tiles), all segments and
// channels of a batch in ONE launch: one hipMemset2DAsync per segment was one small fill kernel per segment, each waiting
// for a CU between the persistent launches.
__global__ __launch_bounds__(256) void zb_zero_tails(float* __restrict__ d, uint64_t d_seg, uint64_t d_stride, uint64_t done, uint32_t rows)
{
    const uint32_t row = blockIdx.y % rows, seg = blockIdx.y / rows;
    float* p = d + (uint64_t)seg * d_seg + (uint64_t)row * d_stride;
    for (uint64_t i = done + (uint64_t)blockIdx.x * 256u + threadIdx.x; i < d_stride; i += (uint64_t)gridDim.x * 256u) p[i] = 0.0f;
}

// The lanes read the discriminator rows in whole tiles: what lies behind the last channelizer tile (index `done` on) must be
// zero.  The channelizer rewrites [0, done) of every row each time, so the tails stay zero from one segment to the next
// unless this one is SHORTER than something written since they were zeroed (PfbZbTarget::dirty_to, kept by the owner of
// the buffer): then -- and the first time -- one fill launch over all rows of the buffer.
void PfbCtx::zero_tails(const PfbZbTarget& zt, uint64_t done, uint32_t count, uint64_t d_seg, hipStream_t st)
{
    const bool tracked = zt.dirty_to != nullptr && (count <= 1u || d_seg == (uint64_t)M * zt.d_stride);
    if (tracked && done >= *zt.dirty_to) { *zt.dirty_to = done; return; }
    if (done < zt.d_stride) {
        const uint32_t rows = tracked ? std::max(zt.zero_rows, M * count) : M * count;
        hipLaunchKernelGGL(zb_zero_tails, dim3((uint32_t)std::min<uint64_t>(cdiv(zt.d_stride - done, 256), 16u), rows), dim3(256), 0, st,
                           zt.d, tracked ? (uint64_t)M * zt.d_stride : d_seg, zt.d_stride, done, M);
    }
    if (zt.dirty_to) *zt.dirty_to = tracked ? done : ~0ull;
}

int PfbCtx::init(uint32_t M_, uint32_t n_cus_, uint32_t reserved_)
{
    M = M_;
    n_cus = n_cus_ ? n_cus_ : 256u;
    reserved_cus = reserved_ < n_cus ? reserved_ : n_cus - 1u;
    if (M != 40 && M != 16) { set_last_error("channelizer supports M = 40 or 16, not %u", M); return SNOUT_EINVAL; }
    if (const char* e = getenv("SNOUT_PFB_BLOCKS")) grid_blocks = (uint32_t)atoi(e);
    if (const char* e = getenv("SNOUT_PFB_MIN_ITEM")) min_item_tiles = std::max(1u, (uint32_t)atoi(e));
    if (const char* e = getenv("SNOUT_PFB_IMPL")) {
        // the kernel actually launched is recorded per launch (last_kernel); an unknown name is an error, not the default
        static const char* const names[] = {"valu", "mfma", "spec16", "spec12", "spec"};
        int found = -1;
        for (int k = 0; k < 5; k++) if (strcmp(e, names[k]) == 0) found = k;
#ifdef SNOUT_AB_KERNELS
        if (found < 0) { set_last_error("SNOUT_PFB_IMPL=%s: one of valu, mfma, spec16, spec12, spec", e); return SNOUT_EINVAL; }
        impl = found;
#else
        if (found != 4) {
            set_last_error("SNOUT_PFB_IMPL=%s: this library carries the shipped channelizer (spec) only; the A/B kernels are in "
                           "libsnout_rx_ab.so (make -C snout_amd/csrc ab, load it through SNOUT_RX_LIB)", e);
            return SNOUT_EINVAL;
        }
#endif
    }
#ifdef SNOUT_AB_KERNELS
    if (const char* e = getenv(M_ == 40 ? "SNOUT_PFB_SMALL40" : "SNOUT_PFB_SMALL16")) small_tiles = (uint32_t)atoi(e);
    if (int rc = d_tw.ensure(2 * M * 4)) return rc;
    if (int rc = d_tw5.ensure(10 * 4)) return rc;
    SNOUT_HIP(hipMemcpy(d_tw.p, (M == 40 ? kTw40 : kTw16), 2 * M * 4, hipMemcpyHostToDevice));
    SNOUT_HIP(hipMemcpy(d_tw5.p, kTw5, 10 * 4, hipMemcpyHostToDevice));
#endif
    if (int rc = d_proto.ensure(M * 16 * 4)) return rc;
    SNOUT_HIP(hipMemcpy(d_proto.p, (M == 40 ? kPfbProto40 : kPfbProto16), M * 16 * 4, hipMemcpyHostToDevice));
    return 0;
}

void PfbCtx::destroy()
{
    d_proto.release(); d_tw.release(); d_tw5.release(); d_y.release();
}

uint64_t PfbCtx::n_out_for(uint64_t n) const
{
    const uint64_t L = (uint64_t)M * 16u, D = M / 2u;
    return n >= L ? (n - L) / D + 1u : 0u;
}

int PfbCtx::run(const void* d_iq, uint64_t n, hipStream_t st, uint16_t* planes16, uint64_t plane_stride,
                const PfbZbTarget* zbt, int fmt)
{
    return run_batch(&d_iq, 1, n, st, planes16, plane_stride, 0, zbt, 0, 0, fmt);
}

// `count` segments of n samples each in ONE launch: segment k reads iqs[k] and writes its bit planes
// planes_seg uint16 further than segment k - 1 (fused BTLE), its discriminator rows / sub-block sums
// d_seg floats / S_seg doubles further (fused 802.15.4).
//
// One workgroup of 16 waves per CU is resident (pfb_spec.hip) and walks a contiguous range of one segment's tiles.  One
// segment: 256 ranges, one per CU.  A batch: every segment is cut into `per_seg` ranges, chosen so that the launch's
// workgroups fill whole rounds of 256 (the hardware starts the next workgroup on a CU when the one before it has finished:
// 48 segments x 16 ranges = 3 rounds; round 3 gave a segment of a batch 256 / count workgroups, 42 of them for 6 segments,
// and a launch per batch spent a quarter of its time in prologues and partly filled rounds).
int PfbCtx::run_batch(const void* const* iqs, uint32_t count, uint64_t n, hipStream_t st, uint16_t* planes16,
                      uint64_t plane_stride, uint64_t planes_seg, const PfbZbTarget* zbt, uint64_t d_seg,
                      uint64_t S_seg, int fmt)
{
    if (count == 0 || count > kMaxBatch || (count > 1 && !planes16 && !zbt)) {
        set_last_error("channelizer batch of %u segments (1..%u; more than one only in the fused modes)", count, kMaxBatch);
        return SNOUT_EINVAL;
    }
    PfbSegs segs{};
    for (uint32_t k = 0; k < count; k++) segs.x[k] = iqs[k];
    segs.planes_seg = planes_seg; segs.d_seg = d_seg; segs.S_seg = S_seg;
    n_out = n_out_for(n);
    y_stride = (n_out + 64 + 1) & ~1ull;      // even: channel rows stay 16-byte aligned
    if (!planes16 && !zbt) { if (int rc = d_y.ensure(y_stride * M * 8u)) return rc; }
    PfbZbOut zb{};
    if (zbt) zb = PfbZbOut{zbt->d, zbt->d_stride, zbt->S, zbt->nsb, zbt->atan_tab, zbt->iir_w};
    if (n_out == 0) return 0;
    const int mode = planes16 ? 1 : (zbt ? 2 : 0);
    const uint32_t wgs_all = grid_blocks ? grid_blocks : n_cus - reserved_cus;     // one 16-wave workgroup per CU it may use
#ifdef SNOUT_AB_KERNELS
    const bool small = (uint64_t)cdiv(n_out, 128u) * count < small_tiles;      // tiles of the whole launch
    const bool use_valu = impl == 0 || small;
    const bool use_mfma = !use_valu && M == 40 && (impl == 1 || impl == 2);
    if (use_valu || use_mfma) {
        // static ranges: wgs_per_seg workgroups per segment, one range each
        const uint32_t T = use_valu ? pfb_valu_tile(M) : 128u;
        const uint32_t n_tiles = cdiv(n_out, T);
        const uint32_t resident = use_mfma ? 256u : (M == 40 ? 512u : (zbt ? 768u : 1024u));    // what is resident at once
        const uint32_t wgs = std::max(1u, (grid_blocks ? grid_blocks : resident) / count);
        const uint32_t tpw = cdiv(n_tiles, wgs), nwg = cdiv(n_tiles, tpw);
        segs.wgs_per_seg = nwg;
        if (zbt) zero_tails(*zbt, (uint64_t)n_tiles * T, count, d_seg, st);
        PfbMfArgs a{segs, n, n_out, n_tiles, tpw, d_proto.as<float>(), mode == 0 ? d_y.as<float2>() : nullptr, y_stride,
                    planes16, plane_stride, zb};
        last_kernel = use_valu ? kKernelValu : kKernelMfma;
        if (use_valu) return pfb_valu_launch(M, mode, fmt, nwg * count, st, a, d_tw.as<float>(), d_tw5.as<float>());
        return pfb_mfma_launch(40, planes16 != nullptr, fmt, impl == 1 ? 0 : 1, nwg * count, st, a);
    }
    const int waves = (M == 40 && impl == 3) ? 12 : 16;
#else
    const int waves = 16;
#endif
    const uint32_t T = pfb_spec_tile(M);
    const uint32_t n_tiles = cdiv(n_out, T);
    // ranges per segment: the split with the shortest makespan in tile times, rounds x (tiles of a range + ~5 for its
    // prologue and pipeline drain)
    uint32_t per_seg = std::max(1u, wgs_all / count);
    if (count > 1) {
        uint64_t best = ~0ull;
        // with CUs reserved for other streams the launch must not exceed one round: a second round's workgroups would
        // start on whatever CU is free, the reserved ones included
        const uint32_t p_hi = std::max(1u, std::min(n_tiles / min_item_tiles, (reserved_cus ? wgs_all : 4096u) / count));
        for (uint32_t p = 1; p <= p_hi; p++) {
            const uint64_t span = (uint64_t)cdiv((uint64_t)count * p, wgs_all) * (cdiv(n_tiles, p) + 5u);
            if (span < best) { best = span; per_seg = p; }
        }
    }
    const uint32_t tpw = cdiv(n_tiles, per_seg);
    per_seg = cdiv(n_tiles, tpw);                           // every range non-empty
    segs.wgs_per_seg = per_seg;
    const uint32_t grid = per_seg * count;
    if (zbt) zero_tails(*zbt, (uint64_t)n_tiles * T, count, d_seg, st);
    PfbMfArgs a{segs, n, n_out, n_tiles, tpw, d_proto.as<float>(), mode == 0 ? d_y.as<float2>() : nullptr,
                y_stride, planes16, plane_stride, zb};
    last_kernel = waves == 12 ? kKernelSpec12 : kKernelSpec;
    hipEvent_t e0 = ev_start, e1 = ev_stop;
    ev_start = ev_stop = nullptr;
    return pfb_spec_launch(M, mode, fmt, waves, grid, st, a, e0, e1);
}

}  // namespace snout
