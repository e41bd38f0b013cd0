// zigbee.hip — IEEE 802.15.4 O-QPSK receive kernels for gfx950 (CDNA4, wave64).
//
// Replaces the GNU Radio receive flowgraph the reference spawns for `snout zigbee scan`
// (snout/modulations/Zigbee/hackrf/Zigbee_rx/top_block.py:52-89):
//   a4 quadrature_demod_cf(1)                       -> zb_discrim   (pointwise, HBM-bound)
//   a5 x - single_pole_iir_filter_ff(0.00016)(x)    -> zb_mm        (serial per lane, fp64 state)
//   a6 clock_recovery_mm_ff(2, .000225, .5, .03, .0002)              (serial feedback loop)
//   a7 ieee802_15_4.packet_sink(10)                                  (serial FSM)
//
// The clock-recovery loop is sequential, so parallelism comes from lanes: lane i covers the core
// [i*core, (i+1)*core) of one channel and starts `warmup` samples early (IIR state carried in
// exactly; M&M from its initial state).  Lanes only produce chips; the owned chips of consecutive
// lanes are stitched into one chip stream per channel and the packet sink runs on that bit stream
// (oracle_zigbee.c states the same rules).  One wave = 64 lanes.
//
//   zb_discrim  streaming: discriminator output d[slot][t] and the IIR sub-block sums S_j
//   zb_mm       IIR + M&M per lane (thread = lane reads its own row of d in 16-byte pieces), chips +
//               window advances as 64-bit words per 64-sample tile
//   zb_stitch / zb_offsets / zb_scatter   first owned chip per lane, stream offsets, bit stream
//   zb_walk     the sink FSM per lane on the bit stream (per-chip search, 32-chip symbol steps)
#include <type_traits>
#include "common.h"
#include "zb_discrim.h"
#include "iq_fmt.h"

namespace snout {

// RX correlator words of gr-ieee802-15-4's packet_sink (FM-domain chip words, MSB = first chip).
__constant__ uint32_t kChipMap[16] = {
    1618456172u, 1309113062u, 1826650030u, 1724778362u, 778887287u, 2061946375u, 2007919840u,
    125494990u,  529027475u,  838370585u,  320833617u,  422705285u, 1368596360u, 85537272u,
    139563807u,  2021988657u};

static const float kMmseTapsHost[129][8] = {
#include "mmse_taps.inc"
};

// a4: the discriminator arithmetic (fast_atan2f_tab) lives in zb_discrim.h, shared with the fused
// channelizer epilogue.

constexpr uint32_t kTrFields = 9;  // tile record: cw lo/hi, step codes of half A lo/hi, of half B lo/hi,
                                   // nc | nc_A << 16, cstart, ii_start

__device__ __forceinline__ uint64_t tr_index(uint32_t w, uint32_t nt, uint32_t t, uint32_t field, uint32_t row)
{
    return (((uint64_t)w * nt + t) * kTrFields + field) * 64u + row;
}

// Streaming: a wave turns 256 consecutive samples of one channel into discriminator values
// d[slot][t] (8 B read, 4 B written per sample, 16-byte stores; x[t-1] by whole-wave DPP shift, the
// first lane's from its own load); rows beyond n up to the padded stride are written as zeros so
// that every lane of zb_mm may read whole tiles.  Each 64-sample sub-block also yields S_j, the
// zero-state response of the single-pole IIR to its samples (double, fixed order), from which
// zb_iir_fold and zb_mm's prologue build every lane's initial filter state (the "IIR carry-in", see the
// oracle): four threads per sub-block sum 16 terms each in sequence, S = (P0 + P1) + (P2 + P3).
constexpr uint32_t kDiscChunks = 4;     // 1024-sample chunks per zb_discrim block

template <int FMT>
__global__ __launch_bounds__(256) void zb_discrim(const void* __restrict__ iq, uint64_t n,
                                                  uint64_t iq_stride, uint64_t d_stride, uint64_t nsb,
                                                  const float* __restrict__ atan_tab,
                                                  const double* __restrict__ iir_w,
                                                  float* __restrict__ d, double* __restrict__ S)
{
    __shared__ float tab[257];
    __shared__ double wts[64];
    __shared__ float ang_s[2][1024];
    for (uint32_t i = threadIdx.x; i < 257; i += 256) tab[i] = atan_tab[i];
    if (threadIdx.x < 64) wts[threadIdx.x] = iir_w[threadIdx.x];
    __syncthreads();
    const uint32_t slot = blockIdx.y;
    const char* x = reinterpret_cast<const char*>(iq) + (uint64_t)fmt_bytes(FMT) * slot * iq_stride;
    // A block walks kDiscChunks consecutive 1024-sample chunks, so the table fill above (a fifth of
    // the instructions of a one-chunk block) is paid once per 4096 samples.  The S_j sums of chunk c
    // are taken by wave c mod 4 from a double-buffered copy of the chunk's values while the other
    // waves go on to the next chunk.
    for (uint32_t c = 0; c < kDiscChunks; c++) {
        const uint64_t chunk = (uint64_t)blockIdx.x * kDiscChunks + c;
        if (chunk * 1024u >= d_stride) break;
        const uint64_t t0 = chunk * 1024u + 4u * threadIdx.x;      // this thread's 4 samples
        float ang[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (t0 < n) {
            float2 xs[5];
            xs[0] = t0 ? iq_sample<FMT>(x, t0 - 1u) : make_float2(0.0f, 0.0f);
            if (t0 + 3u < n) {
                iq_quad<FMT>(x, t0, &xs[1]);
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) xs[1 + k] = t0 + k < n ? iq_sample<FMT>(x, t0 + k) : make_float2(0.0f, 0.0f);
            }
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {
                const float2 a = xs[k + 1], p = xs[k];
                const float re = a.x * p.x + a.y * p.y;          // contraction is off: products round first
                const float im = a.y * p.x - a.x * p.y;
                float v = fast_atan2f_tab(im, re, tab);
                if (!(fabsf(v) <= 4.0f)) v = 0.0f;              // non-finite input: defined as 0 (as the oracle)
                ang[k] = v;                                     // (a sample past n was loaded as 0: 0/0 -> NaN -> 0)
            }
        }
        *reinterpret_cast<float4*>(&d[(uint64_t)slot * d_stride + t0]) = make_float4(ang[0], ang[1], ang[2], ang[3]);
        float* as = ang_s[c & 1u];
        *reinterpret_cast<float4*>(&as[4u * threadIdx.x]) = make_float4(ang[0], ang[1], ang[2], ang[3]);
        __syncthreads();
        if ((threadIdx.x >> 6) == (c & 3u)) {
            const uint32_t l = threadIdx.x & 63u;
            const uint32_t sb = l >> 2, part = l & 3u;          // 16 sub-blocks x 4 parts
            double acc = 0.0;
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++)
                acc = acc + wts[63u - (16u * part + k)] * (double)as[64u * sb + 16u * part + k];
            acc = acc + __shfl_down(acc, 1);                    // P0 + P1 (part 0), P2 + P3 (part 2)
            acc = acc + __shfl_down(acc, 2);                    // (P0 + P1) + (P2 + P3)
            const uint64_t j = chunk * 16u + sb;
            if (part == 0u && j < nsb) S[(uint64_t)slot * nsb + j] = acc;
        }
    }
}

// Lane block i = [s0_i, s0_{i+1}): fold its sub-block sums, L = D64 L + S_j (one thread per lane).
__global__ __launch_bounds__(256) void zb_iir_fold(const double* __restrict__ S, uint64_t nsb,
                                                   uint32_t lanes_per_slot, uint32_t total_lanes,
                                                   uint32_t core, uint32_t warmup, double d64,
                                                   double* __restrict__ Lblk)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= total_lanes) return;
    const uint32_t slot = g / lanes_per_slot, li = g % lanes_per_slot;
    const uint64_t b0 = li == 0 ? 0ull : ((uint64_t)li * core - warmup) / 64u;
    const uint64_t b1 = ((uint64_t)(li + 1) * core - warmup) / 64u;
    const double* s = S + (uint64_t)slot * nsb;
    double L = 0.0;
    uint64_t j = b0;
    for (; j + 8u <= b1 && j + 8u <= nsb; j += 8u) {        // eight loads in flight per step
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = s[j + k];
#pragma unroll
        for (int k = 0; k < 8; k++) L = d64 * L + v[k];
    }
    for (; j < b1; j++) L = d64 * L + (j < nsb ? s[j] : 0.0);
    Lblk[g] = L;
}

// ---------------------------------------------------------------------------------------------
// a5-a6: lanes (IIR + Mueller & Mueller), chips out.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kMmWaves = 4;   // waves per zb_mm workgroup
constexpr int kZRow = 40;          // z buffer of a wave: 8 samples of history + half a 64-sample tile, SAMPLE-major ([sample][lane]):
                                   // a lane's bank is its lane number whatever its window offset -- lane-major rows (odd stride 41) put
                                   // lanes at different offsets on the same banks: 537 against 177 cycles per 8-sample window at 12
                                   // waves per CU (tools/lds_unaligned_probe.hip)
// MMSE table in LDS as two float4 arrays (taps 0-3, taps 4-7 of every row): one ds_read_b128 each,
// 16-B slot = row mod 16, so the 16 lanes of a read group spread over all slots
constexpr uint32_t kMaxCand = 12;
constexpr uint32_t kSinkWarmChips = 512;    // sink warm-up on the stitched stream (ORACLE_ZB_SINK_WARM)
#ifndef SNOUT_ZB_COOP_GROUPS
#define SNOUT_ZB_COOP_GROUPS 4
#endif
constexpr uint32_t kCoopGroups = SNOUT_ZB_COOP_GROUPS;     // frames decoded per cooperative round of zb_walk (1, 2 or 4)

// What a lane hands to the stitcher (see oracle_zigbee.c "Stitching").
struct ZbLaneOut {
    uint32_t nc;            // chips produced
    uint32_t t_last;        // lane-relative key of the last chip: 128 * window start + rint(128 mu)
    uint32_t c0;            // index of the first candidate chip (window start in [core_start-3, core_start+5])
    uint32_t cand_n;
    uint64_t hist_end;      // the last 64 chips, most recent in bit 0
    uint64_t hist_cand;     // the 64 chips ending at the last candidate
};

// Where a lane's loop stands at its core end: what the frame repair (zb_repair) goes on from.
struct ZbLaneEnd {
    double lp;              // filter state before lane-relative sample rce = (core start - lane start) + core
    float mu, omega, last;  // the loop after its last step
    uint32_t ii;            // lane-relative window start of the chip it would produce next (>= rce unless the segment ended)
};

// One wave = 64 lanes, thread = lane.  Per 64-sample tile: the thread's own 64 discriminator
// samples arrive in registers (prefetched during the previous tile), the fp64 IIR turns them into
// z (LDS row of the lane: 8 samples of history + the tile), then the M&M steps whose window starts
// inside the tile run, shifting hard decisions and window advances into 64-bit words that are
// stored as the tile's record.  No sink here: chips go to the stitched stream (zb_scatter) and the
// sinks run on bits (zb_walk).
template <bool TAP>
__global__ __launch_bounds__(kMmWaves * 64) void zb_mm(
    const float* __restrict__ d, uint64_t d_stride, uint64_t n, uint32_t nt, uint32_t lanes_per_slot,
    uint32_t total_lanes, uint32_t core, uint32_t warmup, const float* __restrict__ mmse,
    const double* __restrict__ Lblk, uint32_t window, double dfirst, double dcore,
    uint32_t* __restrict__ TR, ZbLaneOut* __restrict__ lane_out, ZbLaneEnd* __restrict__ lane_end,
    uint32_t* __restrict__ cand_keys,
    float* __restrict__ soft_z, float* __restrict__ soft_chips, uint32_t soft_lane, uint32_t soft_cap,
    uint32_t* __restrict__ soft_n)
{
    // kMmWaves independent waves per workgroup share one copy of the tap table (LDS decides how many
    // waves a CU holds: 10 KB of z samples per wave + 4.1 KB of taps per workgroup)
    __shared__ float zb_all[kMmWaves * 64 * kZRow];
    __shared__ float4 tapsA[129], tapsB[129];
    const uint32_t l = threadIdx.x & 63u, w = blockIdx.x * kMmWaves + (threadIdx.x >> 6);
    float* zb = &zb_all[(threadIdx.x >> 6) * 64u * kZRow];
    for (uint32_t i = threadIdx.x; i < 129u; i += kMmWaves * 64u) {
        tapsA[i] = make_float4(mmse[i * 8u + 0u], mmse[i * 8u + 1u], mmse[i * 8u + 2u], mmse[i * 8u + 3u]);
        tapsB[i] = make_float4(mmse[i * 8u + 4u], mmse[i * 8u + 5u], mmse[i * 8u + 6u], mmse[i * 8u + 7u]);
    }
    const uint32_t g = w * 64u + l;
    const bool active = g < total_lanes;
    const uint32_t li = active ? g % lanes_per_slot : 0u;
    const uint64_t core_start = (uint64_t)li * core;
    const uint64_t s0 = core_start > warmup ? core_start - warmup : 0ull;
    const uint32_t rcs = (uint32_t)(core_start - s0);       // lane-relative core start
    const uint32_t rce = rcs + core;
    const uint32_t avail = active && n > s0 ? (uint32_t)((n - s0) < (uint64_t)(rce + 64u) ? (n - s0) : (rce + 64u)) : 0u;
    const uint32_t tb = warmup >> 6;                        // the tile that holds the stitch candidates
    __syncthreads();
    if (w * 64u >= total_lanes) return;                     // a wave past the last lane (no records of its own)

    const double alpha = 0.00016, one_minus = 1.0 - 0.00016;
    const float omega_mid = 2.0f, gain_omega = 0.000225f, gain_mu = 0.03f;
    const float omega_lim = omega_mid * 0.0002f;
    // The filter state at the lane's first sample (the "IIR carry-in"): the fold of the W = ceil(2^18 / core) lane
    // blocks before this lane, from 0 -- older samples have decayed below 2^-60, see the oracle.  (Round 5: here instead of
    // in a kernel of its own between zb_iir_fold and this one: one launch and its gap less in front of every segment's
    // lanes; ~40 dependent multiply-adds per lane.)
    double lp = 0.0;
    if (active) {
        const double* L = Lblk + (g - li);
        for (uint32_t i = li > window ? li - window : 0u; i < li; i++) lp = (i == 0u ? dfirst : dcore) * lp + L[i];
    }
    double lp_rce = lp;
    float mu = 0.5f, omega = 2.0f, last = 0.0f;
    uint32_t ii = 0, n_chips = 0, t_last = 0, c0 = 0, cand_n = 0;
    uint64_t hist = 0, hist_cand = 0;
    const bool tap = TAP && active && g == soft_lane && soft_chips != nullptr;
    float* zcol = &zb[l];                                   // sample j of this lane at zcol[64 j]
    // the 8 samples of history in front of a half are the last 8 rows of the half before it: copied inside LDS (this
    // lane's own column; a wave's LDS operations execute in order) instead of kept in 8 registers
#pragma unroll
    for (int k = 0; k < 8; k++) zcol[64 * (32 + k)] = 0.0f;

    // The lane's samples are a row of d: tile t = 64 floats at d_row + 64 t, fetched as 16-byte
    // pieces one tile ahead.  Every lane reads its own row (the 64 lanes of a load touch 64 lines,
    // each line is used by eight consecutive loads) -- the texture path has room for that beside
    // ~2 600 VALU instructions per tile, and nothing has to be transposed anywhere.
    const float* d_row = d + (uint64_t)(active ? g / lanes_per_slot : 0u) * d_stride + (active ? s0 : 0ull);
    // (half a tile in registers, the next half in flight during the half's M&M steps: 32 registers instead of 64)
    float pre[32];
    {
        const float4* tp = reinterpret_cast<const float4*>(d_row);
#pragma unroll
        for (uint32_t c4 = 0; c4 < 8u; c4++) {
            const float4 v4 = tp[c4];
            pre[4 * c4] = v4.x; pre[4 * c4 + 1] = v4.y; pre[4 * c4 + 2] = v4.z; pre[4 * c4 + 3] = v4.w;
        }
    }

    for (uint32_t tile = 0; tile < nt; tile++) {
        const uint32_t r0 = tile * 64u;
        if (r0 == rce) lp_rce = lp;     // the filter at the core end (the last tile; lane 0 of a channel: earlier)
        // The tile is consumed in two halves of 32 samples (the LDS row holds 8 + 32 samples, which
        // lets ten waves share a CU): a5, DC removal (sequential fp64 recurrence), then a6, the M&M
        // steps whose window ends inside the half.  The last tile only feeds the windows that start
        // before the core end: 8 samples are enough.
        const uint32_t nz = (tile + 1u == nt) ? 8u : 64u;
        const uint32_t cstart = n_chips, ii_start = ii;
        // window advances (step - 1 = 0..2) as 2-bit codes, one 64-bit word per half tile (a half
        // holds at most 22 chips); the latest chip sits in the low bits
        uint64_t dcode[2] = {0, 0};
        uint32_t nc = 0, nc_a = 0;
#pragma unroll
        for (uint32_t hb = 0; hb < 64u; hb += 32u) {
            if (hb < nz) {
#pragma unroll
                for (int k = 0; k < 8; k++) zcol[64 * k] = zcol[64 * (32 + k)];
#pragma unroll
                for (uint32_t q = 0; q < 32u; q += 8u) {
                    if (hb + q < nz) {
#pragma unroll
                        for (uint32_t k = 0; k < 8u; k++) {
                            const float x = pre[q + k];
                            lp = alpha * (double)x + one_minus * lp;
                            const float z = x - (float)lp;
                            zcol[64u * (8u + q + k)] = z;
                            if constexpr (TAP) { if (tap && soft_z && r0 + hb + q + k < soft_cap) soft_z[r0 + hb + q + k] = z; }
                        }
                    }
                }
            }
            // the next half's samples: in flight during this half's M&M steps
            if (hb == 0u || tile + 1u < nt) {
                const float4* tp = reinterpret_cast<const float4*>(d_row + 64u * tile + hb + 32u);
#pragma unroll
                for (uint32_t c4 = 0; c4 < 8u; c4++) {
                    const float4 v4 = tp[c4];
                    pre[4 * c4] = v4.x; pre[4 * c4 + 1] = v4.y; pre[4 * c4 + 2] = v4.z; pre[4 * c4 + 3] = v4.w;
                }
            }
            const uint32_t staged = r0 + (nz < hb + 32u ? nz : hb + 32u);
            const uint32_t hi = staged < avail ? staged : avail;
            const uint32_t zorg = r0 + hb - 8u;                      // sample held by row 0 (mod 2^32)
            // windows must start before the core end and end inside what is staged
            const uint32_t lim = hi >= 8u ? (rce < hi - 7u ? rce : hi - 7u) : 0u;
            uint64_t dc = 0;
            auto mm_steps = [&](auto cand_tag) {
                constexpr bool CAND = decltype(cand_tag)::value;
                while (ii < lim) {
                    const int imu = (int)rintf(mu * 128.0f);
                    const float4 ta = tapsA[imu], tb4 = tapsB[imu];
                    const float* wv = &zcol[64u * (ii - zorg)];           // 8 consecutive samples, 64 floats apart
                    float acc = 0.0f;
                    acc = __builtin_fmaf(ta.x, wv[64 * 7], acc);
                    acc = __builtin_fmaf(ta.y, wv[64 * 6], acc);
                    acc = __builtin_fmaf(ta.z, wv[64 * 5], acc);
                    acc = __builtin_fmaf(ta.w, wv[64 * 4], acc);
                    acc = __builtin_fmaf(tb4.x, wv[64 * 3], acc);
                    acc = __builtin_fmaf(tb4.y, wv[64 * 2], acc);
                    acc = __builtin_fmaf(tb4.z, wv[64 * 1], acc);
                    acc = __builtin_fmaf(tb4.w, wv[0], acc);
                    const float o = acc;
                    if constexpr (TAP) { if (tap && n_chips + nc < soft_cap) soft_chips[n_chips + nc] = o; }
                    hist = (hist << 1) | (o > 0.0f ? 1ull : 0ull);
                    t_last = ii * 128u + (uint32_t)imu;
                    if constexpr (CAND) {       // only the tile that holds the lane's core start records candidates
                        if (li > 0u && ii + 3u - rcs <= 8u && cand_n < kMaxCand) {
                            if (cand_n == 0u) c0 = n_chips + nc;
                            cand_keys[(size_t)g * kMaxCand + cand_n] = t_last;
                            cand_n++;
                            hist_cand = hist;
                        }
                    }
                    const float mm = (last < 0.0f ? -1.0f : 1.0f) * o - (o < 0.0f ? -1.0f : 1.0f) * last;
                    last = o;
                    omega = omega + gain_omega * mm;
                    {
                        const float x = omega - omega_mid;
                        const float c = 0.5f * (fabsf(x + omega_lim) - fabsf(x - omega_lim));
                        omega = omega_mid + c;
                    }
                    mu = mu + omega + gain_mu * mm;
                    const float fl = floorf(mu);
                    const uint32_t step = fl >= 1.0f ? (uint32_t)(int)fl : 1u;    // 1..3 for finite input
                    ii += step;
                    mu = mu - fl;
                    dc = (dc << 2) | (uint64_t)(step - 1u);
                    nc++;
            }
            };
            // The same steps with fewer instructions (the loop is issue-bound as soon as a SIMD holds two of these waves,
            // profiles/r5_repair.md section 6), for every tile but the candidates' (and the soft tap's lane):
            //   rint(128 mu) by one FMA onto 1.5 * 2^23 (mu in [0, 1): the integer is the low mantissa bits, exact);
            //   the window address carried instead of the window index; sign(o) from o's sign bit (o is never -0: the
            //   accumulator starts at +0); sign(last) carried; hard decisions in a 32-bit word per half tile (<= 22
            //   chips); the 2-bit advance codes as (dc << 2) + step in one 64-bit shift-add, the "- 1" of all codes at once.
            auto mm_fast = [&]() {
                if (!(ii < lim)) return;
                const float kMagic = 12582912.0f;                       // 1.5 * 2^23
                uint32_t woff = (ii - zorg) * 256u;                      // byte offset of the window in this lane's column
                const uint32_t wlim = (lim - zorg) * 256u;
                uint32_t h32 = 0u, step = 0u;
                uint64_t dq = 0ull;
                float tmu = kMagic;
                float sl = __uint_as_float((__float_as_uint(last) & 0x80000000u) | 0x3f800000u);
                const char* zc = reinterpret_cast<const char*>(zcol);
                do {
                    tmu = __builtin_fmaf(mu, 128.0f, kMagic);
                    const uint32_t imu = __float_as_uint(tmu) - 0x4B400000u;
                    const float4 ta = tapsA[imu], tb4 = tapsB[imu];
                    const float* wv = reinterpret_cast<const float*>(zc + woff);
                    float acc = 0.0f;
                    acc = __builtin_fmaf(ta.x, wv[64 * 7], acc);
                    acc = __builtin_fmaf(ta.y, wv[64 * 6], acc);
                    acc = __builtin_fmaf(ta.z, wv[64 * 5], acc);
                    acc = __builtin_fmaf(ta.w, wv[64 * 4], acc);
                    acc = __builtin_fmaf(tb4.x, wv[64 * 3], acc);
                    acc = __builtin_fmaf(tb4.y, wv[64 * 2], acc);
                    acc = __builtin_fmaf(tb4.z, wv[64 * 1], acc);
                    acc = __builtin_fmaf(tb4.w, wv[0], acc);
                    const float o = acc;
                    // h32 = 2 h32 + (o > 0): a compare and an add with carry
                    asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(h32) : "v"(o) : "vcc");
                    const float so = __uint_as_float((__float_as_uint(o) & 0x80000000u) | 0x3f800000u);
                    const float mm = __builtin_fmaf(-so, last, sl * o);  // sl o - so last: so last is exact, one rounding
                    last = o;
                    sl = so;
                    omega = omega + gain_omega * mm;
                    {
                        const float x = omega - omega_mid;
                        const float c = 0.5f * (fabsf(x + omega_lim) - fabsf(x - omega_lim));
                        omega = omega_mid + c;
                    }
                    mu = mu + omega + gain_mu * mm;
                    const float fl = floorf(mu);
                    step = fl >= 1.0f ? (uint32_t)(int)fl : 1u;
                    mu = mu - fl;
                    woff += step * 256u;
                    dq = (dq << 2) + (uint64_t)step;
                } while (woff < wlim);
                const uint32_t cnt = (65u - (uint32_t)__builtin_clzll(dq)) >> 1;     // every 2-bit field is 1..3
                ii = zorg + (woff >> 8);
                t_last = (ii - step) * 128u + (__float_as_uint(tmu) - 0x4B400000u);
                hist = (hist << cnt) | (uint64_t)h32;
                dc = dq - (0x5555555555555555ull & ((1ull << (2u * cnt)) - 1ull));
                nc += cnt;
            };
            if (tile == tb) mm_steps(std::true_type{});
            else if (TAP) mm_steps(std::false_type{});
            else mm_fast();
            dcode[hb >> 5] = dc;
            if (hb == 0u) nc_a = nc;
        }
        n_chips += nc;
        // ---- tile record: chip c of the tile at bit 63 - c
        {
            const uint32_t sh = 64u - nc;
            const uint64_t cw = nc ? hist << sh : 0ull;
            TR[tr_index(w, nt, tile, 0, l)] = (uint32_t)cw;
            TR[tr_index(w, nt, tile, 1, l)] = (uint32_t)(cw >> 32);
            TR[tr_index(w, nt, tile, 2, l)] = (uint32_t)dcode[0];
            TR[tr_index(w, nt, tile, 3, l)] = (uint32_t)(dcode[0] >> 32);
            TR[tr_index(w, nt, tile, 4, l)] = (uint32_t)dcode[1];
            TR[tr_index(w, nt, tile, 5, l)] = (uint32_t)(dcode[1] >> 32);
            TR[tr_index(w, nt, tile, 6, l)] = nc | (nc_a << 16);
            TR[tr_index(w, nt, tile, 7, l)] = cstart;
            TR[tr_index(w, nt, tile, 8, l)] = ii_start;
        }
    }
    if (active) {
        ZbLaneOut lo;
        lo.nc = n_chips; lo.t_last = t_last; lo.c0 = cand_n ? c0 : n_chips; lo.cand_n = cand_n;
        lo.hist_end = hist; lo.hist_cand = hist_cand;
        lane_out[g] = lo;
        ZbLaneEnd le;
        le.lp = lp_rce; le.mu = mu; le.omega = omega; le.last = last; le.ii = ii;
        lane_end[g] = le;
    }
    if constexpr (TAP) { if (tap && soft_n) *soft_n = n_chips; }
}

// ---------------------------------------------------------------------------------------------
// Stitching: first owned chip of every lane (oracle_zigbee.c "Stitching"), owned counts and their
// per-channel exclusive scan (stream offset of every lane), 1024 lanes per block.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_sum_256(uint32_t x, uint32_t* lds4)
{
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) x += __shfl_down(x, dlt);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) lds4[threadIdx.x >> 6] = x;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

__global__ __launch_bounds__(256) void zb_stitch(const ZbLaneOut* __restrict__ lane_out,
                                                 const uint32_t* __restrict__ cand_keys,
                                                 uint32_t lanes_per_slot, uint32_t core, uint32_t warmup,
                                                 uint32_t tiles_per_slot, uint32_t* __restrict__ first_owned,
                                                 uint32_t* __restrict__ owned, uint32_t* __restrict__ tsum,
                                                 unsigned long long* __restrict__ seam, uint32_t* __restrict__ req)
{
    __shared__ uint32_t lds4[4];
    const uint32_t slot = blockIdx.y, tile = blockIdx.x;
    if (slot == 0u && tile == 0u && threadIdx.x == 0u) req[0] = 0u;     // zb_walk's repair requests of this segment
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t li = tile * kScanTile + threadIdx.x * 4u + k;
        if (li >= lanes_per_slot) continue;
        const uint32_t g = slot * lanes_per_slot + li;
        const ZbLaneOut me = lane_out[g];
        uint32_t f = 0;
        // XOR of this lane's 48 chips before the seam with the last 48 of the lane before it (bit 0 = the last chip):
        // what zb_resolve flags frames by (oracle: seam[]); all ones = no comparison was made
        unsigned long long sd = li > 0u ? 0xFFFFFFFFFFFFull : 0ull;
        if (li > 0u) {
            const ZbLaneOut pv = lane_out[g - 1u];
            const uint64_t cs = (uint64_t)li * core;
            const uint64_t s0 = cs - warmup;                          // li >= 1 and warmup < core
            const uint64_t ps0 = (li - 1u) ? cs - core - warmup : 0ull;
            const uint64_t E = pv.nc ? ps0 * 128u + pv.t_last + 128u : cs * 128u;
            uint32_t f0 = me.c0;
            for (uint32_t j = 0; j < me.cand_n; j++) {
                if (s0 * 128u + cand_keys[(size_t)g * kMaxCand + j] < E) f0++; else break;
            }
            f = f0;
            if (pv.nc >= 48u && me.cand_n > 0u) {
                const uint32_t c_end = me.c0 + me.cand_n - 1u;
                const uint64_t m48 = 0xFFFFFFFFFFFFull;
                int best = -1;
#pragma unroll
                for (int k2 = 0; k2 < 3; k2++) {
                    const int sft = k2 == 0 ? 0 : (k2 == 1 ? -1 : 1);
                    const int64_t e = (int64_t)f0 + sft - 1;
                    if (e < 47 || e > (int64_t)c_end) continue;
                    const uint64_t own = (me.hist_cand >> (c_end - (uint32_t)e)) & m48;
                    const int agree = 48 - __popcll(own ^ (pv.hist_end & m48));
                    if (agree > best) { best = agree; f = (uint32_t)((int64_t)f0 + sft); sd = own ^ (pv.hist_end & m48); }
                }
            }
        }
        if (f > me.nc) f = me.nc;
        seam[g] = sd;
        first_owned[g] = f;
        owned[g] = me.nc - f;
        s += me.nc - f;
    }
    const uint32_t tot = block_sum_256(s, lds4);
    if (threadIdx.x == 0) tsum[slot * tiles_per_slot + tile] = tot;
}

__global__ __launch_bounds__(256) void zb_offsets(const uint32_t* __restrict__ owned,
                                                  const uint32_t* __restrict__ tsum, uint32_t lanes_per_slot,
                                                  uint32_t tiles_per_slot, uint32_t* __restrict__ offs,
                                                  uint32_t* __restrict__ slot_total)
{
    __shared__ uint32_t lds4[4];
    const uint32_t slot = blockIdx.y, tile = blockIdx.x;
    uint32_t x = 0;
    for (uint32_t i = threadIdx.x; i < tile; i += 256u) x += tsum[slot * tiles_per_slot + i];
    const uint32_t base = block_sum_256(x, lds4);
    uint32_t c[4], s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t li = tile * kScanTile + threadIdx.x * 4u + k;
        c[k] = li < lanes_per_slot ? owned[slot * lanes_per_slot + li] : 0u;
        s += c[k];
    }
    uint32_t inc = s;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const uint32_t t = __shfl_up(inc, dlt);
        if ((int)lane >= dlt) inc += t;
    }
    __syncthreads();
    if (lane == 63) lds4[wv] = inc;
    __syncthreads();
    uint32_t off = base + inc - s;
    for (uint32_t q = 0; q < wv; q++) off += lds4[q];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t li = tile * kScanTile + threadIdx.x * 4u + k;
        if (li < lanes_per_slot) offs[slot * lanes_per_slot + li] = off;
        off += c[k];
    }
    if (tile + 1u == tiles_per_slot && threadIdx.x == 255u) slot_total[slot] = off;
}

// Owned chips of every lane -> the channel's chip stream (chip q at bit 63 - q % 64 of word q / 64).
// Thread = lane: it appends the owned part of its tile records to a 64-bit accumulator and emits every
// completed word.  The records of 64 consecutive lanes are coalesced lines; the words are not -- every
// lane fills its own stretch of the stream, so 64 lanes storing at once touch 64 different lines with
// 8 bytes each (round 2: 4.3 x the stream's bytes in HBM writes).  A wave whose lanes lie in one channel
// therefore assembles its stretch (its lanes' stretches are adjacent: ~ core / 2 chips each) in LDS with
// ds_or_b64 and writes it out in whole lines; only the first and the last word of the stretch, which it may
// share with the neighbouring waves, go through atomicOr into the zeroed stream.  Waves that span two
// channels, and lane shapes whose stretch does not fit (wcap words per wave, 0 = never), store directly.
template <bool STAGED>
__device__ __forceinline__ void scatter_lane(const uint32_t* __restrict__ TR, uint32_t nt, uint32_t w, uint32_t row,
                                             uint32_t t0, uint32_t f, uint32_t o, unsigned long long* __restrict__ sw,
                                             unsigned long long* lds, uint32_t wb)
{
    uint32_t wi = o >> 6, fill = o & 63u;
    bool shared = fill != 0u;                   // the word being filled started before this lane
    uint64_t acc = 0;
    auto put = [&](uint32_t widx, uint64_t v, bool sh) {
        if constexpr (STAGED) { atomicOr(&lds[widx - wb], (unsigned long long)v); }
        else { if (sh) atomicOr(&sw[widx], (unsigned long long)v); else sw[widx] = v; }
    };
#pragma unroll 4
    for (uint32_t t = t0; t < nt; t++) {
        const uint32_t nc = TR[tr_index(w, nt, t, 6, row)] & 0xFFFFu;
        const uint32_t c = TR[tr_index(w, nt, t, 7, row)];   // lane chip index of the tile's first chip
        const uint64_t cw = (uint64_t)TR[tr_index(w, nt, t, 0, row)] | ((uint64_t)TR[tr_index(w, nt, t, 1, row)] << 32);
        const uint32_t lo = f > c ? f : c;
        if (lo >= c + nc) continue;                         // empty tile, or all its chips are warm-up
        const uint32_t skip = lo - c, cnt = nc - skip;
        const uint64_t bits = (cw << skip) & (~0ull << (64u - cnt));
        acc |= bits >> fill;
        if (fill + cnt >= 64u) {
            put(wi, acc, shared);
            shared = false;
            wi++;
            acc = fill ? bits << (64u - fill) : 0ull;
            fill = fill + cnt - 64u;
        } else {
            fill += cnt;
        }
    }
    if (fill != 0u && acc != 0ull) put(wi, acc, true);
}

__global__ __launch_bounds__(128) void zb_scatter(const uint32_t* __restrict__ TR, uint32_t nt,
                                                  uint32_t lanes_per_slot, uint32_t total_lanes,
                                                  uint32_t first_tile,
                                                  const uint32_t* __restrict__ first_owned,
                                                  const uint32_t* __restrict__ offs, const uint32_t* __restrict__ owned,
                                                  unsigned long long* __restrict__ stream, uint64_t stream_words,
                                                  uint32_t wcap)
{
    extern __shared__ unsigned long long scat_lds[];
    const uint32_t g = blockIdx.x * 128u + threadIdx.x;
    const uint32_t l = threadIdx.x & 63u;
    const uint32_t g0 = g - l;                                   // the wave's first lane
    if (g0 >= total_lanes) return;
    const uint32_t g1 = (g0 + 63u < total_lanes ? g0 + 63u : total_lanes - 1u);          // ... and its last
    const bool active = g < total_lanes;
    const uint32_t gs = active ? g : g1;
    const uint32_t w = gs >> 6, row = gs & 63u;
    const uint32_t f = first_owned[gs], o = offs[gs];
    unsigned long long* sw = stream + (uint64_t)(gs / lanes_per_slot) * stream_words;
    // chips before the candidate tile are never owned (a lane without predecessor owns from chip 0)
    const uint32_t t0 = (gs % lanes_per_slot) ? first_tile : 0u;
    // the wave's stretch of the stream: words wb .. we of one channel
    const uint32_t o0 = offs[g0], e1 = offs[g1] + owned[g1];
    const uint32_t wb = o0 >> 6, we = e1 ? (e1 - 1u) >> 6 : 0u;
    const bool staged = wcap != 0u && g0 / lanes_per_slot == g1 / lanes_per_slot && e1 > o0 && we - wb < wcap;
    if (!staged) {
        if (active) scatter_lane<false>(TR, nt, w, row, t0, f, o, sw, nullptr, 0u);
        return;
    }
    unsigned long long* lds = scat_lds + (size_t)(threadIdx.x >> 6) * wcap;
    const uint32_t nw = we - wb + 1u;
    for (uint32_t k = l; k < nw; k += 64u) lds[k] = 0ull;
    // (one wave's LDS instructions execute in order: no barrier between its own zeroing, OR-ing and reading)
    if (active) scatter_lane<true>(TR, nt, w, row, t0, f, o, sw, lds, wb);
    for (uint32_t k = l; k < nw; k += 64u) {
        const unsigned long long v = lds[k];
        if (k == 0u || k + 1u == nw) { if (v) atomicOr(&sw[wb + k], v); }
        else sw[wb + k] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// a7: the packet sink on the stitched chip stream, one thread per lane.
// ---------------------------------------------------------------------------------------------
struct SinkState {
    int state;          // 0 search, 1 have_sync, 2 have_header
    uint32_t shift;
    int preamble_cnt, chip_cnt, packet_byte, byte_index, packetlen, packetlen_cnt, payload_cnt;
    uint32_t lqi, lqi_cnt;
    uint32_t trigger;   // stream index of the chip that completed the first preamble match
    uint32_t c0, c1, c2;   // running FCS: after all bytes, one byte ago, two bytes ago
    uint32_t b_prev, b_last;   // the last two PSDU bytes
    uint32_t chip_err;  // the last return to search came from a symbol without a chip word within the threshold
};

// A sink that is busy with a synchronised frame where its lane's chips end (zb_walk) -- at the last symbol
// boundary before that seam -- for zb_repair to go on from with the lane's own loop (oracle_zigbee.c "Frame repair").
struct ZbSnap {
    SinkState s;
    uint32_t q_b;       // stream index of the first chip of the symbol the sink is in (q_b <= own1 < q_b + 32)
    uint32_t own1;      // the next lane's first owned chip
    uint32_t slot;      // record slot of the lane that the frame occupies
    uint32_t pad;
};

__device__ __forceinline__ void enter_search(SinkState& s)
{
    s.state = 0; s.shift = 0; s.preamble_cnt = 0; s.chip_cnt = 0; s.packet_byte = 0; s.chip_err = 0;
}

__device__ __forceinline__ uint32_t chip_dist(uint32_t shift, uint32_t word)
{
    return (uint32_t)__popc((shift & 0x7FFFFFFEu) ^ (word & 0x7FFFFFFEu));
}

__device__ __forceinline__ int decode_chips(SinkState& s, uint32_t th)
{
    // The sink takes the first of the 16 words with the smallest distance.  Words 8..15 are the
    // 30-bit complements of words 0..7 (under the mask), dist = 30 - dist, so: first minimum of
    // d[0..7] (key d*8 + i, smallest wins), first maximum of d[0..7] (key d*8 + 7-i, largest wins),
    // and the lower half wins a tie between the two.
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (uint32_t i = 0; i < 8u; i++) {
        const uint32_t d = chip_dist(s.shift, kChipMap[i]);
        const uint32_t a = d * 8u + i, b = d * 8u + (7u - i);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    const uint32_t d_lo = kmin >> 3, d_hi = 30u - (kmax >> 3);
    const bool low = d_lo <= d_hi;
    const uint32_t min_t = low ? d_lo : d_hi;
    const int best = (int)(low ? (kmin & 7u) : 8u + (7u - (kmax & 7u)));
    if (min_t < th) {
        if (s.lqi_cnt < 8) { s.lqi += 32 - min_t; s.lqi_cnt++; }
        return best;
    }
    return 0xFF;
}

// The nearest of the 16 chip words to a 32-chip window and its distance (what decode_chips
// computes, as a pure function): result = nibble | distance << 8.
__device__ __forceinline__ uint32_t nearest_word(uint32_t window)
{
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (uint32_t i = 0; i < 8u; i++) {
        const uint32_t d = chip_dist(window, kChipMap[i]);
        const uint32_t a = d * 8u + i, b = d * 8u + (7u - i);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    const uint32_t d_lo = kmin >> 3, d_hi = 30u - (kmax >> 3);
    const bool low = d_lo <= d_hi;
    return (low ? (kmin & 7u) : 8u + (7u - (kmax & 7u))) | ((low ? d_lo : d_hi) << 8);
}

// One byte of the reflected CCITT CRC (poly 0x8408), closed form of the eight shift steps.
__device__ __forceinline__ uint32_t crc16_step(uint32_t c, uint32_t byte)
{
    uint32_t x = (c ^ byte) & 0xFFu;
    x ^= (x << 4) & 0xFFu;
    return ((c >> 8) ^ (x << 8) ^ (x << 3) ^ (x >> 4)) & 0xFFFFu;
}

// Called at a symbol boundary (32 chips after the previous one; s.shift holds the symbol's chips).
// Returns true when a frame completed (caller publishes, then enter_search).  Together with the
// per-chip search in zb_walk this is gr-ieee802-15-4's per-chip state machine.
__device__ __forceinline__ bool sink_symbol(SinkState& s, uint32_t th, uint8_t* __restrict__ pkt_bytes)
{
    s.chip_cnt = 0;
    if (s.state == 0) {
        if (s.packet_byte == 0) {
            if (chip_dist(s.shift, kChipMap[0]) <= th) {
                s.preamble_cnt++;
            } else if (chip_dist(s.shift, kChipMap[7]) <= th) {
                s.packet_byte = 7 << 4;
            } else {
                enter_search(s);
            }
        } else {
            if (chip_dist(s.shift, kChipMap[10]) <= th) {
                s.state = 1; s.packetlen_cnt = 0; s.packet_byte = 0; s.byte_index = 0;
                s.lqi = 0; s.lqi_cnt = 0;
            } else {
                enter_search(s);
            }
        }
        return false;
    }
    const int c = decode_chips(s, th);
    if (c == 0xFF) { enter_search(s); s.chip_err = 1u; return false; }
    if (s.byte_index == 0) s.packet_byte = c; else s.packet_byte |= c << 4;
    s.byte_index++;
    if ((s.byte_index & 1) != 0) return false;
    if (s.state == 1) {
        const int len = s.packet_byte;
        if (len <= 127) {
            s.state = 2; s.packetlen = len; s.payload_cnt = 0; s.packet_byte = 0;
            s.byte_index = 0; s.c0 = s.c1 = s.c2 = 0;
        } else {
            enter_search(s);
        }
        return false;
    }
    if (pkt_bytes) pkt_bytes[s.packetlen_cnt] = (uint8_t)s.packet_byte;
    s.c2 = s.c1; s.c1 = s.c0; s.c0 = crc16_step(s.c0, (uint32_t)s.packet_byte);
    s.b_prev = s.b_last; s.b_last = (uint32_t)s.packet_byte;
    s.packetlen_cnt++;
    s.payload_cnt++;
    s.byte_index = 0;
    return s.payload_cnt >= s.packetlen;
}

// Full-register preamble matches of the whole stream, all chips in parallel: bit (63 - q % 64) of
// match[q / 64] = popcount((window of 32 chips ending at q) ^ symbol 0, masked) < threshold, i.e. what
// the sink's search test yields once at least 32 chips have been shifted in since it was cleared.
// Written as PAIRS {chips, match} (16 bytes per 64 chips): zb_walk's lanes read scattered words, one lane per
// line, and what bounds it at scale is the number of such requests -- one 16-byte load now brings both words.
__global__ __launch_bounds__(256) void zb_match(const unsigned long long* __restrict__ stream,
                                                uint64_t stream_words, const uint32_t* __restrict__ slot_total,
                                                uint32_t th, ulonglong2* __restrict__ pairs)
{
    const uint32_t slot = blockIdx.y;
    const uint64_t wi = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (wi >= stream_words) return;
    if (wi * 64u >= (uint64_t)slot_total[slot]) {          // behind the channel's last chip: defined (zero) words
        pairs[(uint64_t)slot * stream_words + wi] = make_ulonglong2(0ull, 0ull);
        return;
    }
    const unsigned long long* sw = stream + (uint64_t)slot * stream_words;
    const uint64_t cur = sw[wi], prev = wi ? sw[wi - 1u] : 0ull;
    const uint32_t sym0 = kChipMap[0] & 0x7FFFFFFEu;
    // the window ending at chip i = the low 32 bits of {prev, cur} >> (63 - i): one v_alignbit_b32 on 32-bit words
    // (64-bit shifts cost four times as much); (x & mask) ^ sym0 is one v_bitop3_b32
    const uint32_t v0 = (uint32_t)cur, v1 = (uint32_t)(cur >> 32), v2 = (uint32_t)prev;
    uint32_t m_hi = 0, m_lo = 0;            // bit 63 - i of the match word
#pragma unroll
    for (uint32_t i = 0; i < 64u; i++) {
        const uint32_t sft = 63u - i;
        const uint32_t x = sft >= 32u ? (sft == 32u ? v1 : __builtin_amdgcn_alignbit(v2, v1, sft - 32u))
                                      : (sft == 0u ? v0 : __builtin_amdgcn_alignbit(v1, v0, sft));
        const uint32_t dist = (uint32_t)__popc(__builtin_amdgcn_bitop3_b32(x, 0x7FFFFFFEu, sym0, 0x6A));   // (a & b) ^ c
        const uint32_t hit = dist < th ? 1u : 0u;
        if (sft >= 32u) m_hi |= hit << (sft - 32u); else m_lo |= hit << sft;
    }
    pairs[(uint64_t)slot * stream_words + wi] = make_ulonglong2(cur, ((uint64_t)m_hi << 32) | m_lo);
}

// Lane-relative window start of chip j of lane gt, from its tile records: the last tile with cstart <= j, its first
// window start, plus the window advances of the chips before j in it (2-bit codes step - 1, half A then half B, latest
// chip low).
__device__ __forceinline__ uint32_t lane_chip_pos(const uint32_t* __restrict__ TR, uint32_t nt, uint32_t gt, uint32_t j)
{
    const uint32_t w = gt >> 6, row = gt & 63u;
    uint32_t lo = 0, hi2 = nt;          // last tile with cstart <= j and nc > 0 reaching j
    while (hi2 - lo > 1u) {
        const uint32_t mid = (lo + hi2) >> 1;
        if (TR[tr_index(w, nt, mid, 7, row)] <= j) lo = mid; else hi2 = mid;
    }
    const uint32_t i = j - TR[tr_index(w, nt, lo, 7, row)];
    const uint32_t ncw = TR[tr_index(w, nt, lo, 6, row)];
    const uint32_t n_a = ncw >> 16, n_b = (ncw & 0xFFFFu) - n_a;
    const uint64_t ca = (uint64_t)TR[tr_index(w, nt, lo, 2, row)] | ((uint64_t)TR[tr_index(w, nt, lo, 3, row)] << 32);
    const uint64_t cb = (uint64_t)TR[tr_index(w, nt, lo, 4, row)] | ((uint64_t)TR[tr_index(w, nt, lo, 5, row)] << 32);
    auto code_sum = [](uint64_t word, uint32_t n_in, uint32_t first) -> uint32_t {
        // sum of the codes of the first `first` chips of a word holding n_in chips
        if (first == 0u) return 0u;
        const uint64_t x = word >> (2u * (n_in - first));
        return (uint32_t)__popcll(x & 0x5555555555555555ull) + 2u * (uint32_t)__popcll(x & 0xAAAAAAAAAAAAAAAAull);
    };
    const uint32_t ia = i < n_a ? i : n_a, ib = i < n_a ? 0u : i - n_a;
    return TR[tr_index(w, nt, lo, 8, row)] + i + code_sum(ca, n_a, ia) + code_sum(cb, n_b, ib);
}

// Sequential reader of one channel's {chips, match} pairs: the sink only moves forward, so the words around
// the cursor stay in registers and the next two are always in flight.
struct ChipReader {
    const ulonglong2* sw;
    uint32_t wi;                    // index of `cur`
    uint64_t prev, cur, n1, n2;     // chips of words wi - 1 .. wi + 2
    uint64_t mcur, mn1, mn2;        // match masks of words wi .. wi + 2
    __device__ __forceinline__ void open(const ulonglong2* s, uint32_t q)
    {
        sw = s; wi = q >> 6;
        const ulonglong2 a = sw[wi], b = sw[wi + 1u], c = sw[wi + 2u];
        prev = wi ? sw[wi - 1u].x : 0ull;
        cur = a.x; mcur = a.y; n1 = b.x; mn1 = b.y; n2 = c.x; mn2 = c.y;
    }
    __device__ __forceinline__ void seek(uint32_t q)        // q >= 64 * wi
    {
        if ((q >> 6) > wi + 2u) { open(sw, q); return; }     // far jump: four independent loads
        while (wi < (q >> 6)) {
            prev = cur; cur = n1; n1 = n2; mcur = mn1; mn1 = mn2; wi++;
            const ulonglong2 c = sw[wi + 2u];
            n2 = c.x; mn2 = c.y;
        }
    }
    // match mask of word wq >= wi: from the registers when it is one of the three held, else one load
    __device__ __forceinline__ uint64_t match_word(uint32_t wq) const
    {
        return wq == wi ? mcur : (wq == wi + 1u ? mn1 : (wq == wi + 2u ? mn2 : sw[wq].y));
    }
    __device__ __forceinline__ uint32_t bit(uint32_t q) const { return (uint32_t)(cur >> (63u - (q & 63u))) & 1u; }
    // the 32 chips ending at chip q (chip q in bit 0); q >= 31, in the current word
    __device__ __forceinline__ uint32_t window32(uint32_t q) const
    {
        const uint32_t sh = 63u - (q & 63u);
        uint64_t x = cur >> sh;
        if (sh > 32u) x |= prev << (64u - sh);
        return (uint32_t)x;
    }
};

#ifdef SNOUT_ZB_WALK_STAMPS
__device__ unsigned long long g_walk_stamps[8192 * 4];      // diagnostic build (tools/walk_stamps.py): per wave iterations, cycles, lane-iterations searching / in a symbol
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void zb_walk(
    const ulonglong2* __restrict__ pairs,
    uint64_t stream_words, const uint32_t* __restrict__ offs,
    const uint32_t* __restrict__ first_owned, const uint32_t* __restrict__ slot_total,
    const uint32_t* __restrict__ TR, uint32_t nt, uint32_t lanes_per_slot, uint32_t total_lanes,
    uint32_t core, uint32_t warmup, uint32_t th, const uint16_t* __restrict__ slot_channel,
    SegBatch segs, snout_pkt* __restrict__ stage, uint32_t K, uint32_t* __restrict__ lane_cnt,
    ZbSnap* __restrict__ snaps, uint32_t* __restrict__ req,        // req[0]: count, req[1 + i]: lane of repair request i
    uint32_t hiprio)
{
    // a few latency-bound waves that run beside the next segment's front end: issue them first -- when something waits
    // for them (two work sets); with three the front end is the critical path and they take what it leaves
    if (hiprio) __builtin_amdgcn_s_setprio(3);
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool exists = g < total_lanes;            // lanes past the end still help their wave decode
    const uint32_t gs = exists ? g : 0u;
    const uint32_t slot = gs / lanes_per_slot, li = gs % lanes_per_slot;
    const uint64_t first_index = segs.first[slot / segs.slots_per_seg];     // the slot's segment of a batch
    const ulonglong2* sw = pairs + (uint64_t)slot * stream_words;
    const uint32_t total = exists ? slot_total[slot] : 0u;
    const uint32_t own0 = offs[gs];
    const uint32_t own1 = li + 1u < lanes_per_slot ? offs[gs + 1u] : total;
    SinkState s;
    enter_search(s);
    s.byte_index = s.packetlen = s.packetlen_cnt = s.payload_cnt = 0;
    s.lqi = s.lqi_cnt = 0; s.trigger = 0; s.c0 = s.c1 = s.c2 = 0; s.b_prev = s.b_last = 0;
    uint32_t n_pk = 0, sync_q = 0;
    bool snapped = false;           // this sink was busy with a synchronised frame at its lane's seam (snaps[g] holds it there)
    const bool has_next = li + 1u < lanes_per_slot && snaps != nullptr;      // (snaps == nullptr: the repair is switched off)
    uint32_t q = own0 > kSinkWarmChips ? own0 - kSinkWarmChips : 0u;
    ChipReader rd;
    rd.open(sw, q);
    bool alive = exists && q < total;
#ifdef SNOUT_ZB_WALK_STAMPS
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_it = 0, st_srch = 0, st_sym = 0;
#endif
    while (__ballot(alive) != 0ull) {
#ifdef SNOUT_ZB_WALK_STAMPS
        st_it++;
        st_srch += __popcll(__ballot(alive && s.state == 0 && s.preamble_cnt == 0));
        st_sym += __popcll(__ballot(alive && !(s.state == 0 && s.preamble_cnt == 0)));
#endif
        bool fin = false, stepped = false, gave_up = false;
        // Frame repair: a sink that is busy with a synchronised frame where its lane's chips end leaves a snapshot of itself
        // there (at the symbol boundary at or before the seam).  A cooperative round below stops at the seam, which may be
        // exactly a byte boundary: then the crossing is seen HERE, at the top of the next iteration (found by
        // tools/fuzz_parity.py: the one-symbol path alone missed it and the frame was never asked for); otherwise in the
        // one-symbol path, inside the symbol that straddles the seam.
        if (has_next && !snapped && alive && s.state != 0 && q == own1) {
            snapped = true;
            ZbSnap sn;
            sn.s = s; sn.q_b = q; sn.own1 = own1; sn.slot = n_pk; sn.pad = 0u;
            snaps[g] = sn;
        }
        // ---- payload, cooperatively.  A lane inside the payload of a frame (state 2, at a byte
        //      boundary, at least two bytes to go) would otherwise take one symbol per iteration while
        //      the other 63 lanes of the wave wait: instead the whole wave decodes up to 64 symbols of
        //      that lane's frame at once (lane l takes symbol l), the valid prefix is consumed in whole
        //      bytes, and a symbol further than the threshold from every word is left to the
        //      one-symbol path below, which gives up the frame exactly as the sink does.
        const bool elig = alive && s.state == 2 && s.byte_index == 0 && s.packetlen - s.payload_cnt >= 2 &&
                          q + 63u < total;
        uint64_t need = __ballot(elig);
        const uint64_t pb_me = (exists && n_pk < K) ? (uint64_t)(uintptr_t)stage[(size_t)g * K + n_pk].bytes : 0ull;
        // kCoopGroups frames per round: group gi = lane / W decodes up to W = 64 / kCoopGroups symbols of ITS source lane's
        // frame (lane takes symbol lane % W).  A round stands on one load however many frames it serves, and a wave
        // that receives traffic has several lanes inside payloads at once (tools/walk_stamps.py: 9 of 64 in a symbol).
        constexpr uint32_t G = kCoopGroups, W = 64u / G;
        const uint32_t gi = lane / W, sl = lane % W;
        while (need != 0ull) {
            int src = -1;                       // my group's source lane this round
            int srcs[G];
#pragma unroll
            for (uint32_t k = 0; k < G; k++) {
                srcs[k] = -1;
                if (need != 0ull) { srcs[k] = __builtin_ctzll(need); need &= need - 1ull; }
                if (k == gi) src = srcs[k];
            }
            const bool has = src >= 0;
            const int from = has ? src : (int)lane;
            const uint32_t q0 = (uint32_t)__shfl((int)q, from), tot0 = (uint32_t)__shfl((int)total, from);
            const uint32_t rem = (uint32_t)__shfl((int)(s.packetlen - s.payload_cnt), from);
            const ulonglong2* sw0 = (const ulonglong2*)(uintptr_t)(
                (uint64_t)(uint32_t)__shfl((int)(uint32_t)(uintptr_t)sw, from) |
                ((uint64_t)(uint32_t)__shfl((int)(uint32_t)((uint64_t)(uintptr_t)sw >> 32), from) << 32));
            uint32_t S = 2u * rem < W ? 2u * rem : W;
            const uint32_t fit = (tot0 - q0) >> 5;                 // symbols whose last chip is in the stream
            S = S < fit ? S : fit;
            // a round stops at the source lane's seam: the sink crosses it symbol by symbol (one-symbol path), which is
            // where the snapshot for the frame repair is taken
            const uint32_t own1_0 = (uint32_t)__shfl((int)own1, from);
            if (q0 < own1_0) { const uint32_t upto = (own1_0 - q0) >> 5; S = S < upto ? S : upto; }
            uint32_t nwv = 0xFF00u;                                 // distance 255: invalid
            if (has && sl < S) {
                const uint32_t qe = q0 + 31u + 32u * sl;            // last chip of symbol `sl`
                const uint32_t wi2 = qe >> 6, sh = 63u - (qe & 63u);
                const uint64_t cur = sw0[wi2].x, prv = wi2 ? sw0[wi2 - 1u].x : 0ull;
                uint64_t xw = cur >> sh;
                if (sh > 32u) xw |= prv << (64u - sh);
                nwv = nearest_word((uint32_t)xw);
            }
            const uint64_t okm_all = __ballot(has && sl < S && (nwv >> 8) < th);
            const uint64_t maskW = W == 64u ? ~0ull : ((1ull << (W & 63u)) - 1ull);
            const uint64_t bad = ~(okm_all >> (W * gi)) & maskW;    // my group's symbols that failed (or lie behind S)
            const uint32_t first_bad = bad ? (uint32_t)__builtin_ctzll(bad) : W;
            const uint32_t vb = has ? (first_bad < S ? first_bad : S) >> 1 : 0u;       // whole bytes in the valid prefix
            // vb == 0: the one-symbol path takes that frame from here.  Everything below is computed alike by all lanes of
            // a group; the source lane, wherever it sits, picks its group's results up afterwards.
            const int base = (int)(W * gi);
            // link quality: the first eight symbols of a frame
            uint32_t lqi0 = (uint32_t)__shfl((int)s.lqi, from), lqc0 = (uint32_t)__shfl((int)s.lqi_cnt, from);
#pragma unroll
            for (uint32_t j = 0; j < (W < 8u ? W : 8u); j++) {
                const uint32_t dj = (uint32_t)__shfl((int)nwv, base + (int)j) >> 8;
                if (j < 2u * vb && lqc0 < 8u) { lqi0 += 32u - dj; lqc0++; }
            }
            // bytes: low nibble first
            const uint32_t nib = nwv & 15u;
            const uint32_t byte = nib | (((uint32_t)__shfl_down((int)nib, 1)) << 4);    // even lanes
            const uint32_t cnt0 = (uint32_t)__shfl(s.packetlen_cnt, from);
            uint8_t* pb0 = (uint8_t*)(uintptr_t)((uint64_t)(uint32_t)__shfl((int)(uint32_t)pb_me, from) |
                                                 ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(pb_me >> 32), from) << 32));
            if (pb0 && !(sl & 1u) && sl < 2u * vb) pb0[cnt0 + (sl >> 1)] = (uint8_t)byte;
            // running FCS and the last two bytes: sequential over the bytes
            uint32_t c0 = (uint32_t)__shfl((int)s.c0, from), c1 = (uint32_t)__shfl((int)s.c1, from), c2 = (uint32_t)__shfl((int)s.c2, from);
            uint32_t bp = (uint32_t)__shfl((int)s.b_prev, from), bl = (uint32_t)__shfl((int)s.b_last, from);
#pragma unroll
            for (uint32_t k = 0; k < W / 2u; k++) {
                const uint32_t bk = (uint32_t)__shfl((int)byte, base + (int)(2u * k));
                if (k < vb) {
                    c2 = c1; c1 = c0; c0 = crc16_step(c0, bk);
                    bp = bl; bl = bk;
                }
            }
            // hand the results to the source lanes
#pragma unroll
            for (uint32_t k = 0; k < G; k++) {
                const int gl = (int)(W * k);                         // a lane of group k
                const uint32_t r_vb = (uint32_t)__builtin_amdgcn_readlane((int)vb, gl);
                const uint32_t r_lqi = (uint32_t)__builtin_amdgcn_readlane((int)lqi0, gl), r_lqc = (uint32_t)__builtin_amdgcn_readlane((int)lqc0, gl);
                const uint32_t r_c0 = (uint32_t)__builtin_amdgcn_readlane((int)c0, gl), r_c1 = (uint32_t)__builtin_amdgcn_readlane((int)c1, gl);
                const uint32_t r_c2 = (uint32_t)__builtin_amdgcn_readlane((int)c2, gl);
                const uint32_t r_bp = (uint32_t)__builtin_amdgcn_readlane((int)bp, gl), r_bl = (uint32_t)__builtin_amdgcn_readlane((int)bl, gl);
                if (srcs[k] >= 0 && (int)lane == srcs[k] && r_vb != 0u) {
                    s.lqi = r_lqi; s.lqi_cnt = r_lqc;
                    s.c0 = r_c0; s.c1 = r_c1; s.c2 = r_c2; s.b_prev = r_bp; s.b_last = r_bl;
                    s.packetlen_cnt += (int)r_vb; s.payload_cnt += (int)r_vb;
                    s.packet_byte = (int)r_bl;
                    q += 64u * r_vb;
                    fin = s.payload_cnt >= s.packetlen;
                    stepped = true;
                }
            }
        }
        if (alive && !stepped) {
            if (s.state == 0 && s.preamble_cnt == 0) {
                // searching: one test per chip
                if (q >= own1) {
                    alive = false;                      // idle at or past the next lane's first chip
                } else {
                    // The register was cleared at chip q.  While fewer than 32 chips are in, the test sees
                    // a zero-filled register: 31 explicit tests on the 64 chips that start at q ...
                    const uint32_t lim = own1 < total ? own1 : total;
                    rd.seek(q);
                    const uint32_t bo = q & 63u;
                    const uint64_t x64 = bo ? (rd.cur << bo) | (rd.n1 >> (64u - bo)) : rd.cur;
                    const uint32_t sym0 = kChipMap[0] & 0x7FFFFFFEu;
                    uint32_t pm = 0;                    // bit k: match after chip q + k, k = 0..30
                    // With k + 1 chips in, the zero bits above them already differ from symbol 0 in
                    // popcount(sym0 >> (k + 1)) places = 17 17 16 15 15 14 13 13 13 12 11 10 10 | 9 9 8 ...: up to the
                    // flowgraph's threshold (10) the first 13 tests cannot pass and are not made.
                    auto test = [&](uint32_t k) {
                        const uint32_t reg = (uint32_t)(x64 >> (63u - k));
                        pm |= (uint32_t)((uint32_t)__popc((reg & 0x7FFFFFFEu) ^ sym0) < th) << k;
                    };
                    if (th > 10u) {
#pragma unroll
                        for (uint32_t k = 0; k < 13u; k++) test(k);
                    }
#pragma unroll
                    for (uint32_t k = 13u; k < 31u; k++) test(k);
                    bool hit = false;
                    uint32_t qh = 0;
                    if (pm) {
                        qh = q + (uint32_t)__ffs((int)pm) - 1u;
                        hit = qh < lim;
                    } else {
                        // ... then the precomputed full-register matches, a word at a time: the reader holds the
                        // masks of the cursor's word and the two behind it beside their chips
                        uint32_t qs = q + 31u;
                        while (qs < lim) {
                            const uint32_t wq = qs >> 6;
                            const uint64_t word = rd.match_word(wq);
                            const uint64_t mw = word & (~0ull >> (qs & 63u));
                            if (mw) { qh = (qs & ~63u) + (uint32_t)__clzll((long long)mw); hit = qh < lim; break; }
                            qs = (qs & ~63u) + 64u;
                        }
                    }
                    if (hit) {
                        q = qh + 1u;
                        rd.seek(q);
                        s.preamble_cnt = 1;             // chip_cnt stays 0: the boundary is 32 chips on
                        s.trigger = q - 1u;
                    } else {
                        alive = false;                  // ran to the end of the lane (or of the stream) idle
                    }
                }
            }
#ifndef SNOUT_ZB_WALK_SPLIT
            // A lane that has just found a first preamble symbol checks the next symbol in the SAME iteration: a wave
            // executes both parts every iteration anyway (some lane is always in the other state), so a refuted match --
            // nearly all of them, on noise -- costs its lane one iteration instead of two.
            if (alive && !(s.state == 0 && s.preamble_cnt == 0)) {
#else
            else {
#endif
                // inside a (candidate) frame, one symbol: jump to the next symbol boundary
                const uint32_t qb = q + 31u;            // q is the first chip of the symbol
                if (qb >= total) {
                    alive = false;                      // the stream ends inside the frame
                } else {
                    rd.seek(qb);
                    if (has_next && !snapped && s.state != 0 && q <= own1 && own1 <= qb) {
                        // busy with a synchronised frame at the seam: the sink as it stands at this symbol boundary
                        snapped = true;
                        ZbSnap sn;
                        sn.s = s; sn.q_b = q; sn.own1 = own1; sn.slot = n_pk; sn.pad = 0u;
                        snaps[g] = sn;
                    }
                    s.shift = rd.window32(qb);
                    uint8_t* pb = (uint8_t*)(uintptr_t)pb_me;
                    const int state_before = s.state;
                    fin = sink_symbol(s, th, pb);
                    if (state_before == 0 && s.state == 1) sync_q = qb;      // the chip that completed the SFD
                    gave_up = !fin && state_before != 0 && s.state == 0 && s.chip_err != 0u;
                    q = qb + 1u;
                }
            }
        }
        if (alive && q >= total) alive = false;
        if (fin || gave_up) {
            // Sinks may first match different preamble symbols but find the SFD at the same chip:
            // the frame belongs to the lane that owns that chip.
            const uint32_t len = fin ? (uint32_t)s.packetlen_cnt : 0u;
            // FCS: CRC-16 over all but the last two bytes == those two bytes (LE)
            const bool fcs_ok = fin && len >= 3u && s.c2 == (s.b_prev | (s.b_last << 8));
            // Frame repair (oracle_zigbee.c): a frame this sink was busy with at its lane's seam and then gave up at a
            // symbol, or finished with a bad FCS, is received again by the lane's own loop going on (zb_repair); until
            // then its record slot is void (trigger 0xFFFFFFFF: zb_resolve passes over it)
            const bool ask = snapped && !fcs_ok;
            if (sync_q >= own0 && sync_q < own1 && (fin || ask)) {
                if (n_pk < K) {
                    // window start of the trigger chip: chip j of the lane gt that owns it (this lane
                    // or one before it), found in that lane's tile records
                    // sample_index: the chip 319 chips before the one that completed the SFD (the first chip
                    // of a regular preamble + SFD), the same for every sink that finds this frame
                    const uint32_t rq = sync_q >= 319u ? sync_q - 319u : 0u;
                    uint32_t gt = g;
                    while (rq < offs[gt]) gt--;             // same channel: chip 0 belongs to its lane 0
                    const uint32_t rel = lane_chip_pos(TR, nt, gt, first_owned[gt] + (rq - offs[gt]));
                    const uint64_t cs = (uint64_t)(gt % lanes_per_slot) * core;
                    const uint64_t s0 = cs > warmup ? cs - warmup : 0ull;
                    snout_pkt* p = &stage[(size_t)g * K + n_pk];
                    if (fin) { for (uint32_t b = len; b < 136u; b++) p->bytes[b] = 0; }
                    p->sample_index = first_index + s0 + rel;
                    p->proto = SNOUT_PROTO_ZIGBEE;
                    p->channel = slot_channel[slot];
                    p->len = (uint16_t)len;
                    const uint32_t scaled = (s.lqi / 8u) << 3;
                    p->lqi = (uint8_t)(scaled >= 256u ? 255u : scaled);
                    p->pdu_type = 0;
                    p->flags = 0;
                    p->aux = li;
                    p->crc_ok = (uint8_t)fcs_ok;
                    // for zb_resolve (cleared by zb_emit): the chips this sink was busy with the frame
                    uint32_t* span = reinterpret_cast<uint32_t*>(&p->bytes[128]);
                    span[0] = ask ? 0xFFFFFFFFu : s.trigger;
                    span[1] = q - 1u;
                    if (ask) req[1u + atomicAdd(&req[0], 1u)] = g;
                }
                n_pk++;
            }
            if (fin) enter_search(s);
        }
    }
    if (exists) lane_cnt[g] = n_pk;
#ifdef SNOUT_ZB_WALK_STAMPS
    if (lane == 0 && (g >> 6) < 8192u) {
        g_walk_stamps[(g >> 6) * 4u + 0u] = st_it;
        g_walk_stamps[(g >> 6) * 4u + 1u] = __builtin_amdgcn_s_memtime() - st_t0;
        g_walk_stamps[(g >> 6) * 4u + 2u] = st_srch;
        g_walk_stamps[(g >> 6) * 4u + 3u] = st_sym;
    }
#endif
}
#ifdef SNOUT_ZB_WALK_STAMPS
extern "C" int snout_debug_walk_stamps(unsigned long long* out, uint32_t n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_stamps), (size_t)n * 8u, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -4;
}
#endif

// Frame repair (oracle_zigbee.c "Frame repair"): one thread per request.  The lane's own loop goes on from where it stood
// at its core end (ZbLaneEnd; the filter by its recurrence from the state saved there, so every z is the value the lane
// itself would have computed), the lane's sink from where it stood at the seam (ZbSnap + the chips of the stream between
// its last symbol boundary and the seam), until the frame completes -- then the void record slot becomes the frame -- or
// is given up.  Tile loop and arithmetic are zb_mm's; a few dozen waves per segment, on the tail stream.
__global__ __launch_bounds__(64) void zb_repair(
    const float* __restrict__ d, uint64_t d_stride, uint64_t n, uint32_t lanes_per_slot, uint32_t core, uint32_t warmup,
    const float* __restrict__ mmse, const ZbLaneEnd* __restrict__ lane_end, const ZbSnap* __restrict__ snaps,
    const uint32_t* __restrict__ req, const ulonglong2* __restrict__ pairs, uint64_t stream_words, uint32_t th,
    snout_pkt* __restrict__ stage, uint32_t K, uint32_t hiprio,
    const uint32_t* __restrict__ TR, uint32_t nt, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ first_owned,
    const uint32_t* __restrict__ owned, const uint32_t* __restrict__ slot_total)
{
    __shared__ float zb[64 * kZRow];
    __shared__ float4 tapsA[129], tapsB[129];
    __shared__ float4 stg[2][8][64];                // two halves of 32 samples per lane, as the loads deliver them
    if (hiprio) __builtin_amdgcn_s_setprio(3);      // a few long dependent chains beside the next segment's front end
    const uint32_t l = threadIdx.x;
    const uint32_t count = req[0];
    if (blockIdx.x * 64u >= count) return;
    for (uint32_t i = l; i < 129u; i += 64u) {
        tapsA[i] = make_float4(mmse[i * 8u + 0u], mmse[i * 8u + 1u], mmse[i * 8u + 2u], mmse[i * 8u + 3u]);
        tapsB[i] = make_float4(mmse[i * 8u + 4u], mmse[i * 8u + 5u], mmse[i * 8u + 6u], mmse[i * 8u + 7u]);
    }
    __syncthreads();
    const uint32_t ri = blockIdx.x * 64u + l;
    const bool active = ri < count;
    const uint32_t g = active ? req[1u + ri] : req[1u];
    const uint32_t li = g % lanes_per_slot, slot = g / lanes_per_slot;
    const uint64_t core_start = (uint64_t)li * core;
    const uint64_t s0 = core_start > warmup ? core_start - warmup : 0ull;
    const uint32_t rce = (uint32_t)(core_start - s0) + core;
    const uint32_t avail = n > s0 ? (uint32_t)((n - s0) < 0xFFFFFF00ull ? (n - s0) : 0xFFFFFF00ull) : 0u;
    const ZbLaneEnd le = lane_end[g];
    const ZbSnap sn = snaps[g];
    SinkState s = sn.s;
    snout_pkt* p = &stage[(size_t)g * K + sn.slot];
    {   // the chips of the symbol the sink is in, up to the seam
        const ulonglong2* sw = pairs + (uint64_t)slot * stream_words;
        for (uint32_t q = sn.q_b; active && q < sn.own1; q++) {
            const uint32_t bit = (uint32_t)(sw[q >> 6].x >> (63u - (q & 63u))) & 1u;
            s.shift = (s.shift << 1) | bit;
            s.chip_cnt++;
        }
    }
    // Where the lane after the next takes over again (if it owns chips): from half a chip before the window start of its
    // first owned chip the chips are that lane's, and the sink returns to the stitched stream (bounded work: one lane's
    // samples).  Keys are lane-relative, 128 per sample + rint(128 mu).
    const ulonglong2* sw = pairs + (uint64_t)slot * stream_words;
    uint32_t hand_q = 0u, hand_key = 0xFFFFFFFFu;
    if (active && li + 2u < lanes_per_slot && owned[g + 2u] != 0u && core <= (1u << 23)) {       // (lane-relative keys in 32 bits)
        const uint64_t s0_2 = (uint64_t)(li + 2u) * core - warmup;
        hand_q = offs[g + 2u];
        hand_key = (uint32_t)(s0_2 + lane_chip_pos(TR, nt, g + 2u, first_owned[g + 2u]) - s0) * 128u + 64u;
    }
    const double alpha = 0.00016, one_minus = 1.0 - 0.00016;
    const float omega_mid = 2.0f, gain_omega = 0.000225f, gain_mu = 0.03f;
    const float omega_lim = omega_mid * 0.0002f;
    double lp = le.lp;
    float mu = le.mu, omega = le.omega, last = le.last;
    uint32_t ii = le.ii, more = 0;
    bool alive = active && ii >= rce && ii + 8u <= avail;
    bool ok = false, handed = false;
    float* zcol = &zb[l];
    float zl[8];
#pragma unroll
    for (int k = 0; k < 8; k++) zl[k] = 0.0f;
    const float* d_row = d + (uint64_t)slot * d_stride + s0 + rce;      // 16-byte aligned: s0 and rce are multiples of 64
    // The samples come half a tile (32 per lane) at a time, straight into LDS (global_load_lds_dwordx4: per-lane source
    // address, destination = wave-uniform base + 16 lane; no registers), the next half in flight during a half's M&M steps.
    // So the kernel stays within 96 VGPRs, which is what a SIMD has left beside the four waves of the 802.15.4
    // channelizer's workgroup (pfb_spec<16>: 98 VGPRs, 110 KB of LDS): these waves run BESIDE the next segment's
    // channelizer instead of keeping its workgroups off their CUs until they are done (a persistent grid with a static
    // tile partition ends with its last workgroup: + 0.3 ms per step measured, profiles/r5_repair.md).
    auto fetch_half = [&](uint32_t h) {
        const float* src = d_row + 32u * h;
#pragma unroll
        for (uint32_t c4 = 0; c4 < 8u; c4++)
            __builtin_amdgcn_global_load_lds(src + 4u * c4, &stg[h & 1u][c4][0], 16, 0, 0);
    };
    fetch_half(0u);
    for (uint32_t half = 0; __ballot(alive) != 0ull; half++) {
        const uint32_t r0 = rce + (half >> 1) * 64u, hb = (half & 1u) * 32u;
        {
#pragma unroll
            for (int k = 0; k < 8; k++) zcol[64 * k] = zl[k];
#pragma unroll
            for (uint32_t c4 = 0; c4 < 8u; c4++) {
                const float4 v4 = stg[half & 1u][c4][l];
                const float xs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++) {
                    const uint32_t k = 4u * c4 + j;
                    lp = alpha * (double)xs[j] + one_minus * lp;
                    const float z = xs[j] - (float)lp;
                    zcol[64u * (8u + k)] = z;
                    if (k >= 24u) zl[k - 24u] = z;
                }
            }
            // the next half, into the other buffer, in flight during this half's M&M steps (the compiler drains the loads
            // before the first read of the staging buffer; rows are padded with zeros far beyond n)
            fetch_half(half + 1u);
            const uint32_t staged = r0 + hb + 32u;
            const uint32_t hi = staged < avail ? staged : avail;
            const uint32_t zorg = r0 + hb - 8u;
            const uint32_t lim = alive && hi >= 8u ? hi - 7u : 0u;  // windows end inside what is staged
            while (ii < lim) {
                // up to the next symbol boundary without looking at the sink: ONE exit test per chip -- the hand-back test of
                // the chip to come (its key needs rint(128 mu) of the updated mu, which the next step starts with anyway)
                // is part of the loop condition, and looked at once more in front of the loop
                float tmu = __builtin_fmaf(mu, 128.0f, 12582912.0f);
                if (ii * 128u + (__float_as_uint(tmu) - 0x4B400000u) + 128u >= hand_key) { handed = true; alive = false; break; }
                uint32_t left = 32u - (uint32_t)s.chip_cnt;
                uint32_t sh = s.shift;
                // (zb_mm's shorter forms of the same arithmetic: rint(128 mu) by one FMA, the signs from the sign bits)
                float sl = __uint_as_float((__float_as_uint(last) & 0x80000000u) | 0x3f800000u);
                uint32_t key;
                do {
                    const uint32_t imu = __float_as_uint(tmu) - 0x4B400000u;
                    const float4 ta = tapsA[imu], tb4 = tapsB[imu];
                    const float* wv = &zcol[64u * (ii - zorg)];
                    float acc = 0.0f;
                    acc = __builtin_fmaf(ta.x, wv[64 * 7], acc);
                    acc = __builtin_fmaf(ta.y, wv[64 * 6], acc);
                    acc = __builtin_fmaf(ta.z, wv[64 * 5], acc);
                    acc = __builtin_fmaf(ta.w, wv[64 * 4], acc);
                    acc = __builtin_fmaf(tb4.x, wv[64 * 3], acc);
                    acc = __builtin_fmaf(tb4.y, wv[64 * 2], acc);
                    acc = __builtin_fmaf(tb4.z, wv[64 * 1], acc);
                    acc = __builtin_fmaf(tb4.w, wv[0], acc);
                    const float o = acc;
                    const float so = __uint_as_float((__float_as_uint(o) & 0x80000000u) | 0x3f800000u);
                    const float mm = __builtin_fmaf(-so, last, sl * o);
                    last = o;
                    sl = so;
                    omega = omega + gain_omega * mm;
                    {
                        const float x = omega - omega_mid;
                        const float c = 0.5f * (fabsf(x + omega_lim) - fabsf(x - omega_lim));
                        omega = omega_mid + c;
                    }
                    mu = mu + omega + gain_mu * mm;
                    const float fl = floorf(mu);
                    ii += fl >= 1.0f ? (uint32_t)(int)fl : 1u;
                    mu = mu - fl;
                    asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(sh) : "v"(o) : "vcc");
                    left--;
                    tmu = __builtin_fmaf(mu, 128.0f, 12582912.0f);
                    key = ii * 128u + (__float_as_uint(tmu) - 0x4B400000u) + 128u;
                } while ((left != 0u) & (ii < lim) & (key < hand_key));
                const uint32_t took = 32u - (uint32_t)s.chip_cnt - left;
                more += took;
                s.shift = sh;
                s.chip_cnt += (int)took;
                if (left == 0u) {                                   // a whole symbol is in
                    const bool fin = sink_symbol(s, th, p->bytes);
                    if (fin) { ok = true; alive = false; break; }
                    if (s.state == 0) { alive = false; break; }     // given up (chip errors, or a length > 127)
                }
            }
            if (alive && ii + 8u > avail) alive = false;            // the segment ends inside the frame
        }
    }
    uint32_t end_chip = sn.own1 + more - 1u;
    if (handed) {
        // the rest of the frame from the stitched stream, a symbol at a time
        const uint32_t total = slot_total[slot];
        uint32_t q = hand_q;
        for (;;) {
            const uint32_t need = 32u - (uint32_t)s.chip_cnt;        // 1 .. 32 chips to the symbol boundary
            if (q + need > total) break;                            // the stream ends inside the frame
            const uint32_t wi = q >> 6, bo = q & 63u;
            const uint64_t w0 = sw[wi].x, w1 = sw[wi + 1u].x;       // (the stream is padded with zero words)
            const uint64_t x64 = bo ? (w0 << bo) | (w1 >> (64u - bo)) : w0;
            const uint32_t bits = (uint32_t)(x64 >> (64u - need));
            s.shift = need == 32u ? bits : ((s.shift << need) | bits);
            q += need;
            const bool fin = sink_symbol(s, th, p->bytes);
            if (fin) { ok = true; end_chip = q - 1u; break; }
            if (s.state == 0) break;
        }
    }
    if (ok) {
        const uint32_t len = (uint32_t)s.packetlen_cnt;
        for (uint32_t b = len; b < 128u; b++) p->bytes[b] = 0;
        p->len = (uint16_t)len;
        const uint32_t scaled = (s.lqi / 8u) << 3;
        p->lqi = (uint8_t)(scaled >= 256u ? 255u : scaled);
        p->crc_ok = (uint8_t)(len >= 3u && s.c2 == (s.b_prev | (s.b_last << 8)));
        p->flags = SNOUT_PKT_ZB_REPAIRED;
        uint32_t* span = reinterpret_cast<uint32_t*>(&p->bytes[128]);
        span[0] = sn.s.trigger;
        span[1] = end_chip;
    }
}

// The sequential rule over the candidate frames of all lanes (see the oracle, "Resolve"): in the
// order of their SFD chips (= lane order, then time), a frame is kept iff its trigger chip lies after
// the last chip of the frame kept before it -- the one sequential sink is busy until then, only a
// lane sink that started inside that frame can have found something there.  One thread per lane
// decides its own records: a frame ends at most kMaxBusy chips after its SFD chip, so only the
// records of the last few lanes can reach a given trigger; walking back, the thread gathers the
// records that can (transitively) matter -- those whose last chip is not before the smallest
// trigger gathered so far -- and replays the rule over them.  Marks dropped records (pdu_type = 1)
// and writes the lane's kept count.
constexpr uint32_t kMaxBusy = (2u + 2u * 128u) * 32u;       // PHR + PSDU symbols after the SFD chip
constexpr int kResolveSet = 48;

__global__ __launch_bounds__(256) void zb_resolve(snout_pkt* __restrict__ stage, const uint32_t* __restrict__ lane_cnt,
                                                  uint32_t K, uint32_t lanes_per_slot, uint32_t total_lanes,
                                                  const uint32_t* __restrict__ offs, SegBatch segs,
                                                  const unsigned long long* __restrict__ seam,
                                                  uint32_t* __restrict__ lane_kept)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= total_lanes) return;
    // records that start before this index belong to the capture segment before this one (sharded
    // scans: the overlap / pre-roll of a segment); the sink was busy with them all the same
    const uint64_t min_index = segs.min_index[g / lanes_per_slot / segs.slots_per_seg];
    const uint32_t raw = lane_cnt[g];
    if (raw > K) { lane_kept[g] = raw; return; }          // overflow: the segment is run again with more slots
    const uint32_t g_first = g - g % lanes_per_slot;        // first lane of this channel
    uint32_t kept = 0;
    for (uint32_t i = 0; i < raw; i++) {
        uint32_t trig[kResolveSet], endc[kResolveSet];
        int ns = 0;
        bool full = false;
        const uint32_t* me = reinterpret_cast<const uint32_t*>(&stage[(size_t)g * K + i].bytes[128]);
        if (me[0] == 0xFFFFFFFFu) {     // a void slot: a frame given up behind a seam that zb_repair gave up as well
            stage[(size_t)g * K + i].pdu_type = 1;
            continue;
        }
        trig[0] = me[0]; endc[0] = me[1]; ns = 1;
        uint32_t tmin = me[0];
        // walk back over the records before (g, i)
        uint32_t gl = g;
        int idx = (int)i - 1;
        for (;;) {
            if (idx < 0) {
                if (gl == g_first) break;
                gl--;
                // every frame of the lanes up to gl has its SFD chip before offs[gl + 1]
                if ((uint64_t)offs[gl + 1u] + kMaxBusy < (uint64_t)tmin) break;
                const uint32_t c = lane_cnt[gl];
                idx = (int)(c < K ? c : K) - 1;
                continue;
            }
            const uint32_t* r = reinterpret_cast<const uint32_t*>(&stage[(size_t)gl * K + (uint32_t)idx].bytes[128]);
            const uint32_t rt = r[0], re = r[1];
            if (rt != 0xFFFFFFFFu && re >= tmin) {          // (not void and) it reaches something gathered: it matters
                if (ns == kResolveSet) { full = true; break; }
                trig[ns] = rt; endc[ns] = re; ns++;
                tmin = rt < tmin ? rt : tmin;
            }
            idx--;
        }
        // replay, oldest first; entry 0 is this record
        bool have = false, keep_me = true;
        uint32_t busy = 0;
        if (!full) {
            for (int k = ns - 1; k >= 0; k--) {
                const bool drop = have && trig[k] <= busy;
                if (!drop) { have = true; busy = endc[k]; }
                if (k == 0) keep_me = !drop;
            }
        } else {
            // more candidates can matter than the set holds (dense false syncs over several lanes): the rule itself,
            // sequentially over every record of the channel up to this one -- exact, and only ever this slow here
            for (uint32_t g2 = g_first; g2 <= g; g2++) {
                const uint32_t c2 = g2 == g ? i + 1u : (lane_cnt[g2] < K ? lane_cnt[g2] : K);
                for (uint32_t i2 = 0; i2 < c2; i2++) {
                    const uint32_t* r = reinterpret_cast<const uint32_t*>(&stage[(size_t)g2 * K + i2].bytes[128]);
                    if (r[0] == 0xFFFFFFFFu) continue;
                    const bool drop = have && r[0] <= busy;
                    if (!drop) { have = true; busy = r[1]; }
                    if (g2 == g && i2 == i) keep_me = !drop;
                }
            }
        }
        if (stage[(size_t)g * K + i].sample_index < min_index) keep_me = false;
        stage[(size_t)g * K + i].pdu_type = keep_me ? 0 : 1;
        // SNOUT_PKT_ZB_SEAM_DISAGREED: a seam inside the frame (trigger chip < the lane's first owned chip <= last chip)
        // before which the two timing loops decided a chip of the frame differently (oracle_zigbee.c, same rule).
        // The frame's SFD chip is in this lane: seams before it are those of lanes g, g - 1, ..., seams behind it g + 1, ...
        if (!(stage[(size_t)g * K + i].flags & SNOUT_PKT_ZB_REPAIRED)) {       // (a repaired frame has no seams)
            const uint32_t T = me[0], E = me[1];
            const uint32_t g_last = g_first + lanes_per_slot - 1u;
            bool bad = false;
            for (uint32_t m = g; m > g_first && offs[m] > T; m--) {
                const uint32_t inside = offs[m] - T;
                if (offs[m] <= E && (seam[m] & (inside >= 48u ? 0xFFFFFFFFFFFFull : ((1ull << inside) - 1ull)))) bad = true;
            }
            for (uint32_t m = g + 1u; m <= g_last && offs[m] <= E; m++) {
                const uint32_t inside = offs[m] - T;
                if (offs[m] > T && (seam[m] & (inside >= 48u ? 0xFFFFFFFFFFFFull : ((1ull << inside) - 1ull)))) bad = true;
            }
            if (bad) stage[(size_t)g * K + i].flags |= 4u;
        }
        kept += keep_me ? 1u : 0u;
    }
    lane_kept[g] = kept;
}

// Ordered compaction of per-lane records: lane g holds min(lane_cnt[g], K) records, of which
// lane_kept[g] survived zb_resolve (pdu_type == 0).
__global__ __launch_bounds__(256) void zb_emit(const snout_pkt* __restrict__ stage,
                                               const uint32_t* __restrict__ lane_cnt,
                                               const uint32_t* __restrict__ lane_kept, uint32_t K,
                                               uint32_t total_lanes,
                                               const uint32_t* __restrict__ tile_sums,
                                               const uint32_t* __restrict__ tile_over,
                                               uint32_t n_tiles, uint32_t* __restrict__ totals,
                                               snout_pkt* __restrict__ out, uint32_t out_cap)
{
    __shared__ uint32_t lds[4];
    auto block_sum = [&](const uint32_t* v, uint32_t cnt) -> uint32_t {
        uint32_t x = 0;
        for (uint32_t i = threadIdx.x; i < cnt; i += 256u) x += v[i];
        // reduce across the block
#pragma unroll
        for (int dlt = 32; dlt > 0; dlt >>= 1) x += __shfl_down(x, dlt);
        __syncthreads();
        if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = x;
        __syncthreads();
        return lds[0] + lds[1] + lds[2] + lds[3];
    };
    const uint32_t tile = blockIdx.x;
    if (tile == 0) {
        const uint32_t all = block_sum(tile_sums, n_tiles);
        const uint32_t over = block_sum(tile_over, n_tiles);
        if (threadIdx.x == 0) { totals[0] = total_lanes; totals[1] = all; totals[2] = over; }
    }
    const uint32_t tile_base = block_sum(tile_sums, tile);
    // 1024 lanes per tile, 4 per thread
    const uint32_t g0 = tile * kScanTile + threadIdx.x * kScanItems;
    uint32_t cnt[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t c = (g0 + k < total_lanes) ? lane_cnt[g0 + k] : 0u;
        uint32_t kp = (g0 + k < total_lanes) ? lane_kept[g0 + k] : 0u;
        cnt[k] = c < K ? c : K;
        s += kp < K ? kp : K;
    }
    // exclusive scan of s over the block
    uint32_t inc = s;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const uint32_t t = __shfl_up(inc, dlt);
        if ((int)lane >= dlt) inc += t;
    }
    __syncthreads();
    if (lane == 63) lds[wv] = inc;
    __syncthreads();
    uint32_t off = tile_base + inc - s;
    for (uint32_t w = 0; w < wv; w++) off += lds[w];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        for (uint32_t i = 0; i < cnt[k]; i++) {
            const snout_pkt* rec = &stage[(size_t)(g0 + k) * K + i];
            if (rec->pdu_type != 0) continue;               // dropped by zb_resolve
            if (off < out_cap) {
                const uint4* src = reinterpret_cast<const uint4*>(rec);
                uint4* dst = reinterpret_cast<uint4*>(&out[off]);
#pragma unroll
                for (int q = 0; q < 9; q++) dst[q] = src[q];
                uint4 tail = src[9];
                tail.z = 0u; tail.w = 0u;                   // bytes[128..135]: the resolve stash
                dst[9] = tail;
            }
            off++;
        }
    }
}


// =============================================================================================
// Host side
// =============================================================================================
static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

int ZbCtx::init(uint32_t n_slots_, const uint16_t* slot_channel_, uint32_t threshold_, uint32_t core_,
                uint32_t warmup_, uint32_t batch_cap_)
{
    seg_slots = n_slots = n_slots_;
    batch_cap = batch_cap_ ? batch_cap_ : 1u;
    threshold = threshold_;
    core = core_;
    warmup = warmup_;
    if (const char* e = getenv("SNOUT_ZB_REPAIR")) repair = atoi(e) != 0;      // A/B and tests: 0 = the lanes alone
    if (const char* e = getenv("SNOUT_ZB_TAIL_PRIO")) tail_prio = (uint32_t)atoi(e);
    if (core % 64u || warmup % 64u || warmup >= core) {
        set_last_error("zb_core (%u) and zb_warmup (%u) must be multiples of 64, warmup < core", core, warmup);
        return SNOUT_EINVAL;
    }
    std::vector<float> atan_tab(257);
    for (int i = 0; i < 257; i++) atan_tab[i] = (float)atan((double)i / 255.0);
    {   // IIR carry-in tables: w[m] = alpha (1-alpha)^m, D64 = (1-alpha)^64, block decay factors
        const double alpha = 0.00016, one_minus = 1.0 - 0.00016;
        double w[64], v = alpha, p = 1.0;
        for (int m = 0; m < 64; m++) { w[m] = v; v = v * one_minus; }
        for (int m = 0; m < 64; m++) p = p * one_minus;
        d64 = p;
        set_shape(core, warmup);
        if (int rc = d_iirw.ensure(64 * 8)) return rc;
        SNOUT_HIP(hipMemcpy(d_iirw.p, w, 64 * 8, hipMemcpyHostToDevice));
    }
    if (int rc = d_atan.ensure(257 * 4)) return rc;
    if (int rc = d_mmse.ensure(129 * 8 * 4)) return rc;
    if (int rc = d_slot_channel.ensure((size_t)seg_slots * batch_cap * 2)) return rc;
    SNOUT_HIP(hipMemcpy(d_atan.p, atan_tab.data(), 257 * 4, hipMemcpyHostToDevice));
    SNOUT_HIP(hipMemcpy(d_mmse.p, kMmseTapsHost, 129 * 8 * 4, hipMemcpyHostToDevice));
    for (uint32_t b = 0; b < batch_cap; b++)      // the slots of every segment of a batch carry the same channels
        SNOUT_HIP(hipMemcpy(d_slot_channel.as<uint16_t>() + (size_t)b * seg_slots, slot_channel_, seg_slots * 2, hipMemcpyHostToDevice));
    return 0;
}

void ZbCtx::destroy()
{
    d_atan.release(); d_mmse.release(); d_slot_channel.release();
    d_d.release(); d_TR.release(); d_lane_out.release(); d_cand.release(); d_lane_u32.release();
    d_stream.release(); d_stage.release(); d_lane_cnt.release(); d_soft.release();
    d_iirw.release(); d_S.release(); d_Lblk.release();
    d_lane_end.release(); d_snap.release(); d_req.release();
}

// Lane shape and the block decay factors of the IIR carry-in that depend on it.
void ZbCtx::set_shape(uint32_t core_, uint32_t warmup_)
{
    core = core_;
    warmup = warmup_;
    dcore = 1.0;
    for (uint32_t k = 0; k < core / 64u; k++) dcore = dcore * d64;
    dfirst = 1.0;
    for (uint32_t k = 0; k < (core - warmup) / 64u; k++) dfirst = dfirst * d64;
}

int ZbCtx::reserve(uint64_t n, uint32_t segs)
{
    if (segs == 0 || segs > batch_cap) { set_last_error("batch of %u segments (handle created for %u)", segs, batch_cap); return SNOUT_EINVAL; }
    n_slots = seg_slots * segs;                     // slot = (segment of the batch, channel)
    lanes_per_slot = cdiv(n, core);
    total_lanes = lanes_per_slot * n_slots;
    n_waves = cdiv(total_lanes, 64);
    nt = (core + warmup) / 64u + 1u;
    tiles_per_slot = cdiv(lanes_per_slot, kScanTile);
    stream_words = (n / 64u + 5u) & ~1ull;          // at most one chip per sample; even: cleared in 16-byte units
    if (n >= (1ull << 31)) {    // chip and lane-relative indices are 32-bit
        set_last_error("Zigbee segment of %llu channel samples: at most 2^31 - 1 per call", (unsigned long long)n);
        return SNOUT_ERANGE;
    }
    if (cdiv(total_lanes, 1024) > kMaxTiles) { set_last_error("too many lanes"); return SNOUT_ERANGE; }
    if ((uint64_t)n_waves * nt > 0x7FFFFFFFull) { set_last_error("too many lane tiles"); return SNOUT_ERANGE; }
    d_stride = (uint64_t)(lanes_per_slot + 1u) * core + 1024u;   // whole tiles past the last core, 16-B rows
    d_stride = (d_stride + 1023u) & ~1023ull;
    if (int rc = d_d.ensure(d_stride * n_slots * 4u)) return rc;
    if (int rc = d_TR.ensure((uint64_t)n_waves * nt * 9u * 64u * 4u)) return rc;
    if (int rc = d_lane_out.ensure((uint64_t)total_lanes * 32u)) return rc;
    if (int rc = d_cand.ensure((uint64_t)total_lanes * 12u * 4u)) return rc;
    // first_owned | owned | offs | tsum | slot_total
    // first_owned | owned | offs | tsum | slot_total | (8-byte aligned) seam masks, one u64 per lane
    if (int rc = d_lane_u32.ensure(((uint64_t)total_lanes * 3u + (uint64_t)tiles_per_slot * n_slots + n_slots + 2u) * 4u + (uint64_t)total_lanes * 8u)) return rc;
    if (int rc = d_stream.ensure(stream_words * n_slots * 8u * 3u)) return rc;     // chips | {chips, match} pairs
    if (int rc = d_stage.ensure((uint64_t)total_lanes * pkts_per_lane * sizeof(snout_pkt))) return rc;
    if (int rc = d_lane_cnt.ensure(2u * ((uint64_t)total_lanes + 1024u) * 4u)) return rc;     // raw counts, kept counts
    max_out = total_lanes * pkts_per_lane;
    if (int rc = d_soft.ensure(((uint64_t)kSoftCap * 2u + 16u) * 4u)) return rc;
    nsb = (n + 63u) / 64u;
    if (int rc = d_S.ensure(nsb * n_slots * 8u)) return rc;
    if (int rc = d_Lblk.ensure((uint64_t)total_lanes * 8u)) return rc;
    if (int rc = d_lane_end.ensure((uint64_t)total_lanes * sizeof(ZbLaneEnd))) return rc;
    if (int rc = d_snap.ensure((uint64_t)total_lanes * sizeof(ZbSnap))) return rc;
    if (int rc = d_req.ensure(((uint64_t)total_lanes + 1u) * 4u)) return rc;
    return 0;
}

// Zero-fill of the chip streams.  A kernel of our own rather than hipMemsetAsync: for the 62 MB of a
// 1e9-sample segment the runtime's memset was seen to hold the host until the stream had drained
// and then some (kernels of the tail launched 11 ms after zb_mm had finished, in some processes:
// 16 ms per segment instead of 5).
__global__ __launch_bounds__(256) void zb_clear(ulonglong2* __restrict__ p, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += stride) p[i] = make_ulonglong2(0ull, 0ull);
}

// a7 and the glue before it: stitched chip streams -> sinks -> per-lane records (after zb_mm).
int ZbCtx::launch_sinks(uint64_t n, const SegBatch& segs, hipStream_t st)
{
    uint32_t* first_owned = d_lane_u32.as<uint32_t>();
    uint32_t* owned = first_owned + total_lanes;
    uint32_t* offs = owned + total_lanes;
    uint32_t* tsum = offs + total_lanes;
    uint32_t* slot_total = tsum + (uint64_t)tiles_per_slot * n_slots;
    unsigned long long* seam = seam_masks();
    {
        const uint64_t n16 = stream_words * n_slots / 2u;             // stream_words is even
        hipLaunchKernelGGL(zb_clear, dim3((uint32_t)std::min<uint64_t>(cdiv(n16, 256), 4096u)), dim3(256), 0, st,
                           d_stream.as<ulonglong2>(), n16);
    }
    hipLaunchKernelGGL(zb_stitch, dim3(tiles_per_slot, n_slots), dim3(256), 0, st,
                       d_lane_out.as<ZbLaneOut>(), d_cand.as<uint32_t>(), lanes_per_slot, core, warmup,
                       tiles_per_slot, first_owned, owned, tsum, seam, d_req.as<uint32_t>());
    hipLaunchKernelGGL(zb_offsets, dim3(tiles_per_slot, n_slots), dim3(256), 0, st, owned, tsum,
                       lanes_per_slot, tiles_per_slot, offs, slot_total);
    {
        // a wave's stretch of the stream: 64 lanes x ~ core / 2 chips (the loop's step stays within 2 (1 +- 0.0002));
        // staged through LDS while two waves' stretches fit 48 KB, stored directly otherwise
        const uint32_t want = core / 2u + 64u;                          // words per wave
        const uint32_t wcap = (uint64_t)want * 16u <= 48u * 1024u ? want : 0u;
        hipLaunchKernelGGL(zb_scatter, dim3(cdiv(total_lanes, 128)), dim3(128), (size_t)wcap * 16u, st, d_TR.as<uint32_t>(), nt,
                           lanes_per_slot, total_lanes, warmup >= 64u ? (warmup >> 6) - 1u : 0u, first_owned, offs, owned,
                           d_stream.as<unsigned long long>(), stream_words, wcap);
    }
    hipLaunchKernelGGL(zb_match, dim3(cdiv(stream_words, 256), n_slots), dim3(256), 0, st,
                       d_stream.as<unsigned long long>(), stream_words, slot_total, threshold,
                       reinterpret_cast<ulonglong2*>(d_stream.as<unsigned long long>() + stream_words * n_slots));
    hipLaunchKernelGGL(zb_walk, dim3(cdiv(total_lanes, 256)), dim3(256), 0, st,
                       reinterpret_cast<const ulonglong2*>(d_stream.as<unsigned long long>() + stream_words * n_slots),
                       stream_words, offs, first_owned, slot_total,
                       d_TR.as<uint32_t>(), nt, lanes_per_slot, total_lanes, core, warmup, threshold,
                       d_slot_channel.as<uint16_t>(), segs, d_stage.as<snout_pkt>(), pkts_per_lane,
                       d_lane_cnt.as<uint32_t>(), repair ? d_snap.as<ZbSnap>() : nullptr, d_req.as<uint32_t>(), tail_prio & 1u);
    // frame repair: at most one request per lane; the waves beyond the count return at once
    if (repair)
        hipLaunchKernelGGL(zb_repair, dim3(cdiv(total_lanes, 64)), dim3(64), 0, st, d_d.as<float>(), d_stride, n, lanes_per_slot,
                           core, warmup, d_mmse.as<float>(), d_lane_end.as<ZbLaneEnd>(), d_snap.as<ZbSnap>(), d_req.as<uint32_t>(),
                           reinterpret_cast<const ulonglong2*>(d_stream.as<unsigned long long>() + stream_words * n_slots),
                           stream_words, threshold, d_stage.as<snout_pkt>(), pkts_per_lane, (tail_prio >> 1) & 1u,
                           d_TR.as<uint32_t>(), nt, offs, first_owned, owned, slot_total);
    SNOUT_HIP(hipGetLastError());
    return 0;
}

// Test tap: soft intermediates of one lane of the LAST processed segment (the tiles are still resident).
int ZbCtx::soft(uint32_t stage_id, uint32_t lane, uint64_t n, float* out, uint64_t cap, uint64_t* n_out)
{
    *n_out = 0;
    if (stage_id == SNOUT_STAGE_ZB_DISCRIM) {
        // `lane` selects the channel slot here
        if (lane >= n_slots) return SNOUT_EINVAL;
        const uint64_t m = n < cap ? n : cap;
        if (m) SNOUT_HIP(hipMemcpy(out, d_d.as<float>() + (uint64_t)lane * d_stride, m * 4u, hipMemcpyDeviceToHost));
        *n_out = n;
        return n > cap ? SNOUT_EOVERFLOW : 0;
    }
    if (lane >= total_lanes) return SNOUT_EINVAL;
    float* sz = d_soft.as<float>();
    float* sc = d_soft.as<float>() + kSoftCap;
    uint32_t* sn = (uint32_t*)(d_soft.as<float>() + 2 * kSoftCap);
    // re-run the lanes with the tap on (rewrites identical tile records)
    hipLaunchKernelGGL(zb_mm<true>, dim3(cdiv(n_waves, kMmWaves)), dim3(kMmWaves * 64), 0, nullptr, d_d.as<float>(), d_stride, n, nt, lanes_per_slot,
                       total_lanes, core, warmup, d_mmse.as<float>(), d_Lblk.as<double>(), ((1u << 18) + core - 1u) / core, dfirst, dcore,
                       d_TR.as<uint32_t>(), d_lane_out.as<ZbLaneOut>(), d_lane_end.as<ZbLaneEnd>(), d_cand.as<uint32_t>(),
                       sz, sc, lane, (uint32_t)kSoftCap, sn);
    SNOUT_HIP(hipDeviceSynchronize());
    uint32_t nch = 0;
    SNOUT_HIP(hipMemcpy(&nch, sn, 4, hipMemcpyDeviceToHost));
    if (stage_id == SNOUT_STAGE_ZB_CHIPS) {
        const uint64_t have = nch < (uint32_t)kSoftCap ? nch : (uint32_t)kSoftCap;
        const uint64_t m = have < cap ? have : cap;
        SNOUT_HIP(hipMemcpy(out, sc, m * 4u, hipMemcpyDeviceToHost));
        *n_out = have;
        return have > cap ? SNOUT_EOVERFLOW : 0;
    }
    if (stage_id == SNOUT_STAGE_ZB_DCREMOVED) {
        // z is defined for every sample the lane filtered; report up to the tap capacity
        const uint64_t li = lane % lanes_per_slot;
        const uint64_t cs = li * (uint64_t)core, s0 = cs > warmup ? cs - warmup : 0;
        uint64_t have = n > s0 ? n - s0 : 0;
        have = std::min<uint64_t>(have, (cs - s0) + core);     // always filtered that far
        have = std::min<uint64_t>(have, kSoftCap);
        const uint64_t m = have < cap ? have : cap;
        SNOUT_HIP(hipMemcpy(out, sz, m * 4u, hipMemcpyDeviceToHost));
        *n_out = have;
        return have > cap ? SNOUT_EOVERFLOW : 0;
    }
    return SNOUT_EINVAL;
}

// Front end, first part (a4) on the caller's stream: the discriminator rows and the IIR sub-block sums of a narrowband
// segment (a wideband handle's fused channelizer has written them already: d_iq == nullptr).
// iq: [n_slots][iq_stride] complex samples at 4 Msps per channel, device memory.  No host sync.
int ZbCtx::enqueue_front(const void* d_iq, uint64_t n, uint64_t iq_stride, hipStream_t st,
                         ResultSlot& s, bool time_front, int fmt)
{
    if (time_front) SNOUT_HIP(hipEventRecord(s.ev_k0, st));
    if (n < 9u) return 0;   // no interpolator window fits: nothing to launch
    if (d_iq) {             // otherwise the fused channelizer has already written d and S
#define SNOUT_ZBD(F)                                                                                  \
    hipLaunchKernelGGL(zb_discrim<F>, dim3(cdiv(d_stride, 1024 * kDiscChunks), n_slots), dim3(256), 0, st, d_iq, n, \
                       iq_stride, d_stride, nsb, d_atan.as<float>(), d_iirw.as<double>(), d_d.as<float>(), \
                       d_S.as<double>())
        if (fmt == kFmtSc8) SNOUT_ZBD(kFmtSc8);
        else if (fmt == kFmtSc16) SNOUT_ZBD(kFmtSc16);
        else SNOUT_ZBD(kFmtCf32);
#undef SNOUT_ZBD
    }
    SNOUT_HIP(hipGetLastError());
    return 0;
}

// Front end, second part (a5-a6): IIR carry-in and the lanes.  A narrowband handle runs them on the work set's own (tail)
// stream, so that the NEXT segment's discriminator (a streaming kernel: HBM-bound, 44 registers) runs beside them; a
// wideband handle keeps them on the caller's stream behind the channelizer -- beside the next channelizer they were
// measured too (zb_mm cut to 96 registers so that it fits next to the channelizer's four waves per SIMD): the
// channelizer's 16 fast_atan2f per output time keep the vector pipe as busy as the lanes need it, it took 3.5 ms instead
// of 1.8 and the step did not move (profiles/r5_repair.md).  No host sync.
int ZbCtx::enqueue_lanes(uint64_t n, hipStream_t st)
{
    if (n < 9u) return 0;
    hipLaunchKernelGGL(zb_iir_fold, dim3(cdiv(total_lanes, 256)), dim3(256), 0, st, d_S.as<double>(), nsb,
                       lanes_per_slot, total_lanes, core, warmup, d64, d_Lblk.as<double>());
    hipLaunchKernelGGL(zb_mm<false>, dim3(cdiv(n_waves, kMmWaves)), dim3(kMmWaves * 64), 0, st, d_d.as<float>(), d_stride, n, nt, lanes_per_slot,
                       total_lanes, core, warmup, d_mmse.as<float>(), d_Lblk.as<double>(), ((1u << 18) + core - 1u) / core, dfirst, dcore,
                       d_TR.as<uint32_t>(), d_lane_out.as<ZbLaneOut>(), d_lane_end.as<ZbLaneEnd>(), d_cand.as<uint32_t>(),
                       (float*)nullptr, (float*)nullptr, 0xFFFFFFFFu, 0u, (uint32_t*)nullptr);
    SNOUT_HIP(hipGetLastError());
    return 0;
}

unsigned long long* ZbCtx::seam_masks() const
{
    const uint64_t words = (uint64_t)total_lanes * 3u + (uint64_t)tiles_per_slot * n_slots + n_slots;
    return reinterpret_cast<unsigned long long*>(d_lane_u32.as<uint32_t>() + ((words + 1u) & ~1ull));
}

PfbZbTarget ZbCtx::pfb_target(uint32_t seg)
{
    // the rows' tails are known to be zero only for the buffer, the row length and the rows they were zeroed for
    if (d_d.p != tails_ptr || d_stride != tails_stride || n_slots > tails_rows) {
        tails_dirty_to = ~0ull;
        tails_ptr = d_d.p; tails_stride = d_stride; tails_rows = n_slots;
    }
    return PfbZbTarget{d_d.as<float>() + (uint64_t)seg * seg_slots * d_stride, d_stride,
                       d_S.as<double>() + (uint64_t)seg * seg_slots * nsb, nsb, d_atan.as<float>(), d_iirw.as<double>(),
                       seg == 0 ? &tails_dirty_to : nullptr, tails_rows};
}

// Tail (stitch, a7, ordered compaction into s.d_out / s.d_totals) on the handle's tail stream, so
// that it overlaps the next segment's front end (which uses the other work set).  No host sync.
// totals (u32): [0] lanes  [1] packets  [2] lanes that held more than pkts_per_lane frames
int ZbCtx::enqueue_tail(uint64_t n, const SegBatch& segs_in, hipStream_t st, ResultSlot& s, bool time_front)
{
    if (int rc = s.d_out.ensure((uint64_t)max_out * sizeof(snout_pkt))) return rc;
    uint32_t* tot = s.d_totals.as<uint32_t>();
    uint32_t* sums = tot + 16;
    uint32_t* over = tot + 16 + kMaxTiles;
    if (n < 9u) {
        SNOUT_HIP(hipMemsetAsync(tot, 0, 16, st));
        if (time_front) SNOUT_HIP(hipEventRecord(s.ev_k1, st));
        return 0;
    }
    SegBatch segs = segs_in;
    segs.slots_per_seg = seg_slots;
    if (int rc = launch_sinks(n, segs, st)) return rc;
    if (time_front) SNOUT_HIP(hipEventRecord(s.ev_k1, st));
    const uint32_t n_tiles = cdiv(total_lanes, kScanTile);
    uint32_t* lane_kept = d_lane_cnt.as<uint32_t>() + total_lanes + 1024u;
    hipLaunchKernelGGL(zb_resolve, dim3(cdiv(total_lanes, 256)), dim3(256), 0, st, d_stage.as<snout_pkt>(),
                       d_lane_cnt.as<uint32_t>(), pkts_per_lane, lanes_per_slot, total_lanes, d_lane_u32.as<uint32_t>() + 2u * (uint64_t)total_lanes, segs, seam_masks(), lane_kept);
    launch_tile_reduce(lane_kept, nullptr, total_lanes, total_lanes, pkts_per_lane,
                       sums, over, n_tiles, st);
    hipLaunchKernelGGL(zb_emit, dim3(n_tiles), dim3(256), 0, st, d_stage.as<snout_pkt>(),
                       d_lane_cnt.as<uint32_t>(), lane_kept, pkts_per_lane, total_lanes, sums, over, n_tiles, tot,
                       s.d_out.as<snout_pkt>(), max_out);
    SNOUT_HIP(hipGetLastError());
    return 0;
}

bool ZbCtx::check_overflow(const ResultSlot& s)
{
    overflow = s.h_totals[2] != 0;
    if (!overflow) return false;
    set_last_error("more than %u frames in one lane", pkts_per_lane);
    pkts_per_lane *= 4;       // a lane held more frames than provisioned: the caller runs it again
    return true;
}

}  // namespace snout
