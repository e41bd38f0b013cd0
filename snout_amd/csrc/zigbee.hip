// zigbee.hip — IEEE 802.15.4 O-QPSK receive kernels for gfx950 (CDNA4, wave64).
//
// Replaces the GNU Radio receive flowgraph the reference spawns for `snout zigbee scan`
// (snout/modulations/Zigbee/hackrf/Zigbee_rx/top_block.py:52-89):
//   a4 quadrature_demod_cf(1)                       -> zb_discrim   (pointwise, HBM-bound)
//   a5 x - single_pole_iir_filter_ff(0.00016)(x)    -> zb_lanes     (serial per lane, fp64 state)
//   a6 clock_recovery_mm_ff(2, .000225, .5, .03, .0002)              (serial feedback loop)
//   a7 ieee802_15_4.packet_sink(10)                                  (serial FSM)
//
// The feedback stages are inherently sequential, so parallelism comes from lanes: lane i owns the
// core [i*core, (i+1)*core) of one channel, starts `warmup` samples early from the initial loop
// state and runs past its core only to finish a frame whose preamble it found inside the core
// (the same segmentation the oracle uses).  One wave = 64 lanes; a block is one wave.
//
// zb_lanes data flow per 64-sample tile:  HBM d[] --(64 coalesced 256-B row loads)--> LDS ring
// (transposed, row stride 65 words: bank = (time + lane) mod 32, conflict-free) --> each lane runs
// IIR, M&M and the sink over its own column.
#include "common.h"

namespace snout {

// RX correlator words of gr-ieee802-15-4's packet_sink (FM-domain chip words, MSB = first chip).
__constant__ uint32_t kChipMap[16] = {
    1618456172u, 1309113062u, 1826650030u, 1724778362u, 778887287u, 2061946375u, 2007919840u,
    125494990u,  529027475u,  838370585u,  320833617u,  422705285u, 1368596360u, 85537272u,
    139563807u,  2021988657u};

static const float kMmseTapsHost[129][8] = {
#include "mmse_taps.inc"
};

// ---------------------------------------------------------------------------------------------
// a4: d[t] = fast_atan2f(Im(x[t] conj x[t-1]), Re(...)),  x[-1] = 0.  GNU Radio's table-driven
// fast_atan2f (257-entry atan table, linear interpolation, octant fix-up).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2f_tab(float y, float x, const float* __restrict__ tab)
{
    const float ya = fabsf(y), xa = fabsf(x);
    if (!(ya > 0.0f || xa > 0.0f)) return 0.0f;
    const float z = ya < xa ? ya / xa : xa / ya;
    float base;
    if (z < 0.003921569f) {
        base = z;
    } else {
        float a = z * 255.0f;
        const int k = ((int)a) & 0xff;
        a -= (float)k;
        const float t0 = tab[k];
        base = t0 + (tab[k + 1] - t0) * a;
    }
    float ang;
    if (xa > ya) {
        if (x >= 0.0f) ang = y >= 0.0f ? base : -base;
        else ang = y >= 0.0f ? 3.14159265358979323846f - base : base - 3.14159265358979323846f;
    } else {
        if (y >= 0.0f) ang = x >= 0.0f ? 1.57079632679489661923f - base : 1.57079632679489661923f + base;
        else ang = x >= 0.0f ? -1.57079632679489661923f + base : -1.57079632679489661923f - base;
    }
    return ang;
}

// One wave per 64-sample sub-block of one channel.  Besides d[t] the wave emits S_j, the
// zero-state response of the single-pole IIR to its 64 samples (double, fixed pairwise order), from
// which zb_iir_carry builds every lane's initial filter state (the "IIR carry-in", see the oracle).
__global__ __launch_bounds__(256) void zb_discrim(const float2* __restrict__ iq, uint64_t n,
                                                  uint64_t iq_stride, uint32_t n_slots, uint64_t nsb,
                                                  const float* __restrict__ atan_tab,
                                                  const double* __restrict__ iir_w,
                                                  float* __restrict__ d, uint64_t d_stride,
                                                  double* __restrict__ S)
{
    __shared__ float tab[257];
    __shared__ double wts[64];
    for (uint32_t i = threadIdx.x; i < 257; i += 256) tab[i] = atan_tab[i];
    if (threadIdx.x < 64) wts[threadIdx.x] = iir_w[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t total = nsb * n_slots;
    for (uint64_t w = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); w < total;
         w += (uint64_t)gridDim.x * 4u) {
        const uint64_t slot = w / nsb, j = w - slot * nsb;
        const uint64_t t = 64u * j + lane;
        const float2* x = iq + slot * iq_stride;
        float ang = 0.0f;
        if (t < n) {
            const float2 a = x[t];
            const float2 p = t ? x[t - 1] : make_float2(0.0f, 0.0f);
            const float re = a.x * p.x + a.y * p.y;      // contraction is off: products round first
            const float im = a.y * p.x - a.x * p.y;
            ang = fast_atan2f_tab(im, re, tab);
            if (!(fabsf(ang) <= 4.0f)) ang = 0.0f;      // non-finite input: defined as 0 (as the oracle)
            d[slot * d_stride + t] = ang;
        }
        double v = wts[63u - lane] * (double)ang;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_down(v, off);
        if (lane == 0) S[slot * nsb + j] = v;
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
    for (uint64_t j = b0; j < b1; j++) L = d64 * L + (j < nsb ? s[j] : 0.0);
    Lblk[g] = L;
}

// lp_in[i+1] = Dblk_i lp_in[i] + L_i along the lanes of a channel: one wave per channel walks the
// lanes 64 at a time (coalesced load of L, then the sequential recurrence with v_readlane).
__global__ __launch_bounds__(64) void zb_iir_scan(const double* __restrict__ Lblk, uint32_t lanes_per_slot,
                                                  uint32_t n_slots, double dfirst, double dcore,
                                                  double* __restrict__ lp_in)
{
    const uint32_t slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= n_slots) return;
    double lp = 0.0;
    double nxt = lane < lanes_per_slot ? Lblk[slot * lanes_per_slot + lane] : 0.0;
    for (uint32_t l0 = 0; l0 < lanes_per_slot; l0 += 64u) {
        const uint32_t g = slot * lanes_per_slot + l0 + lane;
        const double v = nxt;
        nxt = (l0 + 64u + lane < lanes_per_slot) ? Lblk[g + 64u] : 0.0;     // in flight during the chain
        double mine = 0.0;
#pragma unroll 8
        for (uint32_t i = 0; i < 64u; i++) {
            if (lane == i) mine = lp;                          // state before lane l0+i
            const double Li = __shfl(v, (int)i);
            lp = ((l0 + i) == 0u ? dfirst : dcore) * lp + Li;
        }
        if (l0 + lane < lanes_per_slot) lp_in[g] = mine;
    }
}

// ---------------------------------------------------------------------------------------------
// a5-a7: lanes.
// ---------------------------------------------------------------------------------------------
constexpr int kRingLen = 128;      // two 64-sample tiles of history per lane
constexpr int kRingStride = 137;   // per-lane row: 128 + 8 mirrored entries, odd -> bank = (lane + time) mod 32
// ring[lane * kRingStride + (t & 127)] holds sample t of the lane; entries 128..135 mirror 0..7 so an
// 8-sample window never wraps and is read with immediate offsets from one address.

struct SinkState {
    int state;          // 0 search, 1 have_sync, 2 have_header
    uint32_t shift;
    int preamble_cnt, chip_cnt, packet_byte, byte_index, packetlen, packetlen_cnt, payload_cnt;
    uint32_t lqi, lqi_cnt;
    uint32_t trigger;   // lane-relative sample index of the first preamble match
    uint32_t c0, c1, c2;   // running FCS: after all bytes, one byte ago, two bytes ago
    uint32_t b_prev, b_last;   // the last two PSDU bytes
};

__device__ __forceinline__ void enter_search(SinkState& s)
{
    s.state = 0; s.shift = 0; s.preamble_cnt = 0; s.chip_cnt = 0; s.packet_byte = 0;
}

__device__ __forceinline__ uint32_t chip_dist(uint32_t shift, uint32_t word)
{
    return (uint32_t)__popc((shift & 0x7FFFFFFEu) ^ (word & 0x7FFFFFFEu));
}

__device__ __forceinline__ int decode_chips(SinkState& s, uint32_t th)
{
    // words 8..15 are the 30-bit complements of words 0..7 (under the mask): dist = 30 - dist
    uint32_t dlo[8];
#pragma unroll
    for (int i = 0; i < 8; i++) dlo[i] = chip_dist(s.shift, kChipMap[i]);
    int best = 0xFF;
    uint32_t min_t = 33;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t t = i < 8 ? dlo[i] : 30u - dlo[i - 8];
        if (t < min_t) { best = i; min_t = t; }
    }
    if (min_t < th) {
        if (s.lqi_cnt < 8) { s.lqi += 32 - min_t; s.lqi_cnt++; }
        return best & 0xF;
    }
    return 0xFF;
}

__device__ __forceinline__ uint32_t crc16_step(uint32_t c, uint32_t byte)
{
    c ^= byte;
#pragma unroll
    for (int k = 0; k < 8; k++) c = (c & 1u) ? ((c >> 1) ^ 0x8408u) : (c >> 1);
    return c;
}

// The packet sink, split by what a chip can trigger.
//  * searching (state 0, preamble_cnt == 0): every chip is tested against symbol 0;
//  * otherwise chips are only counted until the next symbol boundary (32 chips), where
//    sink_symbol() does the work.  The lane loop therefore shifts whole runs of chips in at once.
// Together they are exactly gr-ieee802-15-4's per-chip state machine.
__device__ __forceinline__ void sink_search_chip(SinkState& s, uint32_t bit, uint32_t at, uint32_t th)
{
    s.shift = (s.shift << 1) | bit;
    if (chip_dist(s.shift, kChipMap[0]) < th) {
        s.preamble_cnt = 1;         // chip_cnt stays 0: the boundary is 32 chips after this one
        s.trigger = at;
    }
}

// Called when chip_cnt reached 32 (s.shift holds the symbol's chips).  Returns true when a frame
// completed (caller publishes, then enter_search).
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
    if (c == 0xFF) { enter_search(s); return false; }
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

// Continuation of a lane that reached the end of its core inside a frame it owns.
struct ZbLaneSave {
    double lp;
    float mu, omega, last;
    uint32_t g, ii, znext, n_pk, n_chips;
    SinkState s;
    float zhist[16];        // z[znext-16 .. znext)
};

// Two passes.  RESUME = false: every lane runs from its warm-up start to the end of its core; a
// lane that is inside a frame it owns at that point saves its loop state and stops, so no wave is
// held back by its longest frame.  RESUME = true: the saved lanes (compacted, dense waves) finish
// their frames.  Both passes perform exactly the operations of the single sequential lane.
template <bool RESUME>
__global__ __launch_bounds__(64) void zb_lanes(
    const float* __restrict__ d, uint64_t n, uint64_t d_stride, uint32_t lanes_per_slot,
    uint32_t total_lanes, uint32_t core, uint32_t warmup, uint32_t th,
    const uint16_t* __restrict__ slot_channel, uint64_t first_index,
    const float* __restrict__ mmse, snout_pkt* __restrict__ stage, uint32_t K,
    uint32_t* __restrict__ lane_cnt, const double* __restrict__ lp_in,
    ZbLaneSave* __restrict__ saves, uint32_t* __restrict__ n_saves,
    float* __restrict__ soft_z, float* __restrict__ soft_chips,
    uint32_t soft_lane, uint32_t soft_cap, uint32_t* __restrict__ soft_n)
{
    __shared__ float ring[64 * kRingStride];
    __shared__ __attribute__((aligned(16))) float taps[129 * 8];
    const uint32_t l = threadIdx.x;
    for (uint32_t i = l; i < 129u * 8u; i += 64u) taps[i] = mmse[i];
    const uint32_t job = blockIdx.x * 64u + l;
    uint32_t n_jobs = total_lanes;
    if constexpr (RESUME) {
        n_jobs = *n_saves;
        if (blockIdx.x * 64u >= n_jobs) return;          // whole wave beyond the job list
    }
    const bool active = job < n_jobs;
    ZbLaneSave sv;
    if constexpr (RESUME) { if (active) sv = saves[job]; }
    const uint32_t g = RESUME ? (active ? sv.g : 0u) : job;
    const uint32_t slot = active ? g / lanes_per_slot : 0u;
    const uint32_t li = active ? g % lanes_per_slot : 0u;
    const uint64_t core_start = (uint64_t)li * core;
    const uint64_t s0 = core_start > warmup ? core_start - warmup : 0ull;
    // lane-relative coordinates (r = t - s0); a resumed lane shifts its origin to 64 samples before
    // the first sample it still has to filter, so every lane of the wave starts at tile 0
    const uint32_t origin = RESUME ? (active ? sv.znext - 64u : 0u) : 0u;
    const uint32_t rel_core_start = (uint32_t)(core_start - s0);
    const uint32_t rel_core_end = rel_core_start + core;            // in unshifted coordinates
    const uint64_t avail64 = active && n > s0 + origin ? n - s0 - origin : 0ull;
    // a frame that starts before the core end is over within 17 024 + a few samples
    const uint64_t lane_max = (uint64_t)rel_core_end + 20000u - origin;
    const uint32_t avail = (uint32_t)(avail64 < lane_max ? avail64 : lane_max);
    const uint64_t base = (uint64_t)slot * d_stride + s0 + origin;  // offset of r = 0 in d
    __syncthreads();

    const double alpha = 0.00016, one_minus = 1.0 - 0.00016;
    const float omega_mid = 2.0f, gain_omega = 0.000225f, gain_mu = 0.03f;
    const float omega_lim = omega_mid * 0.0002f;
    double lp = (!RESUME && active) ? lp_in[g] : 0.0;    // IIR state carried in from before the lane
    float mu = 0.5f, omega = 2.0f, last = 0.0f;
    SinkState s;
    enter_search(s);
    s.byte_index = s.packetlen = s.packetlen_cnt = s.payload_cnt = 0;
    s.lqi = s.lqi_cnt = 0; s.trigger = 0; s.c0 = s.c1 = s.c2 = 0; s.b_prev = s.b_last = 0;
    uint32_t ii = 0;            // window start
    uint32_t znext = 0;         // first sample not yet through the IIR
    uint32_t n_pk = 0, n_chips = 0;
    if constexpr (RESUME) {
        if (active) {
            lp = sv.lp; mu = sv.mu; omega = sv.omega; last = sv.last; s = sv.s;
            ii = sv.ii - origin; znext = 64u; n_pk = sv.n_pk; n_chips = sv.n_chips;
        }
    }
    bool done = !active || avail < 8u;
    const bool tap = active && g == soft_lane && soft_chips != nullptr;

    // Tile t+1 is fetched into registers (one value per row) while tile t is consumed from LDS.
    // Row = lane whose samples are loaded: its base and length come from v_readlane (SGPRs) and
    // form a buffer descriptor, so reads past the lane's end return 0 without a branch.
    float pre[64];
    auto fetch_tile = [&](uint32_t r0) {
        const uint32_t voff = (r0 + l) * 4u;
#pragma unroll
        for (uint32_t row = 0; row < 64u; row++) {
            const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, (int)row);
            const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), (int)row);
            const uint32_t av = (uint32_t)__builtin_amdgcn_readlane((int)(done ? 0u : avail), (int)row);
            const float* rp = d + (((uint64_t)b_hi << 32) | b_lo);
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rp, 0, (int)(av * 4u), 0x00020000);
            pre[row] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
        }
    };
    fetch_tile(0u);
    for (uint32_t tile = 0; ; tile++) {
        if (__ballot(!done) == 0ull) break;
        const uint32_t r0 = tile * 64u;
        // ---- stage the prefetched tile: thread l holds sample r0+l of every row (lane)
        {
            const uint32_t tt = (r0 + l) & (kRingLen - 1);
#pragma unroll
            for (uint32_t row = 0; row < 64u; row++) ring[row * kRingStride + tt] = pre[row];
        }
        __builtin_amdgcn_wave_barrier();    // one wave per workgroup: LDS is program-ordered
        fetch_tile(r0 + 64u);               // in flight while this tile is processed
        if constexpr (RESUME) {
            if (tile == 0 && active) {      // the 16 filtered samples before znext
#pragma unroll
                for (uint32_t k = 0; k < 16u; k++) ring[l * kRingStride + 48u + k] = sv.zhist[k];
            }
        }
        const uint32_t staged = r0 + 64u;
        const uint32_t hi = staged < avail ? staged : avail;
        if (!done) {
            // ---- phase 1: a5 + a6 over this tile.  Hard chip decisions and the window advance of
            //      every chip are packed into three 64-bit words (<= 64 chips per 64-sample tile):
            //      chip c sits at bit 63-c, so a run of chips is one shift away from the sink's
            //      shift-register order.
            uint64_t cw = 0, d_lo = 0, d_hi = 0;
            uint32_t nc = 0;
            const uint32_t ii_start = ii;
            float* row = &ring[l * kRingStride];
            // a5: DC removal of the staged tile (sequential fp64 recurrence, 8 samples per step);
            // a resumed lane's first tile is already filtered (its last 16 values were restored)
            if (!(RESUME && tile == 0)) {
                const uint32_t t0 = r0 & (kRingLen - 1);
#pragma unroll 2
                for (uint32_t q = 0; q < 64u; q += 8u) {
                    float xv[8];
#pragma unroll
                    for (uint32_t k = 0; k < 8u; k++) xv[k] = row[t0 + q + k];
#pragma unroll
                    for (uint32_t k = 0; k < 8u; k++) {
                        lp = alpha * (double)xv[k] + one_minus * lp;
                        xv[k] = xv[k] - (float)lp;
                        row[t0 + q + k] = xv[k];
                    }
                    if (t0 + q == 0u) {                 // mirror of entries 0..7
#pragma unroll
                        for (uint32_t k = 0; k < 8u; k++) row[kRingLen + k] = xv[k];
                    }
                    if (tap && soft_z) {
#pragma unroll
                        for (uint32_t k = 0; k < 8u; k++)
                            if (r0 + q + k + origin < soft_cap) soft_z[r0 + q + k + origin] = xv[k];
                    }
                }
                znext = staged;
            }
            // a6: Mueller & Mueller steps with the 8-tap MMSE interpolator
            while (ii + 8u <= hi) {
                const int imu = (int)rintf(mu * 128.0f);
                const float4 t0 = *reinterpret_cast<const float4*>(&taps[imu * 8]);
                const float4 t1 = *reinterpret_cast<const float4*>(&taps[imu * 8 + 4]);
                const float* w = &row[ii & (kRingLen - 1)];        // 8 consecutive samples, no wrap
                float acc = 0.0f;
                acc = __builtin_fmaf(t0.x, w[7], acc);
                acc = __builtin_fmaf(t0.y, w[6], acc);
                acc = __builtin_fmaf(t0.z, w[5], acc);
                acc = __builtin_fmaf(t0.w, w[4], acc);
                acc = __builtin_fmaf(t1.x, w[3], acc);
                acc = __builtin_fmaf(t1.y, w[2], acc);
                acc = __builtin_fmaf(t1.z, w[1], acc);
                acc = __builtin_fmaf(t1.w, w[0], acc);
                const float o = acc;
                if (tap && n_chips + nc < soft_cap) soft_chips[n_chips + nc] = o;
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
                const uint64_t pos_bit = 1ull << (63u - nc);
                if (o > 0.0f) cw |= pos_bit;
                if ((step - 1u) & 1u) d_lo |= pos_bit;
                if ((step - 1u) & 2u) d_hi |= pos_bit;
                nc++;
            }

            // ---- phase 2: a7 over the tile's chips
            uint32_t c = 0, pos = ii_start;     // pos = window start of chip c (shifted coordinates)
            while (c < nc) {
                bool fin = false;
                if (s.state == 0 && s.preamble_cnt == 0) {
                    const uint32_t bit = (uint32_t)(cw >> (63u - c)) & 1u;
                    const uint32_t step = 1u + ((uint32_t)(d_lo >> (63u - c)) & 1u) + 2u * ((uint32_t)(d_hi >> (63u - c)) & 1u);
                    sink_search_chip(s, bit, pos + origin, th);
                    pos += step;
                    c++;
                    if (s.preamble_cnt == 1 && s.trigger >= rel_core_end) { done = true; break; }  // next lane's
                } else {
                    // shift in the chips up to the next symbol boundary (or the end of the tile)
                    const uint32_t need = 32u - (uint32_t)s.chip_cnt;
                    const uint32_t take = need < nc - c ? need : nc - c;
                    const uint64_t fld = (take == 64u) ? ~0ull : ~(~0ull >> take);     // top `take` bits
                    const uint64_t bits = ((cw << c) & fld) >> (64u - take);
                    s.shift = take >= 32u ? (uint32_t)bits : ((s.shift << take) | (uint32_t)bits);
                    pos += take + (uint32_t)__popcll((d_lo << c) & fld) + 2u * (uint32_t)__popcll((d_hi << c) & fld);
                    c += take;
                    s.chip_cnt += (int)take;
                    if (s.chip_cnt == 32) {
                        uint8_t* pb = (n_pk < K) ? stage[(size_t)g * K + n_pk].bytes : nullptr;
                        fin = sink_symbol(s, th, pb);
                    }
                }
                if (fin) {
                    if (s.trigger >= rel_core_start && s.trigger < rel_core_end) {
                        if (n_pk < K) {
                            snout_pkt* p = &stage[(size_t)g * K + n_pk];
                            const uint32_t len = (uint32_t)s.packetlen_cnt;
                            for (uint32_t b = len; b < 136u; b++) p->bytes[b] = 0;
                            p->sample_index = first_index + s0 + s.trigger;
                            p->proto = SNOUT_PROTO_ZIGBEE;
                            p->channel = slot_channel[slot];
                            p->len = (uint16_t)len;
                            const uint32_t scaled = (s.lqi / 8u) << 3;
                            p->lqi = (uint8_t)(scaled >= 256u ? 255u : scaled);
                            p->pdu_type = 0;
                            p->flags = 0;
                            p->aux = li;
                            // FCS: CRC-16 over all but the last two bytes == those two bytes (LE)
                            const uint32_t rx = s.b_prev | (s.b_last << 8);
                            p->crc_ok = (uint8_t)(len >= 3u && s.c2 == rx);
                        }
                        n_pk++;
                    }
                    enter_search(s);
                }
                // the sequential receiver stops once it is idle past the end of its core
                if (s.state == 0 && s.preamble_cnt == 0 && pos + origin >= rel_core_end) { done = true; break; }
            }
            n_chips += c;       // chips the sequential lane would have consumed so far
            if (!done && !RESUME && pos + origin >= rel_core_end) {
                // inside a frame at the end of the core: hand it to the second pass if this lane owns
                // it, drop it otherwise (never reported, and the lane would stop right after it)
                if (s.trigger >= rel_core_start && s.trigger < rel_core_end) {
                    ZbLaneSave o2;
                    o2.lp = lp; o2.mu = mu; o2.omega = omega; o2.last = last;
                    o2.g = g; o2.ii = ii; o2.znext = znext; o2.n_pk = n_pk; o2.n_chips = n_chips;
                    o2.s = s;
#pragma unroll
                    for (uint32_t k = 0; k < 16u; k++)
                        o2.zhist[k] = row[((znext - 16u + k) & (kRingLen - 1))];
                    saves[atomicAdd(n_saves, 1u)] = o2;
                }
                done = true;
            }
            if (hi >= avail && ii + 8u > avail) done = true;      // ran out of samples
        }
        __builtin_amdgcn_wave_barrier();    // one wave per workgroup: LDS is program-ordered
    }
    if (active) lane_cnt[g] = n_pk;
    if (tap && soft_n) *soft_n = n_chips;
}

// Ordered compaction of per-lane records: lane g holds min(lane_cnt[g], K) records.
__global__ __launch_bounds__(256) void zb_emit(const snout_pkt* __restrict__ stage,
                                               const uint32_t* __restrict__ lane_cnt, uint32_t K,
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
        cnt[k] = c < K ? c : K;
        s += cnt[k];
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
            if (off < out_cap) {
                const uint4* src = reinterpret_cast<const uint4*>(&stage[(size_t)(g0 + k) * K + i]);
                uint4* dst = reinterpret_cast<uint4*>(&out[off]);
#pragma unroll
                for (int q = 0; q < 10; q++) dst[q] = src[q];
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
                uint32_t warmup_)
{
    n_slots = n_slots_;
    threshold = threshold_;
    core = core_;
    warmup = warmup_;
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
        dcore = 1.0;
        for (uint32_t k = 0; k < core / 64u; k++) dcore = dcore * d64;
        dfirst = 1.0;
        for (uint32_t k = 0; k < (core - warmup) / 64u; k++) dfirst = dfirst * d64;
        if (int rc = d_iirw.ensure(64 * 8)) return rc;
        SNOUT_HIP(hipMemcpy(d_iirw.p, w, 64 * 8, hipMemcpyHostToDevice));
    }
    if (int rc = d_atan.ensure(257 * 4)) return rc;
    if (int rc = d_mmse.ensure(129 * 8 * 4)) return rc;
    if (int rc = d_slot_channel.ensure(n_slots * 2)) return rc;
    SNOUT_HIP(hipMemcpy(d_atan.p, atan_tab.data(), 257 * 4, hipMemcpyHostToDevice));
    SNOUT_HIP(hipMemcpy(d_mmse.p, kMmseTapsHost, 129 * 8 * 4, hipMemcpyHostToDevice));
    SNOUT_HIP(hipMemcpy(d_slot_channel.p, slot_channel_, n_slots * 2, hipMemcpyHostToDevice));
    return 0;
}

void ZbCtx::destroy()
{
    d_atan.release(); d_mmse.release(); d_slot_channel.release();
    d_d.release(); d_stage.release(); d_lane_cnt.release(); d_soft.release(); d_saves.release();
    d_iirw.release(); d_S.release(); d_Lblk.release(); d_lp_in.release();
}

int ZbCtx::reserve(uint64_t n)
{
    lanes_per_slot = cdiv(n, core);
    total_lanes = lanes_per_slot * n_slots;
    d_stride = n + 64;
    if (cdiv(total_lanes, 1024) > kMaxTiles) { set_last_error("too many lanes"); return SNOUT_ERANGE; }
    if (int rc = d_d.ensure(d_stride * n_slots * 4u)) return rc;
    if (int rc = d_stage.ensure((uint64_t)total_lanes * pkts_per_lane * sizeof(snout_pkt))) return rc;
    if (int rc = d_lane_cnt.ensure(((uint64_t)total_lanes + 1024u) * 4u)) return rc;
    max_out = total_lanes * pkts_per_lane;
    if (int rc = d_soft.ensure(((uint64_t)kSoftCap * 2u + 16u) * 4u)) return rc;
    if (int rc = d_saves.ensure((uint64_t)total_lanes * sizeof(ZbLaneSave))) return rc;
    nsb = (n + 63u) / 64u;
    if (int rc = d_S.ensure(nsb * n_slots * 8u)) return rc;
    if (int rc = d_Lblk.ensure((uint64_t)total_lanes * 8u)) return rc;
    if (int rc = d_lp_in.ensure((uint64_t)total_lanes * 8u)) return rc;
    return 0;
}

int ZbCtx::launch_lanes(uint64_t n, uint64_t first_index, hipStream_t st, int soft_lane)
{
    float* sz = soft_lane >= 0 ? d_soft.as<float>() : nullptr;
    float* sc = soft_lane >= 0 ? d_soft.as<float>() + kSoftCap : nullptr;
    uint32_t* sn = soft_lane >= 0 ? (uint32_t*)(d_soft.as<float>() + 2 * kSoftCap) : nullptr;
    uint32_t* n_saves = (uint32_t*)(d_soft.as<float>() + 2 * kSoftCap) + 4;
    SNOUT_HIP(hipMemsetAsync(n_saves, 0, 4, st));
    hipLaunchKernelGGL(zb_lanes<false>, dim3(cdiv(total_lanes, 64)), dim3(64), 0, st, d_d.as<float>(), n,
                       d_stride, lanes_per_slot, total_lanes, core, warmup, threshold,
                       d_slot_channel.as<uint16_t>(), first_index, d_mmse.as<float>(),
                       d_stage.as<snout_pkt>(), pkts_per_lane, d_lane_cnt.as<uint32_t>(), d_lp_in.as<double>(),
                       d_saves.as<ZbLaneSave>(), n_saves, sz, sc,
                       (uint32_t)(soft_lane >= 0 ? soft_lane : 0xFFFFFFFF), (uint32_t)kSoftCap, sn);
    // second pass: the lanes that stopped inside a frame (grid covers the worst case; waves beyond
    // the saved count exit at once)
    hipLaunchKernelGGL(zb_lanes<true>, dim3(cdiv(total_lanes, 64)), dim3(64), 0, st, d_d.as<float>(), n,
                       d_stride, lanes_per_slot, total_lanes, core, warmup, threshold,
                       d_slot_channel.as<uint16_t>(), first_index, d_mmse.as<float>(),
                       d_stage.as<snout_pkt>(), pkts_per_lane, d_lane_cnt.as<uint32_t>(), d_lp_in.as<double>(),
                       d_saves.as<ZbLaneSave>(), n_saves, sz, sc,
                       (uint32_t)(soft_lane >= 0 ? soft_lane : 0xFFFFFFFF), (uint32_t)kSoftCap, sn);
    SNOUT_HIP(hipGetLastError());
    return 0;
}

// Test tap: soft intermediates of one lane of the LAST processed segment (d is still resident).
int ZbCtx::soft(uint32_t stage_id, uint32_t lane, uint64_t n, float* out, uint64_t cap, uint64_t* n_out)
{
    *n_out = 0;
    if (stage_id == SNOUT_STAGE_ZB_DISCRIM) {
        // `lane` selects the channel slot here
        if (lane >= n_slots) return SNOUT_EINVAL;
        const uint64_t m = n < cap ? n : cap;
        SNOUT_HIP(hipMemcpy(out, d_d.as<float>() + (uint64_t)lane * d_stride, m * 4u, hipMemcpyDeviceToHost));
        *n_out = n;
        return n > cap ? SNOUT_EOVERFLOW : 0;
    }
    if (lane >= total_lanes) return SNOUT_EINVAL;
    if (int rc = launch_lanes(n, 0, nullptr, (int)lane)) return rc;
    SNOUT_HIP(hipDeviceSynchronize());
    uint32_t nch = 0;
    SNOUT_HIP(hipMemcpy(&nch, d_soft.as<float>() + 2 * kSoftCap, 4, hipMemcpyDeviceToHost));
    if (stage_id == SNOUT_STAGE_ZB_CHIPS) {
        const uint64_t have = nch < (uint32_t)kSoftCap ? nch : (uint32_t)kSoftCap;
        const uint64_t m = have < cap ? have : cap;
        SNOUT_HIP(hipMemcpy(out, d_soft.as<float>() + kSoftCap, m * 4u, hipMemcpyDeviceToHost));
        *n_out = have;
        return have > cap ? SNOUT_EOVERFLOW : 0;
    }
    if (stage_id == SNOUT_STAGE_ZB_DCREMOVED) {
        // z is defined for every sample the lane filtered; report up to the tap capacity
        const uint64_t li = lane % lanes_per_slot;
        const uint64_t cs = li * (uint64_t)core, s0 = cs > warmup ? cs - warmup : 0;
        uint64_t have = n > s0 ? n - s0 : 0;
        have = std::min<uint64_t>(have, (uint64_t)warmup + core);     // always filtered that far
        have = std::min<uint64_t>(have, kSoftCap);
        const uint64_t m = have < cap ? have : cap;
        SNOUT_HIP(hipMemcpy(out, d_soft.as<float>(), m * 4u, hipMemcpyDeviceToHost));
        *n_out = have;
        return have > cap ? SNOUT_EOVERFLOW : 0;
    }
    return SNOUT_EINVAL;
}

// iq: [n_slots][iq_stride] complex samples at 4 Msps per channel, device memory.  No host sync.
// totals (u32): [0] lanes  [1] packets  [2] lanes that held more than pkts_per_lane frames
int ZbCtx::enqueue(const float* d_iq, uint64_t n, uint64_t iq_stride, uint64_t first_index, hipStream_t st,
                   ResultSlot& s, bool time_front)
{
    if (int rc = s.d_out.ensure((uint64_t)max_out * sizeof(snout_pkt))) return rc;
    uint32_t* tot = s.d_totals.as<uint32_t>();
    uint32_t* sums = tot + 16;
    uint32_t* over = tot + 16 + kMaxTiles;
    if (time_front) SNOUT_HIP(hipEventRecord(s.ev_k0, st));
    const uint32_t gd = std::min<uint32_t>(cdiv(nsb * n_slots, 4), 256u * 16u);
    hipLaunchKernelGGL(zb_discrim, dim3(gd), dim3(256), 0, st, (const float2*)d_iq, n, iq_stride,
                       n_slots, nsb, d_atan.as<float>(), d_iirw.as<double>(), d_d.as<float>(), d_stride,
                       d_S.as<double>());
    hipLaunchKernelGGL(zb_iir_fold, dim3(cdiv(total_lanes, 256)), dim3(256), 0, st, d_S.as<double>(), nsb,
                       lanes_per_slot, total_lanes, core, warmup, d64, d_Lblk.as<double>());
    hipLaunchKernelGGL(zb_iir_scan, dim3(n_slots), dim3(64), 0, st, d_Lblk.as<double>(),
                       lanes_per_slot, n_slots, dfirst, dcore, d_lp_in.as<double>());
    if (int rc = launch_lanes(n, first_index, st, -1)) return rc;
    if (time_front) SNOUT_HIP(hipEventRecord(s.ev_k1, st));
    const uint32_t n_tiles = cdiv(total_lanes, kScanTile);
    launch_tile_reduce(d_lane_cnt.as<uint32_t>(), nullptr, total_lanes, total_lanes, pkts_per_lane,
                       sums, over, n_tiles, st);
    hipLaunchKernelGGL(zb_emit, dim3(n_tiles), dim3(256), 0, st, d_stage.as<snout_pkt>(),
                       d_lane_cnt.as<uint32_t>(), pkts_per_lane, total_lanes, sums, over, n_tiles, tot,
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
