// btle.hip — BTLE receive kernels for gfx950 (CDNA4, wave64).
//
// Replaces the inner loops of the `btle_rx` child the reference starts at snout/util/btle.py:53,63-69
// (algorithm restated in SURVEY Appendix A.1):
//   a1  search_unique_bits : sign-of-cross-product demod at 4 phases + 32-bit access-address match
//   a2  demod_byte / scramble_byte / crc_check : header, payload, de-whitening, CRC24
//
// Data layout in HBM
//   iq       : interleaved cf32, one capture segment, read exactly once by btle_demod_corr.
//   planes   : hard bits, 1 bit/sample, as [slot][iteration g][phase j] u64 words; bit l of word
//              (g,j) is the bit of channel-sample 256 g + 4 l + j, i.e. each word holds 64
//              consecutive SYMBOLS of one sampling phase.  A packet at one phase is therefore a
//              contiguous bit run of one plane -> decode reads <= 7 words.
//   hits     : per (slot, chunk) fixed-capacity lists of in-segment sample indices, ascending;
//              a prefix sum over the per-chunk counts yields the globally sorted candidate list
//              with no sort and no atomics (deterministic order).
//
// Roofline: btle_demod_corr is HBM-bound, 8 B read per complex sample (+1/64 warm-up re-read,
// +0.125 B/sample plane write). Everything after it is O(candidates).
#include "common.h"
#include <hip/hip_ext.h>
#include "iq_fmt.h"

namespace snout {

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

// 32-bit window of the 128-bit value {prev (older 64 symbols), cur (newer 64)} that ends at
// symbol `lane` of cur, oldest bit in the LSB: bits [33+lane, 64+lane] of (prev | cur << 64).
__device__ __forceinline__ uint32_t aa_window(uint64_t prev, uint64_t cur, uint32_t lane)
{
    const uint32_t w1 = (uint32_t)(prev >> 32), w2 = (uint32_t)cur, w3 = (uint32_t)(cur >> 32);
    const uint32_t sh = 33u + lane;           // 33..96
    const uint32_t q = sh >> 5;               // 1, 2 or 3
    const uint32_t lo = q == 1 ? w1 : (q == 2 ? w2 : w3);
    const uint32_t hi = q == 1 ? w2 : w3;     // q==3 only with shift 0: hi unused
    return __builtin_amdgcn_alignbit(hi, lo, sh & 31u);
}

// The same window by two 64-bit shifts of lane-uniform pairs and one select: 4 VALU + 4 SALU per
// phase instead of 8 VALU.  Used for integer input, where the per-row work is not hidden under the
// memory time (sc8 -4.5 %, sc16 -1..5 %; cf32 +-0, which keeps the form above).
__device__ __forceinline__ uint32_t aa_window64(uint64_t prev, uint64_t cur, uint32_t lane)
{
    const uint64_t mid = (prev >> 32) | (cur << 32);                  // symbols 32..95 of {prev, cur}: uniform
    const uint32_t a = (uint32_t)(mid >> ((1u + lane) & 63u));        // lanes 0..31: bits [1+l, 32+l] of mid
    const uint32_t b = (uint32_t)(cur >> ((lane - 31u) & 63u));       // lanes 32..63: bits [l-31, l] of cur
    return lane < 32u ? a : b;
}

// Append the matches of one 256-sample iteration to the chunk's hit list in ascending sample
// order.  hit[j] is this lane's match flag for sample 4*lane+j of the iteration.
__device__ __forceinline__ void append_hits(const bool hit[4], uint32_t lane, uint32_t n_base,
                                            uint32_t* __restrict__ list, uint32_t cap,
                                            uint32_t& cnt)
{
    uint64_t h[4];
#pragma unroll
    for (int j = 0; j < 4; j++) h[j] = __ballot(hit[j]);
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t rank = 0, total = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        rank += __popcll(h[j] & below);
        total += __popcll(h[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (hit[j]) {
            const uint32_t pos = cnt + rank;
            if (pos < cap) list[pos] = n_base + 4u * lane + (uint32_t)j;
        }
        rank += (uint32_t)((h[j] >> lane) & 1ull);
    }
    cnt += total;
}

// ---------------------------------------------------------------------------------------------
// a1: demodulate + correlate, single narrowband channel.  One wave per 16384-sample chunk.
// ---------------------------------------------------------------------------------------------
// One wave processes one 16384-sample chunk.  DEPTH rows of 2 KiB are kept in flight per wave:
// the loads of iteration r+DEPTH are issued before iteration r is processed.
// (Measured and dropped, MI355X, 1e9 samples: non-temporal loads -40 % (they defeat the L1 reuse of
//  the overlapping half of each lane's 64 B); taking the 4 following samples from the next lane by
//  cross-lane shuffle instead of the overlapping load: +-0; a persistent grid pulling chunks from
//  an atomic ticket: -5 %.  Software-pipelining the loads: +9 %.)
// Integer input (FMT sc8 / sc16, iq_fmt.h): the lane's 8 samples are 16 / 32 bytes; the bit is the
// same comparison on the converted values (sc8: in integers, every product is exact either way;
// sc16: in fp32 like the cf32 path, a 30-bit product rounds).
template <int FMT> struct K1Row {
    static constexpr uint32_t kLoads = FMT == kFmtCf32 ? 4u : (FMT == kFmtSc16 ? 2u : 1u);   // 16-byte loads per lane and row
    static constexpr uint32_t kLaneBytes = 4u * fmt_bytes(FMT), kRowBytes = 256u * fmt_bytes(FMT);
    u32x4 v[kLoads];
};

template <int FMT>
__device__ __forceinline__ void k1_bits(const K1Row<FMT>& q, bool b[4])
{
    if constexpr (FMT == kFmtCf32) {
        const f32x4 v0 = __builtin_bit_cast(f32x4, q.v[0]), v1 = __builtin_bit_cast(f32x4, q.v[1]);
        const f32x4 v2 = __builtin_bit_cast(f32x4, q.v[2]), v3 = __builtin_bit_cast(f32x4, q.v[3]);
        // bit[n] = (I[n]*Q[n+4]) > (I[n+4]*Q[n]); two roundings, no contraction
        b[0] = (v0.x * v2.y) > (v2.x * v0.y);
        b[1] = (v0.z * v2.w) > (v2.z * v0.w);
        b[2] = (v1.x * v3.y) > (v3.x * v1.y);
        b[3] = (v1.z * v3.w) > (v3.z * v1.w);
    } else if constexpr (FMT == kFmtSc8) {
        const uint32_t w[4] = {q.v[0].x, q.v[0].y, q.v[0].z, q.v[0].w};     // two samples per word
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t a = w[j >> 1] >> (16 * (j & 1)), c = w[2 + (j >> 1)] >> (16 * (j & 1));
            const int i0 = (int8_t)a, q0 = (int8_t)(a >> 8), i1 = (int8_t)c, q1 = (int8_t)(c >> 8);
            b[j] = i0 * q1 > i1 * q0;
        }
    } else {
        const uint32_t w[8] = {q.v[0].x, q.v[0].y, q.v[0].z, q.v[0].w, q.v[1].x, q.v[1].y, q.v[1].z, q.v[1].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float i0 = sc16_lo(w[j], 0), q0 = sc16_lo(w[j], 1), i1 = sc16_lo(w[4 + j], 0), q1 = sc16_lo(w[4 + j], 1);
            b[j] = (i0 * q1) > (i1 * q0);
        }
    }
}

template <int DEPTH, int FMT>
__global__ __launch_bounds__(256) void btle_demod_corr(
    const void* __restrict__ iq_all, uint64_t n_samples, uint64_t iq_stride, uint32_t aa,
    uint32_t n_chunks, uint32_t n_slots, uint64_t* __restrict__ planes_all, uint64_t plane_stride,
    uint32_t* __restrict__ chunk_cnt, uint32_t* __restrict__ chunk_hits, uint32_t cap)
{
    using Row = K1Row<FMT>;
    constexpr uint32_t kBps = fmt_bytes(FMT);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (item >= n_chunks * n_slots) return;
    const uint32_t slot = item / n_chunks, chunk = item - slot * n_chunks;
    const char* iq = reinterpret_cast<const char*>(iq_all) + (uint64_t)kBps * slot * iq_stride;
    uint64_t* planes = planes_all + (size_t)slot * plane_stride;
    uint32_t* list = chunk_hits + (size_t)item * cap;

    const uint64_t nb = n_samples - 4u;                       // bits exist for n in [0, nb)
    const uint32_t it0 = chunk * (uint32_t)kChunkIters;       // first iteration of this chunk
    const uint32_t itw = it0 - (chunk > 0 ? 1u : 0u);         // warm-up iteration (history only)
    const uint64_t base_sample = (uint64_t)itw * kIterSamples;
    // (range checks are per dword: an odd number of 2-byte samples is rounded up, the half dword past
    //  the end only reaches bit n_samples - 4, which nothing consumes)
    const uint64_t rem = ((n_samples - base_sample) * kBps + 3u) & ~3ull;
    const uint32_t recs = rem > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)rem;
    // wave-uniform descriptor; out-of-range reads return 0, so the tail needs no branches
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(iq + (uint64_t)kBps * base_sample), 0, (int)recs, 0x00020000);

    uint64_t prev[4] = {0, 0, 0, 0};
    uint64_t keep[4] = {0, 0, 0, 0};
    uint32_t cnt = 0;
    const uint32_t n_it = (uint32_t)kChunkIters + (it0 - itw);

    auto ld = [&](uint32_t off) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); };
    // samples 4l..4l+7 of a row: (I,Q) pairs; the upper half overlaps the next lane's lower half
    Row q[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; u++) {
        const uint32_t voff = (uint32_t)u * Row::kRowBytes + lane * Row::kLaneBytes;   // rows past n_it read as 0 or
#pragma unroll                                                                      // belong to the next chunk: unused
        for (uint32_t k = 0; k < Row::kLoads; k++) q[u].v[k] = u32x4{0u, 0u, 0u, 0u};
        // the warm-up row only feeds the 31-symbol history: its lower half is never looked at
        if (!(u == 0 && it0 != itw && lane < 32u)) {
#pragma unroll
            for (uint32_t k = 0; k < Row::kLoads; k++) q[u].v[k] = ld(voff + 16u * k);
        }
    }
    for (uint32_t r0 = 0; r0 < n_it; r0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; u++) {
            const uint32_t r = r0 + (uint32_t)u;
            if (r >= n_it) break;
            const Row cur_row = q[u];
            if (r + DEPTH < n_it) {
                const uint32_t voff = (r + DEPTH) * Row::kRowBytes + lane * Row::kLaneBytes;
#pragma unroll
                for (uint32_t k = 0; k < Row::kLoads; k++) q[u].v[k] = ld(voff + 16u * k);
            }
            bool b[4];
            k1_bits<FMT>(cur_row, b);
            uint64_t cur[4];
#pragma unroll
            for (int j = 0; j < 4; j++) cur[j] = __ballot(b[j]);

            const uint32_t it = itw + r;
            if (it >= it0) {
                const uint32_t n_base = it * (uint32_t)kIterSamples;
                bool hit[4];
                bool any = false;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    hit[j] = (FMT != kFmtCf32 ? aa_window64(prev[j], cur[j], lane) : aa_window(prev[j], cur[j], lane)) == aa;
                    any |= hit[j];
                }
                if (__ballot(any) != 0ull) {        // rare: the range tests are only paid here
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t n = n_base + 4u * lane + (uint32_t)j;
                        hit[j] = hit[j] && n >= 124u && n < nb;
                    }
                    append_hits(hit, lane, n_base, list, cap, cnt);
                }
                if (lane == it - it0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) keep[j] = cur[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) prev[j] = cur[j];
        }
    }
    // one coalesced store of the chunk's bit planes: lane l holds iteration it0+l
    if (lane < (uint32_t)kChunkIters) {
        uint64_t* dst = planes + ((size_t)it0 + lane) * 4u;
        reinterpret_cast<ulonglong2*>(dst)[0] = make_ulonglong2(keep[0], keep[1]);
        reinterpret_cast<ulonglong2*>(dst)[1] = make_ulonglong2(keep[2], keep[3]);
    }
    if (lane == 0) chunk_cnt[item] = cnt;
}

// ---------------------------------------------------------------------------------------------
// a1 for channelized input: correlate over bit planes already in HBM.  One wave per
// (slot, chunk); per iteration, lane l tests the window ending at symbol l of each phase.
// ---------------------------------------------------------------------------------------------
// (Round 6, measured and dropped: a grid-stride form -- 1024 / 2048 / 4096 workgroups, the next item's plane words requested
//  before the current ones are looked at -- 93 / 85 / 80 us against 80 us for one short wave per item: the kernel is bound
//  by its ~300 VALU instructions per 16 384 symbols x 4 phases, not by the rate its 97 656 waves are dispatched at.  What
//  doubled it in round 5 was the number of streams the handle had created, profiles/r6_streams.md.)
__global__ __launch_bounds__(256) void btle_corr_planes(
    const uint64_t* __restrict__ planes, uint64_t plane_stride, uint64_t nb, uint32_t aa,
    uint32_t n_chunks, uint32_t n_slots, uint32_t* __restrict__ chunk_cnt,
    uint32_t* __restrict__ chunk_hits, uint32_t cap)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wid >= n_chunks * n_slots) return;
    const uint32_t slot = wid / n_chunks, chunk = wid % n_chunks;
    const uint64_t* pl = planes + (size_t)slot * plane_stride;
    const uint32_t it0 = chunk * (uint32_t)kChunkIters;
    uint32_t cnt = 0;
    uint32_t* list = chunk_hits + (size_t)wid * cap;
    // The chunk's 64 x 4 plane words in one coalesced load: lane l holds iteration it0 + l, the
    // iteration before it comes from lane l-1 (lane 0: one uniform load).
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(pl + ((size_t)it0 + lane) * 4u);
    const ulonglong2 wa = src[0], wb = src[1];
    const uint64_t w[4] = {wa.x, wa.y, wb.x, wb.y};
    uint64_t pv[4];
    uint32_t cand = 0;                      // phases of this lane's iteration that may hold a match
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t up = __shfl_up(w[j], 1);
        pv[j] = lane ? up : (it0 > 0 ? pl[(size_t)(it0 - 1u) * 4u + j] : 0ull);
        // Bit-parallel pre-filter over the 64 symbol positions of the word: the newest 16 symbols of
        // the 32-symbol window must equal the top half of the access address.  T_i has, at bit l,
        // the symbol that window bit i of position l looks at.
        // In 32-bit words: S = {pv_lo, pv_hi, w_lo, w_hi}; T_i = S >> (33 + i) = two v_alignbit_b32 (the 64-bit shifts
        // of the plain form cost four times as much); mismatches T_i ^ (aa_i ? ~0 : 0) are OR-ed up, one v_bitop3_b32 each.
        const uint32_t s1 = (uint32_t)(pv[j] >> 32), s2 = (uint32_t)w[j], s3 = (uint32_t)(w[j] >> 32);
        uint32_t mm_lo = 0, mm_hi = 0;
#pragma unroll
        for (uint32_t i = 16; i < 32u; i++) {
            const uint32_t t_lo = i == 31u ? s2 : __builtin_amdgcn_alignbit(s2, s1, 1u + i);
            const uint32_t t_hi = i == 31u ? s3 : __builtin_amdgcn_alignbit(s3, s2, 1u + i);
            const uint32_t pbit = 0u - ((aa >> i) & 1u);        // uniform: all ones where the access address has a 1
            mm_lo = __builtin_amdgcn_bitop3_b32(mm_lo, t_lo, pbit, 0xF6);      // mm | (t ^ pbit) in one v_bitop3_b32
            mm_hi = __builtin_amdgcn_bitop3_b32(mm_hi, t_hi, pbit, 0xF6);
        }
        cand |= ((mm_lo & mm_hi) != 0xFFFFFFFFu ? 1u : 0u) << j;
    }
    // exact test (every lane one symbol position) only for the few iterations that passed
    uint64_t rows = __ballot(cand != 0u);
    while (rows) {
        const uint32_t r = (uint32_t)__builtin_ctzll(rows);
        rows &= rows - 1ull;
        const uint32_t it = it0 + r;
        bool hit[4];
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t cur = __shfl(w[j], (int)r), prv = __shfl(pv[j], (int)r);
            const uint64_t n = (uint64_t)it * kIterSamples + 4u * lane + (uint32_t)j;
            hit[j] = (aa_window(prv, cur, lane) == aa) && n >= 124u && n < nb;
            any |= hit[j];
        }
        if (__ballot(any) != 0ull)
            append_hits(hit, lane, it * (uint32_t)kIterSamples, list, cap, cnt);
    }
    if (lane == 0) chunk_cnt[wid] = cnt;
}

// ---------------------------------------------------------------------------------------------
// Block-level helpers (256 threads = 4 waves).
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(x, d);
        if ((int)lane >= d) x += t;
    }
    return x;
}

// Exclusive scan of one value per thread across the block; returns the prefix, *total = block sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t x, uint32_t* lds4, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(x, lane);
    __syncthreads();
    if (lane == 63) lds4[wv] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t s = lds4[w];
        if (w < wv) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - x;
}

// Sum of sums[0..count) computed by the whole block (count <= a few thousand).
__device__ __forceinline__ uint32_t block_sum_prefix(const uint32_t* __restrict__ sums,
                                                     uint32_t count, uint32_t* lds4)
{
    uint32_t x = 0;
    for (uint32_t i = threadIdx.x; i < count; i += kScanBlock) x += sums[i];
    uint32_t tot;
    (void)block_excl_scan(x, lds4, &tot);
    return tot;
}

// Per-tile sums of min(in[i], clamp) and a per-tile "some value exceeded clamp" flag.  n comes
// from device memory when n_ptr is set (the count of a previous stage).
__global__ __launch_bounds__(256) void tile_reduce(const uint32_t* __restrict__ in,
                                                   const uint32_t* __restrict__ n_ptr,
                                                   uint32_t n_fixed, uint32_t n_limit, uint32_t clamp,
                                                   uint32_t* __restrict__ tile_sums,
                                                   uint32_t* __restrict__ tile_over)
{
    __shared__ uint32_t lds4[4];
    uint32_t n = n_ptr ? *n_ptr : n_fixed;
    n = n < n_limit ? n : n_limit;
    const uint32_t i0 = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t s = 0, over = 0;
    if (i0 < n) {
        const uint4 v = *reinterpret_cast<const uint4*>(in + i0);   // buffers are padded to a tile
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t k = 0; k < kScanItems; k++) {
            if (i0 + k < n) {
                over |= x[k] > clamp ? 1u : 0u;
                s += x[k] < clamp ? x[k] : clamp;
            }
        }
    }
    uint32_t tot, tot_over;
    (void)block_excl_scan(s, lds4, &tot);
    (void)block_excl_scan(over, lds4, &tot_over);
    if (threadIdx.x == 0) {
        tile_sums[blockIdx.x] = tot;
        if (tile_over) tile_over[blockIdx.x] = tot_over;
    }
}

// ---------------------------------------------------------------------------------------------
// Per-(slot,chunk) hit lists -> one globally sorted candidate array (tile sums from tile_reduce).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void btle_flatten(
    const uint32_t* __restrict__ chunk_cnt, const uint32_t* __restrict__ chunk_hits, uint32_t cap,
    uint32_t n_lists, uint32_t n_chunks, const uint32_t* __restrict__ tile_sums, uint32_t n_tiles,
    const uint32_t* __restrict__ tile_over, uint32_t* __restrict__ hit_n,
    uint16_t* __restrict__ hit_slot, uint32_t max_cand, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t lds4[4];
    const uint32_t tile = blockIdx.x;
    const uint32_t tile_base = block_sum_prefix(tile_sums, tile, lds4);
    if (tile == 0) {
        const uint32_t all = block_sum_prefix(tile_sums, n_tiles, lds4);
        const uint32_t over = block_sum_prefix(tile_over, n_tiles, lds4);
        if (threadIdx.x == 0) { totals[0] = all; totals[2] = over; }
    }
    const uint32_t l0 = tile * kScanTile + threadIdx.x * kScanItems;
    uint32_t cnt[kScanItems];
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; k++) {
        uint32_t c = (l0 + k < n_lists) ? chunk_cnt[l0 + k] : 0u;
        cnt[k] = c < cap ? c : cap;
        s += cnt[k];
    }
    uint32_t tot;
    uint32_t off = tile_base + block_excl_scan(s, lds4, &tot);
    if (s == 0) return;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; k++) {
        const uint32_t list = l0 + k;
        for (uint32_t i = 0; i < cnt[k]; i++) {
            if (off < max_cand) {
                hit_n[off] = chunk_hits[(size_t)list * cap + i];
                hit_slot[off] = (uint16_t)(list / n_chunks);
            }
            off++;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// a2: decode every candidate, one thread each.  The packet is a contiguous run of <= 336 bits of
// one phase plane: 7 words are fetched up front, aligned, XORed with the whitening sequence.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void btle_decode(
    const uint64_t* __restrict__ planes, uint64_t plane_stride, uint64_t nb,
    const uint32_t* __restrict__ hit_n, const uint16_t* __restrict__ hit_slot,
    const uint32_t* __restrict__ totals, uint32_t max_cand,
    const uint64_t* __restrict__ whiten /* [n_slots][6] u64 words */,
    const uint16_t* __restrict__ slot_channel, uint32_t crc_init, SegBatch segs,
    BtleCand* __restrict__ cand, snout_pkt* __restrict__ stage)
{
    __shared__ uint32_t crc_tab[256];      // reflected CRC24 table (poly 0x00065B reflected = 0xDA6000)
    {
        uint32_t c = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c & 1u) ? ((c >> 1) ^ 0xDA6000u) : (c >> 1);
        crc_tab[threadIdx.x] = c;
    }
    __syncthreads();
    uint32_t n = totals[0];
    n = n < max_cand ? n : max_cand;
    const uint32_t crc0 = __brev(crc_init & 0xFFFFFFu) >> 8;     // register in reflected form
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += gridDim.x * blockDim.x) {
        const uint32_t n_hit = hit_n[idx];
        const uint32_t slot = hit_slot[idx];
        const uint64_t* pl = planes + (size_t)slot * plane_stride;
        const uint32_t j = n_hit & 3u;
        const uint64_t sym = (uint64_t)(n_hit >> 2) + 1u;     // first header symbol of phase j
        const uint64_t hdr = (uint64_t)n_hit + 4u;
        const uint64_t w0 = sym >> 6;
        const uint32_t sh = (uint32_t)(sym & 63u);
        uint64_t w[7];
#pragma unroll
        for (int k = 0; k < 7; k++) w[k] = pl[(w0 + k) * 4u + j];   // plane has a chunk of padding
        uint64_t p[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const uint64_t al = sh ? ((w[k] >> sh) | (w[k + 1] << (64u - sh))) : w[k];
            p[k] = al ^ whiten[slot * 6u + k];
        }
        BtleCand c;
        c.n_hit = n_hit;
        c.slot = (uint16_t)slot;
        c.accept = 0;
        if (!(hdr + 60u < nb)) {
            c.status = 1;
            c.next = n_hit + 1u;
        } else {
            const uint32_t b0 = (uint32_t)(p[0] & 0xFFu), b1 = (uint32_t)((p[0] >> 8) & 0xFFu);
            const uint32_t plen = b1 & 0x3Fu;
            c.next = (uint32_t)(hdr + 64u);
            c.status = 2;
            if (plen >= 6u && plen <= 37u) {
                const uint32_t total = plen + 5u;
                c.status = 3;
                if (hdr + 4u * (8u * (uint64_t)total - 1u) < nb) {
                    // CRC over header+payload, then compare with the 3 received bytes
                    uint32_t crc = crc0, rx = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
#pragma unroll
                        for (int t = 0; t < 8; t++) {
                            const uint32_t b = 8u * k + t;
                            const uint32_t v = (uint32_t)(p[k] >> (8 * t)) & 0xFFu;
                            if (b < plen + 2u) crc = (crc >> 8) ^ crc_tab[(crc ^ v) & 0xFFu];
                            else if (b < total) rx |= v << (8u * (b - plen - 2u));
                        }
                    }
                    // zero everything past the packet and write the 160-byte record
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const int32_t keep = (int32_t)total - 8 * k;     // bytes of word k in use
                        p[k] = keep >= 8 ? p[k] : (keep <= 0 ? 0ull : (p[k] & ((1ull << (8 * keep)) - 1ull)));
                    }
                    snout_pkt* o = &stage[idx];
                    uint4* o4 = reinterpret_cast<uint4*>(o);
                    const uint64_t si = segs.first[slot / segs.slots_per_seg] + (uint64_t)(n_hit - 124u);
                    const uint32_t ok = (rx == crc) ? 1u : 0u;
                    const uint32_t fl = ((b0 >> 6) & 1u) | (((b0 >> 7) & 1u) << 1);
                    o4[0] = make_uint4((uint32_t)si, (uint32_t)(si >> 32), SNOUT_PROTO_BTLE,
                                       (uint32_t)slot_channel[slot] | (total << 16));
                    o4[1] = make_uint4(ok | ((b0 & 0x0Fu) << 16) | (fl << 24), j,
                                       (uint32_t)p[0], (uint32_t)(p[0] >> 32));
                    o4[2] = make_uint4((uint32_t)p[1], (uint32_t)(p[1] >> 32), (uint32_t)p[2], (uint32_t)(p[2] >> 32));
                    o4[3] = make_uint4((uint32_t)p[3], (uint32_t)(p[3] >> 32), (uint32_t)p[4], (uint32_t)(p[4] >> 32));
                    o4[4] = make_uint4((uint32_t)p[5], (uint32_t)(p[5] >> 32), 0u, 0u);
#pragma unroll
                    for (int q = 5; q < 10; q++) o4[q] = make_uint4(0u, 0u, 0u, 0u);
                    c.status = 0;
                    c.next = (uint32_t)(hdr + 32u * total);
                }
            }
        }
        cand[idx] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// Sequential-search semantics in parallel: the reference search resumes after each examined
// packet, so a later match is only examined if its access address starts at or after the resume
// point.  Candidates further apart than the longest packet cannot influence each other, so the
// sorted list splits into independent clusters; each cluster head walks its own cluster.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void btle_resolve(const BtleCand* __restrict__ cand,
                                                    const uint32_t* __restrict__ totals,
                                                    uint32_t max_cand, SegBatch segs,
                                                    uint32_t* __restrict__ accept_flag)
{
    uint32_t n = totals[0];
    n = n < max_cand ? n : max_cand;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const BtleCand me = cand[i];
        bool head = (i == 0);
        if (!head) {
            const BtleCand pv = cand[i - 1];
            head = pv.slot != me.slot || (me.n_hit - pv.n_hit) >= (uint32_t)kBtleMaxSpan;
        }
        if (!head) continue;
        uint64_t resume = 0;
        uint32_t k = i;
        BtleCand c = me;
        uint32_t last_n = me.n_hit;
        // packets that start before the segment's min index were examined all the same (the search
        // resumes behind them) but belong to the capture segment before this one: not reported
        const uint32_t sg = me.slot / segs.slots_per_seg;
        const uint64_t first = segs.first[sg], min_index = segs.min_index[sg];
        while (true) {
            uint32_t acc = 0;
            if ((uint64_t)c.n_hit >= resume + 124u) {      // examined by the sequential search
                acc = (c.status == 0 && first + (uint64_t)(c.n_hit - 124u) >= min_index) ? 1u : 0u;
                resume = c.next;
            }
            accept_flag[k] = acc;
            k++;
            if (k >= n) break;
            c = cand[k];
            if (c.slot != me.slot || (c.n_hit - last_n) >= (uint32_t)kBtleMaxSpan) break;
            last_n = c.n_hit;
        }
    }
}

// Ordered compaction of the accepted records (tile sums from tile_reduce over accept flags).
__global__ __launch_bounds__(256) void btle_emit(const snout_pkt* __restrict__ stage,
                                                 const uint32_t* __restrict__ accept_flag,
                                                 const uint32_t* __restrict__ tile_sums,
                                                 uint32_t* __restrict__ totals, uint32_t max_cand,
                                                 snout_pkt* __restrict__ out, uint32_t out_cap)
{
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t src_of[kScanTile];     // tile-local compact list: source candidate index
    uint32_t n = totals[0];
    n = n < max_cand ? n : max_cand;
    const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
    const uint32_t tile = blockIdx.x;
    if (tile == 0) {
        const uint32_t all = block_sum_prefix(tile_sums, n_tiles, lds4);
        if (threadIdx.x == 0) totals[1] = all;
    }
    if (tile >= n_tiles) return;
    const uint32_t tile_base = block_sum_prefix(tile_sums, tile, lds4);
    const uint32_t i0 = tile * kScanTile + threadIdx.x * kScanItems;
    uint32_t fl[kScanItems];
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; k++) {
        fl[k] = (i0 + k < n) ? accept_flag[i0 + k] : 0u;
        s += fl[k];
    }
    uint32_t tot;
    uint32_t off = block_excl_scan(s, lds4, &tot);
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; k++)
        if (fl[k]) src_of[off++] = i0 + k;
    __syncthreads();
    // all threads move 16-byte pieces: record r of the tile, piece q
    for (uint32_t g = threadIdx.x; g < tot * 10u; g += kScanBlock) {
        const uint32_t r = g / 10u, q = g % 10u;
        const uint32_t o = tile_base + r;
        if (o < out_cap)
            reinterpret_cast<uint4*>(&out[o])[q] = reinterpret_cast<const uint4*>(&stage[src_of[r]])[q];
    }
}

}  // namespace snout

// =============================================================================================
// Host side: workspace + launch sequence (all on one stream, no host sync until the end).
// =============================================================================================
namespace snout {

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

void launch_tile_reduce(const uint32_t* in, const uint32_t* n_ptr, uint32_t n_fixed, uint32_t n_limit,
                        uint32_t clamp, uint32_t* tile_sums, uint32_t* tile_over, uint32_t n_tiles,
                        hipStream_t st)
{
    hipLaunchKernelGGL(tile_reduce, dim3(n_tiles), dim3(256), 0, st, in, n_ptr, n_fixed, n_limit, clamp,
                       tile_sums, tile_over);
}

int BtleCtx::init(uint32_t seg_slots_, const uint16_t* slot_channel_, uint32_t aa_, uint32_t crc_init_,
                  uint32_t max_hits_, uint32_t batch_cap_)
{
    seg_slots = seg_slots_;
    batch_cap = batch_cap_ ? batch_cap_ : 1u;
    n_slots = seg_slots * batch_cap;                // tables for every slot of a full batch
    aa = aa_;
    crc_init = crc_init_;
    max_hits_cfg = max_hits_;
    if (const char* e = getenv("SNOUT_K1_DEPTH")) variant = (uint32_t)atoi(e);
    // whitening sequences (LFSR x^7+x^4+1, position 0 = 1, positions 1..6 = channel MSB..LSB),
    // 48 bytes per slot packed LSB-first into six u64 words
    std::vector<uint64_t> wh(6u * n_slots, 0ull);
    std::vector<uint16_t> ch(n_slots);
    for (uint32_t s = 0; s < n_slots; s++) {
        ch[s] = slot_channel_[s % seg_slots];       // the slots of every segment of a batch carry the same channels
        uint32_t reg = 1u;                                  // bit i = position i
        for (int i = 0; i < 6; i++) reg |= ((ch[s] >> (5 - i)) & 1u) << (1 + i);
        for (int b = 0; b < 384; b++) {
            const uint32_t o = (reg >> 6) & 1u;
            wh[s * 6u + (b >> 6)] |= (uint64_t)o << (b & 63);
            reg = ((reg << 1) & 0x7Fu) | o;                 // shift, feed back into position 0
            reg ^= o << 4;                                  // and into position 4
        }
    }
    if (int rc = d_whiten.ensure(wh.size() * 8)) return rc;
    if (int rc = d_slot_channel.ensure(ch.size() * 2)) return rc;
    SNOUT_HIP(hipMemcpy(d_whiten.p, wh.data(), wh.size() * 8, hipMemcpyHostToDevice));
    SNOUT_HIP(hipMemcpy(d_slot_channel.p, ch.data(), ch.size() * 2, hipMemcpyHostToDevice));
    n_slots = seg_slots;                            // slots of the current call (reserve)
    return 0;
}

void BtleCtx::destroy()
{
    d_planes.release(); d_chunk_cnt.release(); d_chunk_hits.release(); d_hit_n.release();
    d_hit_slot.release(); d_cand.release(); d_stage.release(); d_accept.release();
    d_whiten.release(); d_slot_channel.release();
}

// Size every buffer for `n` channel-samples per slot.
int BtleCtx::reserve(uint64_t n, uint32_t segs)
{
    if (segs == 0 || segs > batch_cap) { set_last_error("batch of %u segments (handle created for %u)", segs, batch_cap); return SNOUT_EINVAL; }
    n_slots = seg_slots * segs;                     // slot = (segment of the batch, channel)
    n_chunks = cdiv(n, kChunkSamples);
    plane_stride = (uint64_t)(n_chunks + 1) * kChunkIters * 4u;     // u64 words per slot (+1 chunk pad)
    const uint64_t lists = (uint64_t)n_chunks * n_slots;
    const uint32_t auto_cand = (uint32_t)std::min<uint64_t>(n * n_slots / 1024u + 4096u, 1u << 26);
    max_cand = std::max(max_cand_grown, max_hits_cfg ? max_hits_cfg : auto_cand);
    if (cdiv(lists, kScanTile) > kMaxTiles || cdiv(max_cand, kScanTile) > kMaxTiles) {
        set_last_error("segment too large for the tile-sum tables");
        return SNOUT_ERANGE;
    }
    {
        const void* before = d_planes.p;
        if (int rc = d_planes.ensure(plane_stride * n_slots * 8u)) return rc;
        // words never written (tail of the last chunk, padding) must read as zero bits
        if (d_planes.p != before) SNOUT_HIP(hipMemset(d_planes.p, 0, d_planes.cap));
    }
    if (int rc = d_chunk_cnt.ensure((lists + kScanTile) * 4u)) return rc;
    if (int rc = d_chunk_hits.ensure(lists * hit_cap * 4u)) return rc;
    if (int rc = d_hit_n.ensure((uint64_t)max_cand * 4u)) return rc;
    if (int rc = d_hit_slot.ensure((uint64_t)max_cand * 2u)) return rc;
    if (int rc = d_cand.ensure((uint64_t)max_cand * sizeof(BtleCand))) return rc;
    if (int rc = d_stage.ensure((uint64_t)max_cand * sizeof(snout_pkt))) return rc;
    if (int rc = d_accept.ensure(((uint64_t)max_cand + kScanTile) * 4u)) return rc;
    return 0;
}

// Narrowband front end: iq (device) -> planes + per-chunk hit lists.
int BtleCtx::launch_demod_corr(const void* d_iq, uint64_t n, uint64_t iq_stride, hipStream_t st,
                                ResultSlot* timing, int fmt)
{
    const uint32_t total = n_chunks * n_slots;
    if (timing) SNOUT_HIP(hipEventRecord(timing->ev_k0, st));
#define SNOUT_K1(D, F)                                                                               \
    hipLaunchKernelGGL((btle_demod_corr<D, F>), dim3(cdiv(total, 4)), dim3(256), 0, st, d_iq, n, iq_stride, \
                       aa, n_chunks, n_slots, d_planes.as<uint64_t>(), plane_stride,                  \
                       d_chunk_cnt.as<uint32_t>(), d_chunk_hits.as<uint32_t>(), hit_cap)
    if (fmt == kFmtSc8) {
        SNOUT_K1(4, kFmtSc8);   // 16 B per lane and row: rows in flight 1 / 2 / 4 / 8 -> 0.44 / 0.46 / 0.42 / 0.42 ms
    } else if (fmt == kFmtSc16) {
        SNOUT_K1(2, kFmtSc16);  // 1 / 2 / 4 / 8 -> 0.79 / 0.775 / 0.80 / 0.77 ms
    } else switch (variant) {   // prefetch depth; SNOUT_K1_DEPTH overrides for experiments
        case 3: SNOUT_K1(3, kFmtCf32); break;
        case 4: SNOUT_K1(4, kFmtCf32); break;
        case 2: SNOUT_K1(2, kFmtCf32); break;
        default: SNOUT_K1(1, kFmtCf32); break;
    }
#undef SNOUT_K1
    if (timing) SNOUT_HIP(hipEventRecord(timing->ev_k1, st));
    SNOUT_HIP(hipGetLastError());
    return 0;
}

// Wideband front end already filled planes: correlate them.
int BtleCtx::launch_corr_planes(uint64_t n, hipStream_t st, hipEvent_t ev_stop)
{
    hipExtLaunchKernelGGL(btle_corr_planes, dim3(cdiv((uint64_t)n_chunks * n_slots, 4)), dim3(256), 0, st, nullptr, ev_stop, 0,
                          (const uint64_t*)d_planes.as<uint64_t>(), plane_stride, n - 4u, aa, n_chunks, n_slots,
                          d_chunk_cnt.as<uint32_t>(), d_chunk_hits.as<uint32_t>(), hit_cap);
    SNOUT_HIP(hipGetLastError());
    return 0;
}

// Hit lists -> resolved, ordered packet records in s.d_out; counts in s.d_totals.  No host sync.
// totals (u32): [0] candidates  [1] packets  [2] chunks whose list overflowed
//               [16 ..) list tile sums, [16+kMaxTiles ..) accept tile sums, [16+2 kMaxTiles ..) overflow
int BtleCtx::enqueue_tail(uint64_t n, const SegBatch& segs_in, hipStream_t st, ResultSlot& s)
{
    SegBatch segs = segs_in;
    segs.slots_per_seg = seg_slots;
    if (int rc = s.d_out.ensure((uint64_t)max_cand * sizeof(snout_pkt))) return rc;
    const uint32_t lists = n_chunks * n_slots;
    uint32_t* tot = s.d_totals.as<uint32_t>();
    uint32_t* list_tiles = tot + 16;
    uint32_t* acc_tiles = tot + 16 + kMaxTiles;
    uint32_t* over_tiles = tot + 16 + 2 * kMaxTiles;
    const uint32_t n_list_tiles = cdiv(lists, kScanTile);
    const uint32_t n_cand_tiles = cdiv(max_cand, kScanTile);
    hipLaunchKernelGGL(tile_reduce, dim3(n_list_tiles), dim3(256), 0, st, d_chunk_cnt.as<uint32_t>(),
                       (const uint32_t*)nullptr, lists, lists, hit_cap, list_tiles, over_tiles);
    hipLaunchKernelGGL(btle_flatten, dim3(n_list_tiles), dim3(256), 0, st, d_chunk_cnt.as<uint32_t>(),
                       d_chunk_hits.as<uint32_t>(), hit_cap, lists, n_chunks, list_tiles, n_list_tiles,
                       over_tiles, d_hit_n.as<uint32_t>(), d_hit_slot.as<uint16_t>(), max_cand, tot);
    const uint32_t g = std::min<uint32_t>(cdiv(max_cand, 256), 1024u);
    hipLaunchKernelGGL(btle_decode, dim3(g), dim3(256), 0, st, d_planes.as<uint64_t>(), plane_stride,
                       n - 4u, d_hit_n.as<uint32_t>(), d_hit_slot.as<uint16_t>(), tot, max_cand,
                       d_whiten.as<uint64_t>(), d_slot_channel.as<uint16_t>(), crc_init, segs,
                       d_cand.as<BtleCand>(), d_stage.as<snout_pkt>());
    hipLaunchKernelGGL(btle_resolve, dim3(g), dim3(256), 0, st, d_cand.as<BtleCand>(), tot, max_cand, segs,
                       d_accept.as<uint32_t>());
    hipLaunchKernelGGL(tile_reduce, dim3(n_cand_tiles), dim3(256), 0, st, d_accept.as<uint32_t>(),
                       tot, 0u, max_cand, 1u, acc_tiles, (uint32_t*)nullptr);
    hipLaunchKernelGGL(btle_emit, dim3(n_cand_tiles), dim3(256), 0, st,
                       d_stage.as<snout_pkt>(), d_accept.as<uint32_t>(), acc_tiles, tot, max_cand,
                       s.d_out.as<snout_pkt>(), max_cand);
    SNOUT_HIP(hipGetLastError());
    return 0;
}

bool BtleCtx::check_overflow(const ResultSlot& s)
{
    last_n_cand = s.h_totals[0];
    overflow_chunk = s.h_totals[2] != 0;      // some chunk list was longer than hit_cap
    overflow_cand = s.h_totals[0] > max_cand;
    if (!(overflow_chunk || overflow_cand)) return false;
    // more hits than provisioned: grow; the caller runs the segment again (never truncated)
    set_last_error("BTLE hit capacity exceeded: %u candidates, per-chunk cap %u, max_cand %u",
                   s.h_totals[0], hit_cap, max_cand);
    if (overflow_chunk) hit_cap = std::min<uint32_t>(hit_cap * 4u, kChunkSamples);
    if (overflow_cand) max_cand_grown = max_cand * 4u;
    return true;
}

}  // namespace snout
