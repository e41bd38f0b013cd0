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
__global__ __launch_bounds__(256) void btle_demod_corr(
    const float* __restrict__ iq, uint64_t n_samples, uint32_t aa, uint32_t n_chunks,
    uint64_t* __restrict__ planes, uint32_t* __restrict__ chunk_cnt,
    uint32_t* __restrict__ chunk_hits, uint32_t cap)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunk = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (chunk >= n_chunks) return;
    const uint64_t nb = n_samples - 4u;                       // bits exist for n in [0, nb)
    const uint32_t it0 = chunk * (uint32_t)kChunkIters;       // first iteration of this chunk
    const uint32_t itw = it0 - (chunk > 0 ? 1u : 0u);         // warm-up iteration (history only)
    const uint64_t base_sample = (uint64_t)itw * kIterSamples;
    const uint64_t rem = (n_samples - base_sample) * 8ull;
    const uint32_t recs = rem > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)rem;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(iq + 2ull * base_sample), 0, (int)recs, 0x00020000);

    uint64_t prev[4] = {0, 0, 0, 0};
    uint64_t keep[4] = {0, 0, 0, 0};
    uint32_t cnt = 0;
    uint32_t* list = chunk_hits + (size_t)chunk * cap;
    const uint32_t n_it = (uint32_t)kChunkIters + (it0 - itw);

    for (uint32_t r = 0; r < n_it; r++) {
        const uint32_t voff = r * 2048u + lane * 32u;
        // samples 4l..4l+7 of this iteration: (I,Q) pairs, out-of-range reads return 0
        const f32x4 v0 = buf_load16(rs, voff);
        const f32x4 v1 = buf_load16(rs, voff + 16u);
        const f32x4 v2 = buf_load16(rs, voff + 32u);
        const f32x4 v3 = buf_load16(rs, voff + 48u);
        // bit[n] = (I[n]*Q[n+4]) > (I[n+4]*Q[n]); two roundings, no contraction
        bool b[4];
        b[0] = (v0.x * v2.y) > (v2.x * v0.y);
        b[1] = (v0.z * v2.w) > (v2.z * v0.w);
        b[2] = (v1.x * v3.y) > (v3.x * v1.y);
        b[3] = (v1.z * v3.w) > (v3.z * v1.w);
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
                const uint32_t n = n_base + 4u * lane + (uint32_t)j;
                hit[j] = (aa_window(prev[j], cur[j], lane) == aa) && n >= 124u && n < nb;
                any |= hit[j];
            }
            if (__ballot(any) != 0ull) append_hits(hit, lane, n_base, list, cap, cnt);
            if (lane == it - it0) {
#pragma unroll
                for (int j = 0; j < 4; j++) keep[j] = cur[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) prev[j] = cur[j];
    }
    // one coalesced 2 KiB store of the chunk's bit planes: lane l holds iteration it0+l
    uint64_t* dst = planes + ((size_t)it0 + lane) * 4u;
    reinterpret_cast<ulonglong2*>(dst)[0] = make_ulonglong2(keep[0], keep[1]);
    reinterpret_cast<ulonglong2*>(dst)[1] = make_ulonglong2(keep[2], keep[3]);
    if (lane == 0) chunk_cnt[chunk] = cnt;
}

// ---------------------------------------------------------------------------------------------
// a1 for channelized input: correlate over bit planes already in HBM.  One wave per
// (slot, chunk); per iteration, lane l tests the window ending at symbol l of each phase.
// ---------------------------------------------------------------------------------------------
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
    // iterate over the 64 iterations of the chunk; every lane tests symbol `lane` of each
    for (uint32_t r = 0; r < (uint32_t)kChunkIters; r++) {
        const uint32_t it = it0 + r;
        bool hit[4];
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t cur = pl[(size_t)it * 4u + j];
            const uint64_t prv = it > 0 ? pl[(size_t)(it - 1) * 4u + j] : 0ull;
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
// Exclusive prefix sum of min(in[i], clamp) over n items, one workgroup.  total -> *total_out.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_u32(const uint32_t* __restrict__ in,
                                                 uint32_t* __restrict__ out,
                                                 const uint32_t* __restrict__ n_ptr, uint32_t n_fixed,
                                                 uint32_t clamp, uint32_t* __restrict__ total_out)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const uint32_t n = n_ptr ? *n_ptr : n_fixed;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    constexpr uint32_t kPer = 8;
    for (uint32_t base = 0; base < n; base += 1024u * kPer) {
        uint32_t v[kPer];
        uint32_t s = 0;
        const uint32_t i0 = base + tid * kPer;
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            uint32_t x = (i0 + k < n) ? in[i0 + k] : 0u;
            x = x < clamp ? x : clamp;
            v[k] = s;
            s += x;
        }
        // inclusive scan of s across the wave
        uint32_t inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(inc, d);
            if ((int)lane >= d) inc += t;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wv; w++) wbase += wsum[w];
        const uint32_t carry = carry_s;
        const uint32_t excl = carry + wbase + inc - s;
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++)
            if (i0 + k < n) out[i0 + k] = excl + v[k];
        __syncthreads();
        if (tid == 1023) carry_s = carry + wbase + inc;
        __syncthreads();
    }
    if (tid == 0) *total_out = carry_s;
}

// ---------------------------------------------------------------------------------------------
// a2: decode every candidate.  One thread per (slot, chunk) hit list.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t plane_byte(const uint64_t* __restrict__ pl, uint32_t j,
                                               uint64_t sym)
{
    const uint64_t w = sym >> 6;
    const uint32_t s = (uint32_t)(sym & 63u);
    uint64_t v = pl[w * 4u + j] >> s;
    if (s > 56u) v |= pl[(w + 1u) * 4u + j] << (64u - s);
    return (uint32_t)(v & 0xFFu);
}

__device__ __forceinline__ uint32_t crc24_update(uint32_t r, uint32_t byte)
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t t = (r >> 23) & 1u;
        r = (r << 1) & 0xFFFFFFu;
        if (t != ((byte >> k) & 1u)) r ^= 0x00065Bu;
    }
    return r;
}

__global__ __launch_bounds__(256) void btle_decode(
    const uint64_t* __restrict__ planes, uint64_t plane_stride, uint64_t nb,
    const uint32_t* __restrict__ chunk_cnt, const uint32_t* __restrict__ chunk_off,
    const uint32_t* __restrict__ chunk_hits, uint32_t cap, uint32_t n_chunks, uint32_t n_slots,
    const uint8_t* __restrict__ whiten /* [n_slots][42] */,
    const uint16_t* __restrict__ slot_channel, uint32_t crc_init, uint64_t first_index,
    BtleCand* __restrict__ cand, snout_pkt* __restrict__ stage, uint32_t max_cand)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_chunks * n_slots;
         t += gridDim.x * blockDim.x) {
    uint32_t cnt = chunk_cnt[t];
    cnt = cnt < cap ? cnt : cap;
    if (cnt == 0) continue;
    const uint32_t slot = t / n_chunks;
    const uint64_t* pl = planes + (size_t)slot * plane_stride;
    const uint8_t* wh = whiten + slot * 42u;
    const uint32_t off = chunk_off[t];
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t idx = off + k;
        if (idx >= max_cand) break;
        const uint32_t n_hit = chunk_hits[(size_t)t * cap + k];
        const uint32_t j = n_hit & 3u;
        const uint64_t sym = (uint64_t)(n_hit >> 2) + 1u;   // first header symbol, phase j
        const uint64_t hdr = (uint64_t)n_hit + 4u;
        BtleCand c;
        c.n_hit = n_hit;
        c.slot = (uint16_t)slot;
        c.accept = 0;
        snout_pkt* p = &stage[idx];
        if (!(hdr + 60u < nb)) {
            c.status = 1;
            c.next = n_hit + 1u;
        } else {
            const uint32_t b0 = plane_byte(pl, j, sym) ^ wh[0];
            const uint32_t b1 = plane_byte(pl, j, sym + 8u) ^ wh[1];
            const uint32_t plen = b1 & 0x3Fu;
            c.next = (uint32_t)(hdr + 64u);
            if (plen < 6u || plen > 37u) {
                c.status = 2;
            } else {
                const uint32_t total = plen + 5u;
                if (!(hdr + 4u * (8u * (uint64_t)total - 1u) < nb)) {
                    c.status = 3;
                } else {
                    uint32_t crc = crc24_update(crc24_update(crc_init & 0xFFFFFFu, b0), b1);
                    p->bytes[0] = (uint8_t)b0;
                    p->bytes[1] = (uint8_t)b1;
                    uint32_t rx_crc = 0;    // received CRC bits in register order
                    for (uint32_t b = 2; b < total; b++) {
                        const uint32_t v = plane_byte(pl, j, sym + 8u * b) ^ wh[b];
                        p->bytes[b] = (uint8_t)v;
                        if (b < total - 3u) crc = crc24_update(crc, v);
                        else rx_crc = (rx_crc << 8) | (__brev(v) >> 24);
                    }
                    for (uint32_t b = total; b < 136u; b++) p->bytes[b] = 0;
                    p->sample_index = first_index + (uint64_t)(n_hit - 124u);
                    p->proto = SNOUT_PROTO_BTLE;
                    p->channel = slot_channel[slot];
                    p->len = (uint16_t)total;
                    p->crc_ok = (uint8_t)(rx_crc == crc);
                    p->lqi = 0;
                    p->pdu_type = (uint8_t)(b0 & 0x0Fu);
                    p->flags = (uint8_t)(((b0 >> 6) & 1u) | (((b0 >> 7) & 1u) << 1));
                    p->aux = j;
                    c.status = 0;
                    c.next = (uint32_t)(hdr + 32u * total);
                }
            }
        }
        cand[idx] = c;
    }
    }
}

// ---------------------------------------------------------------------------------------------
// Sequential-search semantics in parallel: the reference search resumes after each examined
// packet, so a later match is only examined if its access address starts at or after the resume
// point.  Candidates further apart than the longest packet cannot influence each other, so the
// sorted list splits into independent clusters; each cluster head walks its own cluster.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void btle_resolve(BtleCand* __restrict__ cand,
                                                    const uint32_t* __restrict__ n_cand_ptr,
                                                    uint32_t max_cand,
                                                    uint32_t* __restrict__ accept_flag)
{
    uint32_t n = *n_cand_ptr;
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
    while (true) {
        uint32_t acc = 0;
        if ((uint64_t)c.n_hit >= resume + 124u) {      // examined by the sequential search
            acc = c.status == 0 ? 1u : 0u;
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

__global__ __launch_bounds__(256) void emit_packets(const snout_pkt* __restrict__ stage,
                                                    const uint32_t* __restrict__ accept_flag,
                                                    const uint32_t* __restrict__ out_off,
                                                    const uint32_t* __restrict__ n_cand_ptr,
                                                    uint32_t max_cand, snout_pkt* __restrict__ out,
                                                    uint32_t out_cap)
{
    uint32_t n = *n_cand_ptr;
    n = n < max_cand ? n : max_cand;
    // 10 threads move one 160-byte record as 16-byte pieces
    for (uint64_t g = blockIdx.x * blockDim.x + threadIdx.x; g < (uint64_t)n * 10u;
         g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(g / 10u), piece = (uint32_t)(g % 10u);
        if (!accept_flag[i]) continue;
        const uint32_t o = out_off[i];
        if (o >= out_cap) continue;
        reinterpret_cast<uint4*>(&out[o])[piece] = reinterpret_cast<const uint4*>(&stage[i])[piece];
    }
}

}  // namespace snout

// =============================================================================================
// Host side: workspace + launch sequence (all on one stream, no host sync until the end).
// =============================================================================================
namespace snout {

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

int BtleCtx::init(uint32_t n_slots_, const uint16_t* slot_channel_, uint32_t aa_, uint32_t crc_init_,
                  uint32_t max_hits_)
{
    n_slots = n_slots_;
    aa = aa_;
    crc_init = crc_init_;
    max_hits_cfg = max_hits_;
    // whitening sequences (LFSR x^7+x^4+1, position 0 = 1, positions 1..6 = channel MSB..LSB)
    std::vector<uint8_t> wh(42u * n_slots);
    std::vector<uint16_t> ch(n_slots);
    for (uint32_t s = 0; s < n_slots; s++) {
        ch[s] = slot_channel_[s];
        uint32_t reg = 1u;                                  // bit i = position i
        for (int i = 0; i < 6; i++) reg |= ((ch[s] >> (5 - i)) & 1u) << (1 + i);
        for (int b = 0; b < 42; b++) {
            uint8_t v = 0;
            for (int k = 0; k < 8; k++) {
                const uint32_t o = (reg >> 6) & 1u;
                v |= (uint8_t)(o << k);
                reg = ((reg << 1) & 0x7Fu) | o;             // shift, feed back into position 0
                reg ^= o << 4;                              // and into position 4
            }
            wh[s * 42u + b] = v;
        }
    }
    if (int rc = d_whiten.ensure(wh.size())) return rc;
    if (int rc = d_slot_channel.ensure(ch.size() * 2)) return rc;
    if (int rc = d_totals.ensure(64)) return rc;
    SNOUT_HIP(hipMemcpy(d_whiten.p, wh.data(), wh.size(), hipMemcpyHostToDevice));
    SNOUT_HIP(hipMemcpy(d_slot_channel.p, ch.data(), ch.size() * 2, hipMemcpyHostToDevice));
    SNOUT_HIP(hipHostMalloc((void**)&h_totals, 64, hipHostMallocDefault));
    SNOUT_HIP(hipEventCreate(&ev_t0));
    SNOUT_HIP(hipEventCreate(&ev_k0));
    SNOUT_HIP(hipEventCreate(&ev_k1));
    SNOUT_HIP(hipEventCreate(&ev_t1));
    return 0;
}

void BtleCtx::destroy()
{
    d_planes.release(); d_chunk_cnt.release(); d_chunk_off.release(); d_chunk_hits.release();
    d_cand.release(); d_stage.release(); d_accept.release(); d_out_off.release(); d_out.release();
    d_whiten.release(); d_slot_channel.release(); d_totals.release();
    if (h_totals) (void)hipHostFree(h_totals);
    if (h_out) (void)hipHostFree(h_out);
    h_totals = nullptr; h_out = nullptr; h_out_cap = 0;
    if (ev_t0) { (void)hipEventDestroy(ev_t0); (void)hipEventDestroy(ev_k0);
                 (void)hipEventDestroy(ev_k1); (void)hipEventDestroy(ev_t1); ev_t0 = nullptr; }
}

// Size every buffer for `n` channel-samples per slot.
int BtleCtx::reserve(uint64_t n)
{
    n_chunks = cdiv(n, kChunkSamples);
    plane_stride = (uint64_t)(n_chunks + 1) * kChunkIters * 4u;     // u64 words per slot (+1 chunk pad)
    const uint64_t lists = (uint64_t)n_chunks * n_slots;
    max_cand = max_hits_cfg ? max_hits_cfg : (uint32_t)std::min<uint64_t>(n * n_slots / 1024u + 4096u, 1u << 26);
    if (int rc = d_planes.ensure(plane_stride * n_slots * 8u)) return rc;
    if (int rc = d_chunk_cnt.ensure(lists * 4u)) return rc;
    if (int rc = d_chunk_off.ensure(lists * 4u)) return rc;
    if (int rc = d_chunk_hits.ensure(lists * kChunkHitCap * 4u)) return rc;
    if (int rc = d_cand.ensure((uint64_t)max_cand * sizeof(BtleCand))) return rc;
    if (int rc = d_stage.ensure((uint64_t)max_cand * sizeof(snout_pkt))) return rc;
    if (int rc = d_accept.ensure((uint64_t)max_cand * 4u)) return rc;
    if (int rc = d_out_off.ensure((uint64_t)max_cand * 4u)) return rc;
    if (int rc = d_out.ensure((uint64_t)max_cand * sizeof(snout_pkt))) return rc;
    return 0;
}

// Narrowband front end: iq (device) -> planes + per-chunk hit lists.
int BtleCtx::launch_demod_corr(const float* d_iq, uint64_t n, hipStream_t st)
{
    SNOUT_HIP(hipEventRecord(ev_k0, st));
    hipLaunchKernelGGL(btle_demod_corr, dim3(cdiv(n_chunks, 4)), dim3(256), 0, st, d_iq, n, aa,
                       n_chunks, d_planes.as<uint64_t>(), d_chunk_cnt.as<uint32_t>(),
                       d_chunk_hits.as<uint32_t>(), (uint32_t)kChunkHitCap);
    SNOUT_HIP(hipEventRecord(ev_k1, st));
    SNOUT_HIP(hipGetLastError());
    return 0;
}

// Wideband front end already filled planes: correlate them.
int BtleCtx::launch_corr_planes(uint64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(btle_corr_planes, dim3(cdiv((uint64_t)n_chunks * n_slots, 4)), dim3(256), 0, st,
                       d_planes.as<uint64_t>(), plane_stride, n - 4u, aa, n_chunks, n_slots,
                       d_chunk_cnt.as<uint32_t>(), d_chunk_hits.as<uint32_t>(), (uint32_t)kChunkHitCap);
    SNOUT_HIP(hipGetLastError());
    return 0;
}

// Hit lists -> resolved, ordered packet records in host memory.
int BtleCtx::finish(uint64_t n, uint64_t first_index, hipStream_t st, snout_pkt* out, uint64_t cap,
                    uint64_t* n_out)
{
    const uint32_t lists = n_chunks * n_slots;
    uint32_t* tot = d_totals.as<uint32_t>();     // [0]=n_cand [1]=n_out [2]=sum of raw counts
    hipLaunchKernelGGL(scan_u32, dim3(1), dim3(1024), 0, st, d_chunk_cnt.as<uint32_t>(),
                       d_chunk_off.as<uint32_t>(), (const uint32_t*)nullptr, lists,
                       (uint32_t)kChunkHitCap, tot + 0);
    hipLaunchKernelGGL(scan_u32, dim3(1), dim3(1024), 0, st, d_chunk_cnt.as<uint32_t>(),
                       d_out_off.as<uint32_t>() /*scratch*/, (const uint32_t*)nullptr, lists,
                       0xFFFFFFFFu, tot + 2);
    const uint32_t g = std::min<uint32_t>(cdiv(lists, 256), 2048u);
    hipLaunchKernelGGL(btle_decode, dim3(g), dim3(256), 0, st, d_planes.as<uint64_t>(), plane_stride,
                       n - 4u, d_chunk_cnt.as<uint32_t>(), d_chunk_off.as<uint32_t>(),
                       d_chunk_hits.as<uint32_t>(), (uint32_t)kChunkHitCap, n_chunks, n_slots,
                       d_whiten.as<uint8_t>(), d_slot_channel.as<uint16_t>(), crc_init, first_index,
                       d_cand.as<BtleCand>(), d_stage.as<snout_pkt>(), max_cand);
    hipLaunchKernelGGL(btle_resolve, dim3(1024), dim3(256), 0, st, d_cand.as<BtleCand>(), tot + 0,
                       max_cand, d_accept.as<uint32_t>());
    hipLaunchKernelGGL(scan_u32, dim3(1), dim3(1024), 0, st, d_accept.as<uint32_t>(),
                       d_out_off.as<uint32_t>(), tot + 0, 0u, 1u, tot + 1);
    hipLaunchKernelGGL(emit_packets, dim3(1024), dim3(256), 0, st, d_stage.as<snout_pkt>(),
                       d_accept.as<uint32_t>(), d_out_off.as<uint32_t>(), tot + 0, max_cand,
                       d_out.as<snout_pkt>(), max_cand);
    SNOUT_HIP(hipGetLastError());
    SNOUT_HIP(hipMemcpyAsync(h_totals, tot, 16, hipMemcpyDeviceToHost, st));
    SNOUT_HIP(hipStreamSynchronize(st));
    last_n_cand = h_totals[0];
    const uint32_t raw = h_totals[2];
    uint64_t np = h_totals[1];
    *n_out = np;
    int rc = 0;
    if (raw > h_totals[0] || h_totals[0] > max_cand) {
        set_last_error("BTLE hit capacity exceeded: %u raw hits, per-chunk cap %d, max_cand %u",
                       raw, kChunkHitCap, max_cand);
        rc = SNOUT_EOVERFLOW;
    }
    if (np > cap) { np = cap; if (!rc) { set_last_error("output capacity %llu < %u packets",
                                         (unsigned long long)cap, h_totals[1]); rc = SNOUT_EOVERFLOW; } }
    if (np) {
        // staged through pinned memory so the copy runs at full PCIe rate whatever `out` is
        if (h_out_cap < np) {
            if (h_out) (void)hipHostFree(h_out);
            h_out_cap = np + np / 2 + 1024;
            SNOUT_HIP(hipHostMalloc((void**)&h_out, h_out_cap * sizeof(snout_pkt), hipHostMallocDefault));
        }
        SNOUT_HIP(hipMemcpyAsync(h_out, d_out.p, np * sizeof(snout_pkt), hipMemcpyDeviceToHost, st));
        SNOUT_HIP(hipEventRecord(ev_t1, st));
        SNOUT_HIP(hipStreamSynchronize(st));
        memcpy(out, h_out, np * sizeof(snout_pkt));
    } else {
        SNOUT_HIP(hipEventRecord(ev_t1, st));
        SNOUT_HIP(hipStreamSynchronize(st));
    }
    return rc;
}

}  // namespace snout
