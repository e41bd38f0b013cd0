// pfb.hip — 2x-oversampled polyphase FFT channelizer for gfx950 (CDNA4, wave64).
//
// The reference observes one channel at a time and hops sequentially (snout/core/radio.py:415,
// snout/util/btle.py:62); the channelizer is the north-star addition that feeds all 40 BTLE /
// 16 Zigbee channel demodulators from one wideband capture (SURVEY.md §2.1, §8d cfg #3/#4).
// Its arithmetic is specified in oracle/oracle_pfb.c; this kernel follows the same operation order
// (fmaf chains, butterfly order, twiddle products), so its f32 outputs are bit-identical.
//
//   u_m[r] = sum_p h[r + pM] x[mD + r + pM]           (M branches, P = 16 taps, D = M/2)
//   y_k[m] = (-1)^{km} FFT_M(u_m)[k]
//
// A workgroup walks a contiguous range of tiles of T output times; per tile:
//   1. stage the (T-1)D + MP input samples in LDS: the MP - D samples shared with the previous
//      tile are already there (moved to the front), the T D new ones arrive by 16-byte loads issued
//      one tile ahead, so the input is read from HBM once,
//   2. FIR: thread <-> (branch r, output parity e, group g).  Outputs m = e + 2i of one branch are
//      a sliding dot product over the branch stream z[q] = x[r + eD + qM]: 23 LDS reads feed 8
//      outputs x 16 taps, taps live in registers -> FMA-bound, not LDS-bound,
//   3. FFT in two passes that work in place on the FIR rows (M = M1 M2): thread <-> (m, n2) does the
//      M1-point DFT + twiddle, thread <-> (m, k1) the M2-point DFT,
//   4. either y_k[m] goes to HBM (16-byte stores, m fastest), or a fused epilogue consumes it from
//      LDS: BTLE hard bits into the bit planes (M = 40), 802.15.4 discriminator rows (M = 16).
// LDS rows are padded to M + 1 complex so column walks hit distinct banks.
//
// Roofline: at M = 40 the stage needs ~181 flop per input sample (FIR 128 + FFT), i.e. ~23 flop/B:
// just above the f32 vector ridge (157 TFLOP/s / 8 TB/s = 20 flop/B) -> bound by f32 VALU issue.
// The FIR is NOT GEMM-shaped (a per-branch sliding correlation: a Toeplitz operand with 2 useful
// columns), so MFMA does not apply; see DESIGN.md.
#include "common.h"
#include "zb_discrim.h"
#include "iq_fmt.h"
#include "pfb_tables.inc"

namespace snout {

#ifdef SNOUT_PFB_STAMPS
// Diagnostic build only (tools/pfb_stamps.py): cycles each phase of the tile loop takes, per wave, summed
// over the tiles of a workgroup: [block][wave][slot] (s_memtime deltas; slot 7: shader clock in kHz), and
// per block its start / end time (s_memrealtime, 100 MHz) and hardware ids.
__device__ unsigned long long g_pfb_stamps[1024 * 5 * 8];
__device__ unsigned long long g_pfb_times[1024 * 4];
#define STAMP(k)                                                                       \
    do {                                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                 \
        st_acc[k] += now_ - st_last; st_last = now_;                                   \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

struct cf { float re, im; };
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf cadd(cf a, cf b) { return cf{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return cf{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf cmul_tw(cf a, float c, float d)
{
    cf r;
    r.re = __builtin_fmaf(a.re, c, -(a.im * d));
    r.im = __builtin_fmaf(a.re, d, a.im * c);
    return r;
}

__device__ __forceinline__ void dft4(const cf b[4], cf X[4])
{
    const cf s0 = cadd(b[0], b[2]), s1 = csub(b[0], b[2]);
    const cf s2 = cadd(b[1], b[3]), s3 = csub(b[1], b[3]);
    X[0] = cadd(s0, s2);
    X[2] = csub(s0, s2);
    X[1] = cf{s1.re + s3.im, s1.im - s3.re};
    X[3] = cf{s1.re - s3.im, s1.im + s3.re};
}

__device__ __forceinline__ void dft8(const cf a[8], cf X[8])
{
    const float c = 0.70710678118654752440f;
    const cf e[4] = {a[0], a[2], a[4], a[6]}, o[4] = {a[1], a[3], a[5], a[7]};
    cf E[4], O[4], T[4];
    dft4(e, E);
    dft4(o, O);
    T[0] = O[0];
    T[1] = cf{(O[1].re + O[1].im) * c, (O[1].im - O[1].re) * c};
    T[2] = cf{O[2].im, -O[2].re};
    T[3] = cf{(O[3].im - O[3].re) * c, -((O[3].re + O[3].im) * c)};
#pragma unroll
    for (int k = 0; k < 4; k++) { X[k] = cadd(E[k], T[k]); X[k + 4] = csub(E[k], T[k]); }
}

// 5-point DFT by its real-factor symmetry, the operation order of oracle_pfb.c.
__device__ __forceinline__ void dft5(const cf b[5], cf X[5], const float C1, const float C2, const float S1, const float S2)
{
    const cf t1 = cadd(b[1], b[4]), t2 = cadd(b[2], b[3]), t3 = csub(b[1], b[4]), t4 = csub(b[2], b[3]);
    cf a1, a2, s1, s2;
    X[0] = cadd(cadd(b[0], t1), t2);
    a1.re = __builtin_fmaf(C2, t2.re, __builtin_fmaf(C1, t1.re, b[0].re));
    a1.im = __builtin_fmaf(C2, t2.im, __builtin_fmaf(C1, t1.im, b[0].im));
    a2.re = __builtin_fmaf(C1, t2.re, __builtin_fmaf(C2, t1.re, b[0].re));
    a2.im = __builtin_fmaf(C1, t2.im, __builtin_fmaf(C2, t1.im, b[0].im));
    s1.re = __builtin_fmaf(S2, t4.re, S1 * t3.re);
    s1.im = __builtin_fmaf(S2, t4.im, S1 * t3.im);
    s2.re = __builtin_fmaf(-S1, t4.re, S2 * t3.re);
    s2.im = __builtin_fmaf(-S1, t4.im, S2 * t3.im);
    X[1] = cf{a1.re + s1.im, a1.im - s1.re};
    X[4] = cf{a1.re - s1.im, a1.im + s1.re};
    X[2] = cf{a2.re + s2.im, a2.im - s2.re};
    X[3] = cf{a2.re - s2.im, a2.im + s2.re};
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// wait for the prefetch loads of the next tile and for this tile's output stores at every phase.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// waves per SIMD the register allocator must leave room for: 4 -> at most 128 VGPRs, three
// 320-thread workgroups per CU (M = 40) / four 256-thread ones (M = 16); the fused 802.15.4
// epilogue needs more registers and gets 3 (see PfbCtx::run)
// FIR forms (A/B switches, tools/pfb_variants.sh): by default the window is read with one ds_read_b64
// per sample and the MACs are v_pk_fma_f32 with the tap broadcast by op_sel; -DSNOUT_PFB_PLAIN_FIR
// leaves both to the compiler (ds_read2_b64 pairs, every tap kept twice).
#ifndef SNOUT_PFB_PLAIN_FIR
#define SNOUT_PFB_PKFIR 1
#define SNOUT_PFB_NOREAD2 1
#endif
#ifndef SNOUT_PFB_PRIO_FIR
#define SNOUT_PFB_PRIO_FIR 0
#endif
#ifndef SNOUT_PFB_PRIO_FFT
#define SNOUT_PFB_PRIO_FFT 1
#endif
#ifndef SNOUT_PFB_WPE
#define SNOUT_PFB_WPE 4
#endif
#ifndef SNOUT_PFB_WPE_ZB
#define SNOUT_PFB_WPE_ZB 3
#endif

template <int M> struct PfbGeom;
#ifndef SNOUT_PFB_T64
// 128 output times per tile: half the barriers per output of a 64-time tile, and one FIR window of 31
// samples serves 16 outputs where two windows of 23 served 8 each (3.47 -> 3.27 ms per 8e8 samples)
#define SNOUT_PFB_T128 1
template <> struct PfbGeom<40> { static constexpr int T = 128, M1 = 8, M2 = 5, NT = 320; };
extern __shared__ float4 pfb_dyn_lds[];
#else
template <> struct PfbGeom<40> { static constexpr int T = 64,  M1 = 8, M2 = 5, NT = 320; };
#endif
#if defined(SNOUT_PFB_NO_FUSE3B) && defined(SNOUT_PFB_T128)
#error "the separate slicer pass (SNOUT_PFB_NO_FUSE3B) is written for 64-time tiles: add -DSNOUT_PFB_T64"
#endif
template <> struct PfbGeom<16> { static constexpr int T = 128, M1 = 4, M2 = 4, NT = 256; };

// One workgroup walks a contiguous range of tiles of T output times.  Consecutive tiles share
// (MP - D) input samples: they stay in LDS (moved to the front through registers), only T D new
// samples are fetched per tile, one tile ahead, into registers.
//
// FUSED (802.15.4, M = 16): the tile's outputs never leave LDS either: thread (channel, 8 output
// times) applies the FM discriminator d[m] = fast_atan2f(y[m] conj y[m-1]) (zb_discrim.h) and stores
// d and the IIR sub-block sums S_j straight into the Zigbee context (4 B instead of 8 B written
// per channel sample, and no zb_discrim pass: 8 B read + 4 B written saved again).  y[m0 - 1]
// comes from the previous tile through LDS, so the workgroup first computes the tile before its
// range without emitting it.
//
// FUSED (BTLE, M = 40): instead of writing 16 B of channel IQ per input sample the tile keeps its
// outputs in LDS and emits the BTLE hard bits  bit[m] = (I[m] Q[m+4]) > (I[m+4] Q[m])  of all 40
// channels straight into the bit planes the correlator reads (0.25 B per input sample).  Thread
// (channel, phase) decides 15 symbols of the tile at once and the 16th when the next tile's first
// outputs exist (it carries its last sample and the pending 15 bits in registers); the workgroup
// therefore also computes the tile after its range, without emitting that tile's own bits.
// FMT: input sample format (iq_fmt.h); integer samples are converted as they are fetched.
template <int M, bool FUSED, int FMT>
__device__ __forceinline__ void pfb_body(
    const PfbSegs& segs, uint64_t n, uint64_t n_out, uint32_t n_tiles, uint32_t tiles_per_wg,
    const float* __restrict__ proto, const float* __restrict__ twM, const float* __restrict__ tw5g,
    float2* __restrict__ y, uint64_t y_stride, uint16_t* __restrict__ planes16,
    uint64_t plane_stride, PfbZbOut zb)
{
    // which segment of the launch this workgroup works on, and which of that segment's workgroups it is
    const uint32_t seg = blockIdx.x / segs.wgs_per_seg, bid = blockIdx.x - seg * segs.wgs_per_seg;
    const void* __restrict__ x = segs.x[seg];
    if (planes16) planes16 += (uint64_t)seg * segs.planes_seg;
    zb.d += (uint64_t)seg * segs.d_seg;
    zb.S += (uint64_t)seg * segs.S_seg;
    using G = PfbGeom<M>;
    constexpr int T = G::T, M1 = G::M1, M2 = G::M2, D = M / 2, P = 16, NT = G::NT;
    constexpr int SPAN = (T - 1) * D + M * P;      // input samples one tile needs
    constexpr int NEW = T * D;                     // of which new per tile
    constexpr int ROW = M + 1;                     // padded LDS row, complex
    static_assert(SPAN % 2 == 0 && NEW % 2 == 0, "16-byte staging needs even sample counts");
    constexpr int SPAN4 = SPAN / 2, NEW4 = NEW / 2, OV4 = SPAN4 - NEW4;   // in pairs of samples
    static_assert(NEW4 % NT == 0 && OV4 <= NT, "staging shape");
    constexpr int NPRE = NEW4 / NT;                // new pairs each thread stages per tile
    __shared__ float2 xs[SPAN];                    // input span of the current tile
    constexpr bool ZB = FUSED && M == 16;          // fused 802.15.4 discriminator epilogue
#ifdef SNOUT_PFB_NO_FUSE3B
    constexpr bool FUSE3B = false;
#else
    constexpr bool FUSE3B = FUSED && M == 40;      // BTLE: pass 3b decides the hard bits in registers
#endif
#ifdef SNOUT_PFB_T128
    // M = 40: in dynamic LDS (42 KB) -- static LDS of this size makes the compiler raise next_free_vgpr to an
    // occupancy it derives from it, and the second workgroup no longer fits the CU (profiles/r2_pfb_experiments.md)
    __shared__ float2 us_static[M == 40 ? 1 : T * ROW];
    float2* const us = M == 40 ? reinterpret_cast<float2*>(pfb_dyn_lds) : us_static;
#else
    __shared__ float2 us[T * ROW];                 // FIR outputs u_m[r]; both FFT passes work in place
#endif
    constexpr int DLROW = T + 1;                   // padded row of the tile's d values (S_j reads)
    static_assert(!ZB || (M * DLROW <= 2 * SPAN && T % 128 == 0 && NT == 256), "d tile reuses the input span");
    __shared__ float atan_s[ZB ? 257 : 1];
    __shared__ double wts_s[ZB ? 64 : 1];
    __shared__ float2 prevy[ZB ? M : 1];           // y[m0 - 1] of every channel

    const int t = threadIdx.x;
    // 5-point DFT factors (kTw5): uniform loads, scalar registers
    const float c5_1 = tw5g[2], c5_2 = tw5g[4], s5_1 = -tw5g[3], s5_2 = -tw5g[5];
    if constexpr (ZB) {
        for (int i = t; i < 257; i += NT) atan_s[i] = zb.atan_tab[i];
        if (t < 64) wts_s[t] = zb.iir_w[t];
        if (t < M) prevy[t] = make_float2(0.0f, 0.0f);         // x[-1] = 0 for the very first tile
    }
    // FIR role of this thread: (branch r, output parity e, group grp); taps live in registers
    const int r = t % M, e = (t / M) & 1, grp = t / (2 * M);
    float h[P];
#pragma unroll
    for (int p = 0; p < P; p++) h[p] = proto[r + p * M];
#ifdef SNOUT_PFB_PKFIR
    v2f hp[P / 2];
#pragma unroll
    for (int p = 0; p < P / 2; p++) hp[p] = v2f{h[2 * p], h[2 * p + 1]};
#endif

    static_assert((T * M2) % NT == 0 && T % 64 == 0, "pass 3a: whole rounds, wave-uniform n2");
    // W_M^{n2 k1} by row n2: a wave of pass 3a reads its row as 2 M1 / 4 wave-uniform 16-byte LDS reads per
    // tile (broadcast reads: conflict-free, and no register is held across the other phases)
    __shared__ float4 tw3_s[M2 * (2 * M1 / 4)];
    for (int i = t; i < M2 * 2 * M1; i += NT) {
        const int n2i = i / (2 * M1), k1 = (i % (2 * M1)) >> 1, j = (n2i * k1) % M;    // n2 k1 < M for both geometries
        reinterpret_cast<float*>(tw3_s)[i] = twM[2 * j + (i & 1)];
    }

    const uint32_t t_begin = bid * tiles_per_wg;
    uint32_t t_end = t_begin + tiles_per_wg;
    if (t_end > n_tiles) t_end = n_tiles;
    if (t_begin >= t_end) return;
    // fused BTLE: one more tile (if it exists) supplies the samples the last symbols are compared with;
    // fused 802.15.4: the tile before the range supplies y[m0 - 1]
    const uint32_t t_last = (FUSED && M == 40 && t_end < n_tiles) ? t_end + 1u : t_end;
    const uint32_t t_first = (ZB && t_begin > 0u) ? t_begin - 1u : t_begin;

    // Samples are fetched one tile ahead and stay RAW in registers until they are staged: converting
    // an integer format right behind the load would make the FIR wait for the load.
    using Raw = typename IqRaw<FMT>::pair;
    auto load_pair = [&](uint64_t g) -> Raw {                  // samples g, g+1 (g even), zero past n
        if (g + 1 < n) return iq_pair_raw<FMT>(x, g);
        Raw v = Raw{};
        if (g < n) v = iq_single_raw<FMT>(x, g);
        return v;
    };
    // first tile: the whole span; later tiles: the overlap comes from LDS (keep), the rest from pre
    float4 keep = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    Raw pre[NPRE];
    {
        const uint64_t in0 = (uint64_t)t_first * NEW;            // even -> 16-byte aligned
        if (t < OV4) keep = iq_pair_cvt<FMT>(load_pair(in0 + 2ull * (uint64_t)t));
#pragma unroll
        for (int k = 0; k < NPRE; k++) pre[k] = load_pair(in0 + 2ull * (uint64_t)(OV4 + t + k * NT));
    }
    // fused: state of thread (channel k, phase j) across tiles
    float2 carry = make_float2(0.0f, 0.0f);        // y_k[m0 - 4 + j] of the previous tile
    uint32_t pend = 0;                             // its 15 decided bits
    bool have_prev = false;
    cf carry5[FUSE3B ? M2 : 1];                    // FUSE3B: yB of the previous tile (the b = 7 threads use it)
#pragma unroll
    for (int i = 0; i < (FUSE3B ? M2 : 1); i++) carry5[i] = cf{0.0f, 0.0f};

#ifdef SNOUT_PFB_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_last, st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef SNOUT_PFB_STAGGER
    // Workgroups that share a CU start together and would run their LDS-heavy and VALU-heavy phases
    // in step; delay every other round of the grid by a fraction of a tile.
    for (uint32_t i = 0; i < (blockIdx.x / 256u) % 3u; i++) __builtin_amdgcn_s_sleep(SNOUT_PFB_STAGGER);
#endif
    // The phases of one tile, as closures over the kernel's state (inlined); `us` is the FIR-output /
    // FFT buffer they work on.
    auto do_stage = [&]() {
        // ---- 1. stage: overlap to the front, new samples behind it
        if (t < OV4) reinterpret_cast<float4*>(xs)[t] = keep;
#pragma unroll
        for (int k = 0; k < NPRE; k++) reinterpret_cast<float4*>(xs)[OV4 + t + k * NT] = iq_pair_cvt<FMT>(pre[k]);
    };
    auto do_prefetch = [&](uint32_t nxt) {       // request tile nxt's new samples (raw, into registers)
#if !defined(SNOUT_PFB_LATE_PREFETCH) && !defined(SNOUT_ABL_NOGLOBAL)
        if (nxt < t_last) {
            const uint64_t in1 = (uint64_t)nxt * NEW + 2ull * OV4;   // first new sample of the next tile
            if (in1 + NEW <= n) {                 // uniform: all NEW samples exist, no per-lane range tests
#pragma unroll
                for (int k = 0; k < NPRE; k++) pre[k] = iq_pair_raw<FMT>(x, in1 + 2ull * (uint64_t)(t + k * NT));
            } else {
#pragma unroll
                for (int k = 0; k < NPRE; k++) pre[k] = load_pair(in1 + 2ull * (uint64_t)(t + k * NT));
            }
        }
#endif
    };
    auto do_fir = [&]() {
        // ---- 2. FIR: outputs m = e + 2 (8 grp + i) of branch r: a sliding dot product
        constexpr int OUT = T / (2 * (NT / (2 * M)));         // outputs per (branch, parity, group): 8, or 16
        if constexpr (OUT == 16) {
            // m = e + 2 (16 grp + i), i = 0..15: one window of 31 samples; the first four outputs need samples
            // 0..18, every further four outputs four more, read while the four before them are computed
            const int base = r + e * D + (16 * grp) * M;
            const uint32_t a0 = (uint32_t)(uintptr_t)&xs[base];
            v2f wv[16 + P - 1];
#pragma unroll
            for (int q = 0; q < 19; q++)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(wv[q]) : "v"(a0), "n"(q * M * 8) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]), "+v"(wv[4]), "+v"(wv[5]), "+v"(wv[6]),
                           "+v"(wv[7]), "+v"(wv[8]), "+v"(wv[9]), "+v"(wv[10]), "+v"(wv[11]), "+v"(wv[12]),
                           "+v"(wv[13]), "+v"(wv[14])
                         :: "memory");
            asm volatile("" : "+v"(wv[15]), "+v"(wv[16]), "+v"(wv[17]), "+v"(wv[18]) :: "memory");
#pragma unroll
            for (int i0 = 0; i0 < 16; i0 += 4) {
                if (i0 < 12) {
#pragma unroll
                    for (int q = i0 + 19; q < i0 + 23; q++)
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(wv[q]) : "v"(a0), "n"(q * M * 8) : "memory");
                }
                v2f acc[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(acc[j]) : "v"(hp[0]), "v"(wv[i0 + j]));
#pragma unroll
                for (int p = 1; p < P; p++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (p & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[j]) : "v"(hp[p >> 1]), "v"(wv[i0 + j + p]));
                        else       asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j]) : "v"(hp[p >> 1]), "v"(wv[i0 + j + p]));
                    }
                }
                if (i0 < 12)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wv[i0 + 19]), "+v"(wv[i0 + 20]), "+v"(wv[i0 + 21]), "+v"(wv[i0 + 22]) :: "memory");
#pragma unroll
                for (int j = 0; j < 4; j++) us[(e + 2 * (16 * grp + i0 + j)) * ROW + r] = make_float2(acc[j].x, acc[j].y);
            }
        } else {
            const int base = r + e * D + (8 * grp) * M;       // tile-relative index of z[8 grp]
            float2 w[8 + P - 1];
#ifdef SNOUT_PFB_NOREAD2
            // one ds_read_b64 per window sample: the compiler pairs them into ds_read2_b64, which moves
            // the same bytes at half the LDS rate (MI355X_MICROARCH.md, LDS table)
            {
                const uint32_t a0 = (uint32_t)(uintptr_t)&xs[base];
                v2f wv[8 + P - 1];
#pragma unroll
                for (int q = 0; q < 8 + P - 1; q++)
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(wv[q]) : "v"(a0), "n"(q * M * 8) : "memory");
                // the wait names every loaded register as an operand, so that no use can be scheduled ahead of it
                static_assert(8 + P - 1 == 23, "operand list below");
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]), "+v"(wv[4]), "+v"(wv[5]), "+v"(wv[6]),
                               "+v"(wv[7]), "+v"(wv[8]), "+v"(wv[9]), "+v"(wv[10]), "+v"(wv[11]), "+v"(wv[12]),
                               "+v"(wv[13]), "+v"(wv[14])
                             :: "memory");
                asm volatile("" : "+v"(wv[15]), "+v"(wv[16]), "+v"(wv[17]), "+v"(wv[18]), "+v"(wv[19]), "+v"(wv[20]),
                                  "+v"(wv[21]), "+v"(wv[22]) :: "memory");
#pragma unroll
                for (int q = 0; q < 8 + P - 1; q++) w[q] = make_float2(wv[q].x, wv[q].y);
            }
#else
#pragma unroll
            for (int q = 0; q < 8 + P - 1; q++) w[q] = xs[base + q * M];
#endif
#if defined(SNOUT_ABL_NOFIRMATH)
#pragma unroll
            for (int i = 0; i < 8; i++) us[(e + 2 * (8 * grp + i)) * ROW + r] = make_float2(w[i].x + w[i + 15].x, w[i].y + w[i + 8].y);
#elif defined(SNOUT_PFB_PKFIR)
            // (re, im) of one output advance together in v_pk_fma_f32; the tap is broadcast to both halves
            // by op_sel from a register PAIR holding two consecutive taps, so the 16 taps take 16
            // registers (the compiler's own packing keeps every tap twice: 32).  Four outputs advance
            // side by side, tap by tap: consecutive instructions never depend on each other (the
            // two-chain order the compiler picks needs an s_nop between most of them).
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 4) {
                v2f acc[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const v2f x0 = {w[i0 + j].x, w[i0 + j].y};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(acc[j]) : "v"(hp[0]), "v"(x0));
                }
#pragma unroll
                for (int p = 1; p < P; p++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const v2f xv = {w[i0 + j + p].x, w[i0 + j + p].y};
                        if (p & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[j]) : "v"(hp[p >> 1]), "v"(xv));
                        else       asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j]) : "v"(hp[p >> 1]), "v"(xv));
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) us[(e + 2 * (8 * grp + i0 + j)) * ROW + r] = make_float2(acc[j].x, acc[j].y);
            }
#else
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float ar = 0.0f, ai = 0.0f;
#pragma unroll
                for (int p = 0; p < P; p++) {
                    ar = __builtin_fmaf(h[p], w[i + p].x, ar);
                    ai = __builtin_fmaf(h[p], w[i + p].y, ai);
                }
                us[(e + 2 * (8 * grp + i)) * ROW + r] = make_float2(ar, ai);
            }
#endif
        }
        if (t < OV4) keep = reinterpret_cast<const float4*>(xs)[NEW4 + t];   // next tile's overlap
    };
    auto do_3a = [&]() {
        // ---- 3a. M1-point DFTs over n1 for every (m, n2), then twiddle W_M^{n2 k1}; in place:
        //      slot M2 k1 + n2 of row m receives B[n2][k1]
#ifndef SNOUT_ABL_NO3A
#pragma unroll
        for (int rnd = 0; rnd < T * M2 / NT; rnd++) {
            // T is a multiple of 64, so n2 is the same for a whole wave: its twiddles W_M^{n2 k1} are
            // wave-uniform: one row of tw3_s
            const int it = t + rnd * NT;
            const int m = it % T, n2 = __builtin_amdgcn_readfirstlane(it / T);
            cf a[M1], A[M1];
#pragma unroll
            for (int n1 = 0; n1 < M1; n1++) {
                const float2 v = us[m * ROW + M2 * n1 + n2];
                a[n1] = cf{v.x, v.y};
            }
            if constexpr (M1 == 8) dft8(a, A); else dft4(a, A);
            float tw3a[2 * M1];
#pragma unroll
            for (int q = 0; q < 2 * M1 / 4; q++) {
                const float4 v = tw3_s[n2 * (2 * M1 / 4) + q];
                tw3a[4 * q] = v.x; tw3a[4 * q + 1] = v.y; tw3a[4 * q + 2] = v.z; tw3a[4 * q + 3] = v.w;
            }
            us[m * ROW + n2] = make_float2(A[0].re, A[0].im);           // k1 = 0: W^0
            if (n2 == 0) {                                              // uniform: W^0 throughout
#pragma unroll
                for (int k1 = 1; k1 < M1; k1++) us[m * ROW + M2 * k1 + n2] = make_float2(A[k1].re, A[k1].im);
            } else {
#pragma unroll
                for (int k1 = 1; k1 < M1; k1++) {
                    const cf v = cmul_tw(A[k1], tw3a[2 * k1], tw3a[2 * k1 + 1]);
                    us[m * ROW + M2 * k1 + n2] = make_float2(v.re, v.im);
                }
            }
        }
#endif
    };
    auto do_3b = [&](uint32_t tile) {
        const uint64_t m0 = (uint64_t)tile * T;
        // ---- 3b. M2-point DFTs over n2 for every (m pair, k1); y_k[m] = (-1)^{km} X[k].
        //      Fused: in place, channel k = k1 + M1 k2 ends up in slot M2 k1 + k2 of row m.
        //      Otherwise a thread owns two consecutive output times so each global store is 16 B.
        if constexpr (FUSE3B) {
            // ---- 3b + 4 fused (BTLE).  Thread <-> (phase a, symbol pair b, k1): the 5-point DFTs of the
            //      output times mA = a + 8 b and mB = mA + 4, i.e. symbols 2b and 2b+1 of sampling phase a,
            //      for the channels k = k1 + 8 k2.  bit[m] = (I[m] Q[m+4]) > (I[m+4] Q[m]) needs y[m] and
            //      y[m+4] only, so symbol 2b is decided from this thread's own two results and symbol
            //      2b+1 with yA of the thread 4 lanes up (ds_bpermute; the 32 threads of a k1 share half
            //      a wave) -- the outputs never go back to LDS.  The factor (-1)^{km} is the same for m and
            //      m+4 and cancels in both products exactly, so it is not applied.  The last symbol of a
            //      tile (b = 7) has its partner in the next tile: the thread carries its yB in registers
            //      and its bit is produced one tile later, in the same lane position.  The bits of the 64
            //      lanes come out of v_cmp as a wave mask; lanes (k2, k1 half g, a) pick the 2 x 8 bits of
            //      their channel and phase out of the masks and interleave them into the 16-bit quarter
            //      of the plane word.
            // (T = 128: the tile is two such halves of 64 output times, one after the other; of the tile
            //  behind the range only the first half is needed, to complete the range's last symbols)
#pragma unroll
            for (int hv = 0; hv < T / 64; hv++) {
            if (hv == 1 && tile >= t_end) break;
            const uint64_t m0v = m0 + 64u * (uint32_t)hv;
            const float2* usv = us + 64 * hv * ROW;
            const bool last_half = tile + 1u == n_tiles && hv == T / 64 - 1;
#ifdef SNOUT_ABL_NO3B
            if (n == 0x123456789ull) {             // timing experiment: never true
#else
            if (t < 32 * M1) {
#endif
                const int mp = t & 31, k1 = t >> 5, a = mp & 3, b = mp >> 2;
                const int mA = a + 8 * b;
                cf YA[M2], YB[M2];
                {
                    cf bb[M2];
#pragma unroll
                    for (int n2 = 0; n2 < M2; n2++) { const float2 v = usv[mA * ROW + M2 * k1 + n2]; bb[n2] = cf{v.x, v.y}; }
                    dft5(bb, YA, c5_1, c5_2, s5_1, s5_2);
#pragma unroll
                    for (int n2 = 0; n2 < M2; n2++) { const float2 v = usv[(mA + 4) * ROW + M2 * k1 + n2]; bb[n2] = cf{v.x, v.y}; }
                    dft5(bb, YB, c5_1, c5_2, s5_1, s5_2);
                }
                const int src = ((t & 32) | ((t + 4) & 31)) << 2;         // lane of (a, b + 1 mod 8, k1)
                // lanes 0..39 of the wave: (k2s, g, j) -> channel k = (2 wave + g) + 8 k2s, phase j
                const int L = t & 63, k2s = L >> 3, g = (L >> 2) & 1, j = L & 3;
                uint64_t sa = 0, sb = 0;
#pragma unroll
                for (int k2 = 0; k2 < M2; k2++) {
                    const float pre_ = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(YA[k2].re)));
                    const float pim_ = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(YA[k2].im)));
                    const float ore = b == 7 ? carry5[k2].re : YB[k2].re;  // b = 7: yB of the previous tile
                    const float oim = b == 7 ? carry5[k2].im : YB[k2].im;
                    const uint64_t mA_ = __builtin_amdgcn_ballot_w64((YA[k2].re * YB[k2].im) > (YB[k2].re * YA[k2].im));
                    const uint64_t mB_ = __builtin_amdgcn_ballot_w64((ore * pim_) > (pre_ * oim));
                    if (k2s == k2) { sa = mA_; sb = mB_; }               // the lanes that assemble this k2
                    carry5[k2] = YB[k2];
                }
                const uint32_t xa = (uint32_t)(sa >> (32 * g + j)) & 0x11111111u;     // bit 4b: symbol 2b
                const uint32_t xb = (uint32_t)(sb >> (32 * g + j)) & 0x11111111u;     // bit 4b: symbol 2b+1 (b = 7: previous tile's 15)
                uint32_t c = xa | (xb << 1);
                c = (c | (c >> 2)) & 0x0F0F0F0Fu;
                c = (c | (c >> 4)) & 0x00FF00FFu;
                c = (c | (c >> 8)) & 0x0000FFFFu;                                     // bit l: symbol l
                if (L < 8 * M2) {
                    const uint64_t nbits = n_out >= 4 ? n_out - 4 : 0;              // bits exist for m < n_out-4
                    const int k = 2 * (t >> 6) + g + M1 * k2s;
                    if (have_prev) {
                        const uint64_t mprev = m0v - 4u + (uint64_t)j;                // sample of the carried symbol
                        const uint32_t b15 = (mprev < nbits) ? (c >> 15) : 0u;
                        const uint64_t mq = m0v - 64u;                               // first sample of the previous 64 output times
                        planes16[((uint64_t)k * plane_stride + (mq >> 8) * 4u + (uint32_t)j) * 4u +
                                 (uint32_t)((mq & 255u) >> 6)] = (uint16_t)(pend | (b15 << 15));
                    }
                    const uint32_t left = nbits > m0v ? (uint32_t)(nbits - m0v < 64u ? nbits - m0v : 64u) : 0u;   // uniform
                    const uint32_t cnt = left > (uint32_t)j ? (left - (uint32_t)j + 3u) >> 2 : 0u;
                    pend = c & ((1u << (cnt < 15u ? cnt : 15u)) - 1u);
                    if (last_half)                                                   // no later tile: symbol 15 has no partner
                        planes16[((uint64_t)k * plane_stride + (m0v >> 8) * 4u + (uint32_t)j) * 4u +
                                 (uint32_t)((m0v & 255u) >> 6)] = (uint16_t)pend;
                }
                have_prev = true;
            }
            }   // halves
        } else
        for (int it = t; it < (T / 2) * M1; it += NT) {
            const int mp = it % (T / 2), k1 = it / (T / 2);
            cf Y0[M2], Y1[M2];
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {
                const int m = 2 * mp + h2;
                cf b[M2];
#pragma unroll
                for (int n2 = 0; n2 < M2; n2++) {
                    const float2 v = us[m * ROW + M2 * k1 + n2];
                    b[n2] = cf{v.x, v.y};
                }
                if (h2 == 0) { if constexpr (M2 == 5) dft5(b, Y0, c5_1, c5_2, s5_1, s5_2); else dft4(b, Y0); }
                else         { if constexpr (M2 == 5) dft5(b, Y1, c5_1, c5_2, s5_1, s5_2); else dft4(b, Y1); }
            }
            const uint64_t mg = m0 + 2ull * (uint64_t)mp;        // even: only the odd time flips
#pragma unroll
            for (int k2 = 0; k2 < M2; k2++) {
                const int k = k1 + M1 * k2;
                cf v1 = Y1[k2];
                if (k & 1) { v1.re = -v1.re; v1.im = -v1.im; }
                if constexpr (FUSED) {
                    us[(2 * mp) * ROW + M2 * k1 + k2] = make_float2(Y0[k2].re, Y0[k2].im);
                    us[(2 * mp + 1) * ROW + M2 * k1 + k2] = make_float2(v1.re, v1.im);
                } else {
                    float2* dst = &y[(uint64_t)k * y_stride + mg];
                    if (mg + 1 < n_out) *reinterpret_cast<float4*>(dst) = make_float4(Y0[k2].re, Y0[k2].im, v1.re, v1.im);
                    else if (mg < n_out) *dst = make_float2(Y0[k2].re, Y0[k2].im);
                }
            }
        }

        if constexpr (ZB) {
            lds_barrier();
            // ---- 4z. discriminator.  Thread <-> (channel k, output times TPT seg .. TPT seg + TPT - 1); channel
            //      k = k1 + M1 k2 sits in slot M2 k1 + k2 of every row.
            constexpr int TPT = T / (NT / M);                         // output times per thread: 8 (T = 128) or 16
            float* dl = reinterpret_cast<float*>(xs);                 // [k][DLROW]: the span is dead here
            const int k = t & (M - 1), seg = t / M;
            const float2* col = &us[M2 * (k % M1) + (k / M1)];
            float2 p = seg ? col[(TPT * seg - 1) * ROW] : prevy[k];
            float dv[TPT];
            const uint32_t left = n_out > m0 ? (uint32_t)(n_out - m0 < (uint64_t)T ? n_out - m0 : (uint64_t)T) : 0u;   // outputs of this tile that exist (uniform)
            if (left >= (uint32_t)T) {                                // every tile but the last: no range test per sample
#pragma unroll
                for (int j = 0; j < TPT; j++) {
                    const float2 a = col[(TPT * seg + j) * ROW];
                    dv[j] = zb_discriminate(a, p, atan_s);
                    dl[k * DLROW + TPT * seg + j] = dv[j];
                    p = a;
                }
            } else {
#pragma unroll
                for (int j = 0; j < TPT; j++) {
                    const float2 a = col[(TPT * seg + j) * ROW];
                    const float v = zb_discriminate(a, p, atan_s);
                    dv[j] = ((uint32_t)(TPT * seg + j) < left) ? v : 0.0f;
                    dl[k * DLROW + TPT * seg + j] = dv[j];
                    p = a;
                }
            }
            const bool emit = tile >= t_begin;                        // the tile before the range only primes prevy
            if (emit) {
                float* dst = zb.d + (uint64_t)k * zb.d_stride + m0 + (uint64_t)(TPT * seg);
#pragma unroll
                for (int q = 0; q < TPT / 4; q++)
                    reinterpret_cast<float4*>(dst)[q] = make_float4(dv[4 * q], dv[4 * q + 1], dv[4 * q + 2], dv[4 * q + 3]);
            }
            lds_barrier();
            if (seg == NT / M - 1) prevy[k] = p;                      // y[m0 + T - 1] for the next tile
            // S_j of the tile's T / 64 x M sub-blocks (oracle order): four threads per sub-block sum 16
            // terms each in sequence, S = (P0 + P1) + (P2 + P3)
            if (emit && t < (T / 64) * 4 * M) {
                const int part = t & 3, kk = (t >> 2) & (M - 1), sb = t >> 6;
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < 16; i++)
                    acc = acc + wts_s[63 - (16 * part + i)] * (double)dl[kk * DLROW + 64 * sb + 16 * part + i];
                acc = acc + __shfl_down(acc, 1);
                acc = acc + __shfl_down(acc, 2);
                const uint64_t j = (m0 >> 6) + (uint64_t)sb;
                if (part == 0 && j < zb.nsb) zb.S[(uint64_t)kk * zb.nsb + j] = acc;
            }
            // the staging of the next tile overwrites dl (= xs) only after its own barrier ... no:
            lds_barrier();                                            // ... it writes xs first: wait for the S_j reads
        }

        if constexpr (FUSED && M == 40 && !FUSE3B) {
            lds_barrier();
            // ---- 4. hard bits.  Thread <-> (channel k, phase j), samples m = 4 s + j: symbol 15 of
            //      the previous tile (its sample is carried in a register) completes that tile's
            //      quarter of the 64-symbol plane word, which is stored now; symbols 0..14 of this
            //      tile wait for the next one.
            //      planes: [slot k][word g][phase j] u64, bit l = sample 256 g + 4 l + j.
            if (t < M * 4) {
                const uint64_t nbits = n_out >= 4 ? n_out - 4 : 0;      // bits exist for m < n_out-4
                const int k = t >> 2, j = t & 3;
                const float2* col = &us[M2 * (k % M1) + (k / M1)];       // channel k of row 0
                float2 a = col[j * ROW];
                if (have_prev) {
                    const uint64_t mprev = m0 - 4u + (uint64_t)j;        // sample of the carried value
                    const bool bit = ((carry.x * a.y) > (a.x * carry.y)) && (mprev < nbits);
                    const uint64_t mq = m0 - (uint64_t)T;                // first sample of the previous tile
                    planes16[((uint64_t)k * plane_stride + (mq >> 8) * 4u + (uint32_t)j) * 4u +
                             (uint32_t)((mq & 255u) >> 6)] = (uint16_t)(pend | ((bit ? 1u : 0u) << 15));
                }
                uint32_t bits = 0;
#pragma unroll
                for (int sy = 0; sy < 15; sy++) {
                    const float2 b4 = col[(4 * sy + 4 + j) * ROW];
                    bits |= ((a.x * b4.y) > (b4.x * a.y) ? 1u : 0u) << sy;
                    a = b4;
                }
                // symbols whose bit exists (m0 + 4 sy + j < nbits): a prefix of the 15, as one mask
                const uint32_t left = nbits > m0 ? (uint32_t)(nbits - m0 < 64u ? nbits - m0 : 64u) : 0u;   // uniform
                const uint32_t cnt = left > (uint32_t)j ? (left - (uint32_t)j + 3u) >> 2 : 0u;
                pend = bits & ((1u << (cnt < 15u ? cnt : 15u)) - 1u);
                carry = a;                                               // y_k[m0 + 60 + j]
                have_prev = true;
                if (tile + 1u == n_tiles)                                // no later tile: symbol 15 has no partner
                    planes16[((uint64_t)k * plane_stride + (m0 >> 8) * 4u + (uint32_t)j) * 4u +
                             (uint32_t)((m0 & 255u) >> 6)] = (uint16_t)pend;
            }
        }
        // the next tile's FIR writes us only after the staging barrier, i.e. after every read above
    };

    for (uint32_t tile = t_first; tile < t_last; tile++) {
        do_stage();
        STAMP(0);
        lds_barrier();
        STAMP(1);
        // The FIR is the throughput phase (128 independent packed FMAs per thread); the FFT passes and
        // the slicer are dependency chains with LDS round trips in them, and the workgroup waits for
        // their slowest wave at every barrier.  The SIMD arbitrates by priority, then age: with equal
        // priority another workgroup's FIR waves take the issue slots the FFT waves need between their
        // stalls.  FIR at priority 0, everything else at 1: -10 % on the whole kernel (tools/pfb_ab.py).
        __builtin_amdgcn_s_setprio(SNOUT_PFB_PRIO_FIR);
        do_prefetch(tile + 1u);
        do_fir();
        STAMP(2);
        __builtin_amdgcn_s_setprio(SNOUT_PFB_PRIO_FFT);
        lds_barrier();
        STAMP(3);
        do_3a();
        STAMP(4);
        lds_barrier();
        STAMP(5);
        do_3b(tile);
        // the next tile's FIR writes us only after the staging barrier, i.e. after every read above
        STAMP(6);
    }
#ifdef SNOUT_PFB_STAMPS
    if ((t & 63) == 0 && blockIdx.x < 1024) {
        st_acc[7] = (__builtin_amdgcn_s_memtime() - st_t0) * 100000ull / (__builtin_amdgcn_s_memrealtime() - st_r0 + 1ull);
        for (int k = 0; k < 8; k++) g_pfb_stamps[(blockIdx.x * 5 + (t >> 6)) * 8 + k] = st_acc[k];
        if (t == 0) {
            g_pfb_times[blockIdx.x * 4 + 0] = st_r0;
            g_pfb_times[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            g_pfb_times[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
            g_pfb_times[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        }
    }
#endif
}

#define SNOUT_PFB_ARGS                                                                                   \
    PfbSegs segs, uint64_t n, uint64_t n_out, uint32_t n_tiles, uint32_t tiles_per_wg,                    \
    const float* __restrict__ proto, const float* __restrict__ twM, const float* __restrict__ tw5g,        \
    float2* __restrict__ y, uint64_t y_stride, uint16_t* __restrict__ planes16, uint64_t plane_stride, PfbZbOut zb
// Register budgets: M = 40 needs 4 wave slots per SIMD (<= 128 VGPRs) or a second 5-wave workgroup does not fit
// a CU next to the first; the fused 802.15.4 epilogue is built for 3 (148 VGPRs, nothing spilled).
// NB the backend also RAISES next_free_vgpr to match an occupancy it derives from static LDS: a variant of this
// kernel with > 54.6 KB of static LDS (= fewer than three workgroups per CU by LDS) got next_free_vgpr 129 for
// 114 used registers, hence 3 wave slots per SIMD and ONE resident workgroup (tools/occprobe.hip,
// profiles/r2_pfb_experiments.md).
template <int M, bool FUSED, int FMT>
__global__ __launch_bounds__(PfbGeom<M>::NT) __attribute__((amdgpu_waves_per_eu((FUSED && M == 16) ? SNOUT_PFB_WPE_ZB : SNOUT_PFB_WPE)))
void pfb_channelize(SNOUT_PFB_ARGS)
{
    pfb_body<M, FUSED, FMT>(segs, n, n_out, n_tiles, tiles_per_wg, proto, twM, tw5g, y, y_stride, planes16, plane_stride, zb);
}

// =============================================================================================
// Host side: launcher only (the channelizer context lives in pfb_ctx.hip).  This file is an A/B partner of the shipped
// pfb_spec.hip kernels: it is linked into libsnout_rx_ab.so (make ab), not into the product library.
// =============================================================================================
#ifdef SNOUT_PFB_T128
#define SNOUT_PFB_DYN_LDS(MM) ((MM) == 40 ? (size_t)PfbGeom<40>::T * 41u * 8u : (size_t)0)
#else
#define SNOUT_PFB_DYN_LDS(MM) 0
#endif

#ifdef SNOUT_PFB_STAMPS
extern "C" int snout_debug_pfb_times(unsigned long long* out, uint32_t n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pfb_times), (size_t)n * 8u, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -4;
}
extern "C" int snout_debug_pfb_stamps(unsigned long long* out, uint32_t n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pfb_stamps), (size_t)n * 8u, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -4;
}
#endif

uint32_t pfb_valu_tile(uint32_t M) { return M == 40 ? (uint32_t)PfbGeom<40>::T : (uint32_t)PfbGeom<16>::T; }

// mode: 0 channel IQ, 1 BTLE planes (M = 40), 2 802.15.4 rows (M = 16); grid = workgroups (segs.wgs_per_seg per segment)
int pfb_valu_launch(uint32_t M, int mode, int fmt, uint32_t grid, hipStream_t st, const PfbMfArgs& a,
                    const float* twM, const float* tw5)
{
#define SNOUT_PFB_F(MM, FU, F)                                                                          \
    hipLaunchKernelGGL((pfb_channelize<MM, FU, F>), dim3(grid), dim3(PfbGeom<MM>::NT), SNOUT_PFB_DYN_LDS(MM), st, a.segs, a.n, a.n_out, \
                       a.n_tiles, a.tiles_per_wg, a.proto, twM, tw5, a.y, a.y_stride, a.planes16, a.plane_stride, a.zb)
#define SNOUT_PFB(MM, FU)                                                                               \
    do {                                                                                              \
        if (fmt == kFmtSc8) SNOUT_PFB_F(MM, FU, kFmtSc8);                                              \
        else if (fmt == kFmtSc16) SNOUT_PFB_F(MM, FU, kFmtSc16);                                       \
        else SNOUT_PFB_F(MM, FU, kFmtCf32);                                                            \
    } while (0)
    if (M == 40 && mode == 1) SNOUT_PFB(40, true);
    else if (M == 40 && mode == 0) SNOUT_PFB(40, false);
    else if (M == 16 && mode == 2) SNOUT_PFB(16, true);
    else if (M == 16 && mode == 0) SNOUT_PFB(16, false);
    else return SNOUT_EINVAL;
#undef SNOUT_PFB
#undef SNOUT_PFB_F
    SNOUT_HIP(hipGetLastError());
    return 0;
}

}  // namespace snout
