// pfb_spec.hip — the channelizer (M = 40 and M = 16) as ONE workgroup of specialised waves per CU (gfx950).
//
// Same arithmetic contract as pfb.hip / oracle/oracle_pfb.c (the channelizer replaces the one-channel hop of
// snout/core/radio.py:415, snout/util/btle.py:62: SURVEY.md §8d cfg #3), other division of labour, decided by
// what round 3 measured (profiles/r3_*):
//   * f32 MFMA and VALU instructions of two waves on one SIMD do NOT issue side by side on gfx950
//     (tools/coissue_probe.hip: both = the sum of each alone), so the FIR stays on the vector pipe as the packed-
//     FMA sliding dot product of pfb.hip (the banded-Toeplitz matrix-pipe FIR of pfb_mfma.hip costs twice the
//     cycles for its structural zeros and hides none of them);
//   * pfb.hip's phases (FIR | FFT pass 3a | pass 3b) run in lock step with three barriers per tile, an LDS round
//     trip inside the FFT, and five-wave workgroups that put two waves on one SIMD: ~49 % of the vector issue
//     slots are used.  Here the FIR of tile i runs BESIDE the FFTs of tiles i-1 .. i-3 on other waves:
//
//   FIR + staging waves (M = 40: waves 0-9, M = 16: 0-3): thread <-> (branch, output parity, group of 8 outputs) as in
//               pfb.hip, 128 output times per tile, FIR outputs double-buffered in LDS; the same threads fetch the input
//               two tiles ahead (two register sets) and stage it behind their FIR.
//   FFT waves (M = 40: waves 10-15, M = 16: 4-15): a thread owns one output time of a 64-time block: 20 ds_read_b128 of
//               its row, then the whole 8 x 5 (prime-factor, no twiddles) / 4 x 4 FFT of oracle_pfb.c in registers (no LDS round trip, no barrier
//               inside) and the epilogue on its results; a block is spread over three (six) tile times, barriers inside
//               the straight-line code, so six (twelve) blocks are always in flight.
//               BTLE: lane = 16 (m mod 4) + (m / 4 mod 16): y[m+4] is the next lane of the DPP row, and the 64
//               lanes' hard bits of one channel come out of v_cmp as four 16-symbol pieces of its plane words;
//               the last symbol of each piece needs the next block's first output times: those go through LDS.
//               802.15.4: the FM discriminator of the thread's 16 channels (y[m-1] = the lane below, whole-wave DPP shift)
//               and the IIR sub-block sums of the block.
//   One s_barrier per tile.  Waves w, w + 4, w + 8, w + 12 share a SIMD (cyclic placement): the layouts below balance
//   FIR and FFT waves over the four SIMDs.  (A 12-wave layout with 16 outputs per FIR thread and one idle wave is kept
//   as an A/B partner: SNOUT_PFB_IMPL=spec12.)
#include "common.h"
#include <hip/hip_ext.h>
#include "iq_fmt.h"
#include "zb_discrim.h"
#include <type_traits>
#include "pfb_tables.inc"

namespace snout {

typedef float v2f __attribute__((ext_vector_type(2)));
// v_writelane_b32 as the compiler's own instruction (this clang has no __builtin_amdgcn_writelane): as inline asm
// the hazard recogniser does not see it, and on gfx950 a VALU read of an SGPR needs two wait states behind the
// v_cmp that wrote it.
extern "C" __device__ int __llvm_amdgcn_writelane(int, int, int) __asm("llvm.amdgcn.writelane");

#ifdef SNOUT_MF_STAMPS
// Diagnostic build only (tools/mf_stamps.py): shader cycles each wave spends in its phases, summed over the tiles
// of a workgroup: [block][wave][slot]; slot 7 = the wave's whole run, slot 6 = shader clock in kHz.
__device__ unsigned long long g_sp_stamps[256 * 16 * 8];
#define SP_STAMP(k)                                                              \
    do {                                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();           \
        st_acc[k] += now_ - st_last; st_last = now_;                             \
    } while (0)
#else
#define SP_STAMP(k) do { } while (0)
#endif

namespace sp {

struct cf { float re, im; };
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

// ---- The same FFT on (re, im) register pairs with the packed f32 instructions (v_pk_add / mul / fma_f32: both halves of a
// complex value in one issue slot; op_sel swaps the halves and neg_lo / neg_hi negate one of them, so +-j rotations and
// subtractions are free).  Every component goes through exactly the operations of cf's functions above, in the same order
// (a - b = a + (-b), (-x) c = -(x c) and a + b = b + a are exact), so the results are the same bits.  On gfx950 a packed
// instruction occupies the vector pipe for as long as the two plain ones it replaces (tools/fmabench.hip: v_fma_f32 and
// v_pk_fma_f32 reach the same flop rate), but a wave issues one instruction every five to six cycles at best, and the FFT
// waves' 1 000 issue slots per 64-time block become 580.
#ifndef SNOUT_SP_PACKED
#define SNOUT_SP_PACKED 1
#endif
typedef v2f pc;
__device__ __forceinline__ uint64_t pk2(float lo, float hi) { return ((uint64_t)__float_as_uint(hi) << 32) | (uint64_t)__float_as_uint(lo); }
__device__ __forceinline__ pc padd(pc a, pc b) { pc r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ pc psub(pc a, pc b) { pc r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a - j b = {a.re + b.im, a.im - b.re};  a + j b = {a.re - b.im, a.im + b.re}
__device__ __forceinline__ pc padd_mj(pc a, pc b) { pc r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ pc padd_pj(pc a, pc b) { pc r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// {a.re + b.lo, a.im - b.hi};  {a.re - b.lo, a.im + b.hi}
__device__ __forceinline__ pc padd_nh(pc a, pc b) { pc r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ pc padd_nl(pc a, pc b) { pc r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// {a.im - a.re, a.re + a.im}
__device__ __forceinline__ pc pdiff_sum(pc a) { pc r; asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(r) : "v"(a)); return r; }
// a * k.lo / a * k.hi (both halves);  k = an SGPR pair
__device__ __forceinline__ pc pmul_lo(pc a, uint64_t k) { pc r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "s"(k)); return r; }
__device__ __forceinline__ pc pmul_hi(pc a, uint64_t k) { pc r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "s"(k)); return r; }
// k.lo * a + c;  k.hi * a + c;  -(k.lo) * a + c
__device__ __forceinline__ pc pfma_lo(uint64_t k, pc a, pc c) { pc r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "s"(k), "v"(a), "v"(c)); return r; }
__device__ __forceinline__ pc pfma_hi(uint64_t k, pc a, pc c) { pc r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "s"(k), "v"(a), "v"(c)); return r; }
__device__ __forceinline__ pc pfma_nlo(uint64_t k, pc a, pc c) { pc r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "s"(k), "v"(a), "v"(c)); return r; }
// a (c + j d): re = fma(a.re, c, -(a.im d)), im = fma(a.re, d, a.im c);  k = {c, d}
__device__ __forceinline__ pc pmul_tw(pc a, uint64_t k)
{
    pc t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(k));                       // {a.im d, a.im c}
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "s"(k), "v"(t));       // {a.re c - t.lo, a.re d + t.hi}
    return r;
}
__device__ __forceinline__ void dft4(const pc b[4], pc X[4])
{
    const pc s0 = padd(b[0], b[2]), s1 = psub(b[0], b[2]);
    const pc s2 = padd(b[1], b[3]), s3 = psub(b[1], b[3]);
    X[0] = padd(s0, s2);
    X[2] = psub(s0, s2);
    X[1] = padd_mj(s1, s3);
    X[3] = padd_pj(s1, s3);
}
__device__ __forceinline__ void dft8(const pc a[8], pc X[8])
{
    const uint64_t cc = pk2(0.70710678118654752440f, 0.70710678118654752440f);
    const pc e[4] = {a[0], a[2], a[4], a[6]}, o[4] = {a[1], a[3], a[5], a[7]};
    pc E[4], O[4];
    dft4(e, E);
    dft4(o, O);
    const pc T1 = pmul_lo(padd_mj(O[1], O[1]), cc);          // {(re + im) c, (im - re) c}
    const pc U3 = pmul_lo(pdiff_sum(O[3]), cc);              // {(im - re) c, (re + im) c}: T[3] = {U3.lo, -U3.hi}
    X[0] = padd(E[0], O[0]);     X[4] = psub(E[0], O[0]);
    X[1] = padd(E[1], T1);       X[5] = psub(E[1], T1);
    X[2] = padd_mj(E[2], O[2]);  X[6] = padd_pj(E[2], O[2]);  // T[2] = -j O[2]
    X[3] = padd_nh(E[3], U3);    X[7] = padd_nl(E[3], U3);
}
// cs = {C1, C2}, ss = {S1, S2}
__device__ __forceinline__ void dft5(const pc b[5], pc X[5], const uint64_t cs, const uint64_t ss)
{
    const pc t1 = padd(b[1], b[4]), t2 = padd(b[2], b[3]), t3 = psub(b[1], b[4]), t4 = psub(b[2], b[3]);
    X[0] = padd(padd(b[0], t1), t2);
    const pc a1 = pfma_hi(cs, t2, pfma_lo(cs, t1, b[0]));
    const pc a2 = pfma_lo(cs, t2, pfma_hi(cs, t1, b[0]));
    const pc s1 = pfma_hi(ss, t4, pmul_lo(t3, ss));
    const pc s2 = pfma_nlo(ss, t4, pmul_hi(t3, ss));
    X[1] = padd_mj(a1, s1);
    X[4] = padd_pj(a1, s1);
    X[2] = padd_mj(a2, s2);
    X[3] = padd_pj(a2, s2);
}
__device__ __forceinline__ float re_of(cf a) { return a.re; }
__device__ __forceinline__ float im_of(cf a) { return a.im; }
__device__ __forceinline__ float re_of(pc a) { return a.x; }
__device__ __forceinline__ float im_of(pc a) { return a.y; }
#if SNOUT_SP_PACKED
typedef pc cx;
__device__ __forceinline__ cx mk(float re, float im) { return pc{re, im}; }
__device__ __forceinline__ cx mul_tw(cx a, float c, float d) { return pmul_tw(a, pk2(c, d)); }
#else
typedef cf cx;
__device__ __forceinline__ cx mk(float re, float im) { return cf{re, im}; }
__device__ __forceinline__ cx mul_tw(cx a, float c, float d) { return cmul_tw(a, c, d); }
#endif

// Workgroup barrier that orders LDS traffic only (vector-memory loads and stores stay in flight).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Wave layouts (W = waves per workgroup; waves w, w + 4, w + 8, w + 12 share a SIMD):
//   M = 40, W = 12: waves 0-4 FIR (16 outputs per thread), 8 idle, 5-7 and 9-11 FFT   -- 3 waves per SIMD, <= 168 registers
//   M = 40, W = 16: waves 0-9 FIR ( 8 outputs per thread), 10-15 FFT                  -- 4 waves per SIMD, <= 128 registers:
//       the SIMDs hold 3 FIR + 1 FFT, 3 + 1, 2 + 2, 2 + 2 waves (per tile a FIR wave issues 128 packed FMAs + 20 plain
//       instructions, an FFT wave 113 packed + 75 plain), and a plain VALU stream reaches 81 % of its peak at four waves
//       per SIMD against 73 % at three (tools/fmabench.hip): 2.6 against 2.8 ms per 8e8 samples (round 3).
//   M = 16, W = 16: waves 0-3 FIR (8 outputs per thread), 4-15 FFT + 802.15.4 discriminator (a block is spread over six tile
//       times): one FIR and three FFT waves on every SIMD.
//   M = 16 variants (A/B builds, tools/mf_ab16.sh): SNOUT_SP16_LAYOUT = 1: 8 FIR waves (4 outputs per thread) + 8 FFT waves, a
//       block over four tile times; 2: 256-time tiles, 8 FIR (8 outputs) + 8 FFT waves, a block over two tile times.
template <int M, int W> struct Layout;
template <> struct Layout<40, 12> { static constexpr int T = 128, FIR = 5, OUT = 16, FFT = 6, PERIOD = 3; };
template <> struct Layout<40, 16> { static constexpr int T = 128, FIR = 10, OUT = 8, FFT = 6, PERIOD = 3; };
#ifndef SNOUT_SP16_LAYOUT
#define SNOUT_SP16_LAYOUT 0
#endif
#if SNOUT_SP16_LAYOUT == 1
template <> struct Layout<16, 16> { static constexpr int T = 128, FIR = 8, OUT = 4, FFT = 8, PERIOD = 4; };
#elif SNOUT_SP16_LAYOUT == 2
template <> struct Layout<16, 16> { static constexpr int T = 256, FIR = 8, OUT = 8, FFT = 8, PERIOD = 2; };
#else
template <> struct Layout<16, 16> { static constexpr int T = 128, FIR = 4, OUT = 8, FFT = 12, PERIOD = 6; };
#endif
}  // namespace sp

// Output modes: channel IQ | BTLE hard bits into the planes (M = 40) | 802.15.4 discriminator rows + IIR sums (M = 16)
constexpr int kSpIq = 0, kSpBtle = 1, kSpZb = 2;

template <int M, int MODE, int FMT, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(W / 4, W / 4)))
void pfb_spec(const PfbMfArgs A)
{
    using namespace sp;
    using L_ = Layout<M, W>;
    constexpr int T = L_::T, BPT = T / 64;                              // output times per tile, 64-time blocks per tile
    constexpr int kFirWaves = L_::FIR, OUT = L_::OUT, NG = T / 2 / OUT; // NG groups of OUT outputs per (branch, parity) and tile
    constexpr int kFftWaves = L_::FFT, PERIOD = L_::PERIOD;
    constexpr int M1 = M == 40 ? 8 : 4, M2 = M == 40 ? 5 : 4, ROW = M == 40 ? 42 : 18, D = M / 2, P = 16;
    constexpr bool PHASE_MAJOR = M == 40;               // rows in the FFT waves' lane order (M = 40) or in time order
    constexpr int SPAN = (T - 1) * D + M * P, NEW = T * D, OV = SPAN - NEW;
    constexpr int NST = 64 * kFirWaves;                 // threads that compute the FIR and stage
    static_assert(NST == 2 * M * NG && (NEW / 2) % NST == 0 && OV / 2 <= NST, "FIR / staging thread map");
    static_assert(kFftWaves == BPT * PERIOD && (M != 40 || T == 128) && (MODE != kSpBtle || M == 40) && (MODE != kSpZb || M == 16), "layout");
    static_assert(OUT % 4 == 0 && (PERIOD == 6 || PERIOD == 4 || PERIOD == 3 || PERIOD == 2), "layout");
    constexpr bool BT = MODE == kSpBtle, ZB = MODE == kSpZb;
    // A/B switch, measured and left OFF (profiles/r3_spec_dma_ab.txt): cf32 input global -> LDS by LDS-DMA
    // (global_load_lds_dwordx4: no registers, no ds_write), the whole span of tile i + 1 issued at the start of tile time i
    // (its overlap with tile i out of L2 again instead of an LDS -> LDS copy), by the FFT waves (1) or the FIR waves (2).
    // With two xs buffers the DMA can only run ONE tile ahead and has to land before the next barrier; the register path
    // prefetches two tiles ahead and is 2-12 % faster.  The integer formats need the conversion and use registers anyway.
#ifndef SNOUT_SP_DMA
#define SNOUT_SP_DMA 0
#endif
    constexpr bool DMA = SNOUT_SP_DMA != 0 && FMT == kFmtCf32;

    __shared__ float2 xs[2][SPAN];
    __shared__ float2 us[2][T * ROW];
    // BTLE hand-over: y of a block's first four output times ([block parity][channel][phase]) and of its last four, per FFT
    // wave.  One predicated store of the epilogue writes both (lanes 0 / 16 / 32 / 48 the first, 15 / 31 / 47 / 63 the last
    // four times, 32 contiguous bytes each): `last` sits 8 banks behind `first` (the pad), so the two groups of a store
    // fall on different banks -- with both arrays on bank 0 (rounds 3-4: two arrays of 1 280-byte rows) every one of the
    // 40 stores per block was a two-way conflict (SQ_LDS_BANK_CONFLICT 7.4e7 -> 1.5e8 per launch).
    struct EdgeY { float2 first[2][BT ? M * 4 : 1]; float2 pad[4]; float2 last[BT ? M * 4 : 1]; };
    __shared__ EdgeY edge_y[BT ? kFftWaves : 1];
    // 802.15.4: fast_atan2f table, IIR weights, y of each block's last output time, each wave's d values (for the S_j sums)
#ifndef SNOUT_ATAN_PAIR
#define SNOUT_ATAN_PAIR 0
#endif
    __shared__ float atan_s[(ZB && !SNOUT_ATAN_PAIR) ? 257 : 1];
    __shared__ float2 atan_p[(ZB && SNOUT_ATAN_PAIR) ? 256 : 1];
    __shared__ double wts_s[ZB ? 64 : 1];
    __shared__ float2 ylast[ZB ? kFftWaves + 1 : 1][ZB ? M : 1];       // (row kFftWaves: zeros, "the block before" of the first block)
    __shared__ float dls[ZB ? kFftWaves : 1][ZB ? M * 65 : 1];

    const uint32_t seg = blockIdx.x / A.segs.wgs_per_seg, bid = blockIdx.x - seg * A.segs.wgs_per_seg;
    const void* __restrict__ x = A.segs.x[seg];
    uint16_t* planes16 = A.planes16 ? A.planes16 + (uint64_t)seg * A.segs.planes_seg : nullptr;
    const uint64_t n = A.n, n_out = A.n_out;
    const uint32_t n_tiles = A.n_tiles;

    const uint32_t t_begin = bid * A.tiles_per_wg;
    uint32_t t_end = t_begin + A.tiles_per_wg;
    if (t_end > n_tiles) t_end = n_tiles;
    if (t_begin >= t_end) return;
    // tiles this workgroup computes: its range, plus (BTLE) the tile behind it, whose first four output times complete the
    // range's last symbols, or (802.15.4) the tile before it, whose last output time the first discriminator value needs
    const uint32_t t_lo = (ZB && t_begin > 0u) ? t_begin - 1u : t_begin;
    const uint32_t t_hi = (BT && t_end < n_tiles) ? t_end + 1u : t_end;
    const int NTL = (int)(t_hi - t_lo);
    const int IT = NTL + PERIOD + 1;                     // barriers after the first one (pipeline drain included)

    // M = 40: w in an SGPR -- the roles' branches and the FFT waves' block loops become scalar branches (- 1.5 %; at M = 16 the
    // same costs 1 %: profiles/r4_spec_ab.txt)
    const int t = threadIdx.x, w = M == 40 ? __builtin_amdgcn_readfirstlane(t >> 6) : (t >> 6), l = t & 63;
    if constexpr (ZB) {
        if (SNOUT_ATAN_PAIR) { for (int i = t; i < 256; i += 64 * W) atan_p[i] = make_float2(A.zb.atan_tab[i], A.zb.atan_tab[i + 1]); }
        else { for (int i = t; i < 257; i += 64 * W) atan_s[i] = A.zb.atan_tab[i]; }
        if (t < 64) wts_s[t] = A.zb.iir_w[t];
        if (t < M) ylast[kFftWaves][t] = make_float2(0.0f, 0.0f);
    }
#ifdef SNOUT_MF_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_last, st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // LDS-DMA of tile nt's span into xs[nt & 1]: 1 KiB pieces (64 lanes x 16 B, LDS destination = piece base + 16 lane),
    // piece p by wave p mod CNT of the issuing role.  A span that reaches past the segment's end is staged through registers
    // instead (zeros past n).  Ordered for the FIR waves' reads by this wave's vmcnt(0) before the next barrier.
    auto dma_tile = [&](int nt, int idx, auto cntc) {
        constexpr int CNT = decltype(cntc)::value;
        constexpr int NQ = SPAN / 2, NP = (NQ + 63) / 64;          // float4 = sample pairs of the span, pieces
        const uint64_t base = (uint64_t)(t_lo + (uint32_t)nt) * NEW;
        const bool full = base + SPAN <= n;
        const uint32_t lds0 = (uint32_t)(uintptr_t)&xs[nt & 1][0];
#pragma unroll
        for (int p0 = 0; p0 < NP; p0 += CNT) {
            const int p = __builtin_amdgcn_readfirstlane(p0 + idx);
            if (p < NP) {
                const int q = 64 * p + l;
                if (full) {
                    // address = wave-uniform base (SGPR pair) + 16 lane; LDS destination = M0 + 16 lane
                    const uint64_t gb = (uint64_t)(uintptr_t)x + (base + 128ull * (uint64_t)p) * 8ull;
                    const uint64_t gbase = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb) |
                                           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32)) << 32);
                    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)p * 1024u));
                    if (q < NQ) {
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(16 * l), "s"(gbase), "s"(dst) : "memory");
                    }
                } else if (q < NQ) {
                    const uint64_t g = base + 2ull * (uint64_t)q;
                    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (g + 1 < n) v = *reinterpret_cast<const float4*>(reinterpret_cast<const float2*>(x) + g);
                    else if (g < n) { const float2 a = reinterpret_cast<const float2*>(x)[g]; v.x = a.x; v.y = a.y; }
                    reinterpret_cast<float4*>(&xs[nt & 1][0])[q] = v;
                }
            }
        }
    };
    if (w < kFirWaves) {
        // =====================================================================================
        // FIR + staging waves: thread <-> (branch r, output parity e, group grp): the 16 outputs
        // m = e + 2 (16 grp + i) of one branch are a sliding dot product over z[q] = x[r + e D + q M]
        // =====================================================================================
        using Raw = typename IqRaw<FMT>::pair;
#ifdef SNOUT_SP_PRIO_FIR16
        if (M == 16) __builtin_amdgcn_s_setprio(SNOUT_SP_PRIO_FIR16);
#endif
        const int tf = w * 64 + l, r = tf % M, e = (tf / M) & 1, grp = tf / (2 * M);
        v2f hp[P / 2];
#pragma unroll
        for (int p = 0; p < P / 2; p++) hp[p] = v2f{A.proto[r + (2 * p) * M], A.proto[r + (2 * p + 1) * M]};
        const uint32_t rd = (uint32_t)((r + e * D + OUT * grp * M) * 8);                      // window start, bytes
        // row of output m = e + 2 (OUT grp + i), rows in the FFT waves' lane order (64 (m / 64) + 16 (m mod 4) + (m mod 64) / 4):
        //   OUT = 16: 64 (grp / 2) + 16 (e + 2 (i & 1)) + 8 (grp & 1) + i / 2;   OUT = 8: 64 (grp / 4) + 16 (e + 2 (i & 1)) + 4 (grp & 3) + i / 2
        //   time order (M = 16): row m
        const uint32_t wr = !PHASE_MAJOR ? (uint32_t)(((e + 2 * OUT * grp) * ROW + r) * 8)
                          : OUT == 16 ? (uint32_t)(((64 * (grp >> 1) + 16 * e + 8 * (grp & 1)) * ROW + r) * 8)
                                      : (uint32_t)(((64 * (grp >> 2) + 16 * e + 4 * (grp & 3)) * ROW + r) * 8);

        auto load_pair = [&](uint64_t g) -> Raw {                  // samples g, g+1 (g even), zero past n
            if (g + 1 < n) return iq_pair_raw<FMT>(x, g);
            Raw v = Raw{};
            if (g < n) v = iq_single_raw<FMT>(x, g);
            return v;
        };
        constexpr int NPRE = NEW / 2 / NST;                         // 4 new pairs per thread and tile
        Raw pre[2][NPRE];
        auto fetch = [&](Raw (&set)[NPRE], uint32_t tile) {
            const uint64_t in1 = (uint64_t)tile * NEW + OV;
            if (in1 + NEW <= n) {
#pragma unroll
                for (int k = 0; k < NPRE; k++) set[k] = iq_pair_raw<FMT>(x, in1 + 2ull * (uint64_t)(tf + k * NST));
            } else {
#pragma unroll
                for (int k = 0; k < NPRE; k++) set[k] = load_pair(in1 + 2ull * (uint64_t)(tf + k * NST));
            }
        };
        auto stage = [&](Raw (&set)[NPRE], int nb) {                // xs[nb]: overlap from xs[nb ^ 1], new samples from the set
            float4* dst = reinterpret_cast<float4*>(&xs[nb][0]);
            const float4* src = reinterpret_cast<const float4*>(&xs[nb ^ 1][0]);
            if (tf < OV / 2) dst[tf] = src[tf + NEW / 2];
#pragma unroll
            for (int k = 0; k < NPRE; k++) dst[OV / 2 + tf + k * NST] = iq_pair_cvt<FMT>(set[k]);
        };
        auto fir_tile = [&](auto bufc) {
            constexpr int BUF = decltype(bufc)::value;
            const uint32_t a0 = (uint32_t)(uintptr_t)&xs[BUF][0] + rd;
            char* uo = reinterpret_cast<char*>(&us[BUF][0]) + wr;
            // OUT outputs from one window of OUT + 15 samples: the first four outputs need samples 0..18, every further
            // four outputs four more, read while the four before them are computed
            v2f wv[OUT + P - 1];
            // the 19 reads and their wait as ONE statement (the values must not be touched between the two; the
            // prefetches inside the loop below stay split from their waits on purpose -- they run under the FMAs -- and
            // tests/test_asm_hazards.py checks the ISA for a use in between)
            if constexpr (M == 40) {
                asm volatile("ds_read_b64 %0, %19 offset:0\n\t"
                             "ds_read_b64 %1, %19 offset:320\n\t"
                             "ds_read_b64 %2, %19 offset:640\n\t"
                             "ds_read_b64 %3, %19 offset:960\n\t"
                             "ds_read_b64 %4, %19 offset:1280\n\t"
                             "ds_read_b64 %5, %19 offset:1600\n\t"
                             "ds_read_b64 %6, %19 offset:1920\n\t"
                             "ds_read_b64 %7, %19 offset:2240\n\t"
                             "ds_read_b64 %8, %19 offset:2560\n\t"
                             "ds_read_b64 %9, %19 offset:2880\n\t"
                             "ds_read_b64 %10, %19 offset:3200\n\t"
                             "ds_read_b64 %11, %19 offset:3520\n\t"
                             "ds_read_b64 %12, %19 offset:3840\n\t"
                             "ds_read_b64 %13, %19 offset:4160\n\t"
                             "ds_read_b64 %14, %19 offset:4480\n\t"
                             "ds_read_b64 %15, %19 offset:4800\n\t"
                             "ds_read_b64 %16, %19 offset:5120\n\t"
                             "ds_read_b64 %17, %19 offset:5440\n\t"
                             "ds_read_b64 %18, %19 offset:5760\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(wv[0]), "=&v"(wv[1]), "=&v"(wv[2]), "=&v"(wv[3]), "=&v"(wv[4]), "=&v"(wv[5]), "=&v"(wv[6]), "=&v"(wv[7]), "=&v"(wv[8]), "=&v"(wv[9]), "=&v"(wv[10]), "=&v"(wv[11]), "=&v"(wv[12]), "=&v"(wv[13]), "=&v"(wv[14]), "=&v"(wv[15]), "=&v"(wv[16]), "=&v"(wv[17]), "=&v"(wv[18])
                             : "v"(a0) : "memory");
            } else {
                asm volatile("ds_read_b64 %0, %19 offset:0\n\t"
                             "ds_read_b64 %1, %19 offset:128\n\t"
                             "ds_read_b64 %2, %19 offset:256\n\t"
                             "ds_read_b64 %3, %19 offset:384\n\t"
                             "ds_read_b64 %4, %19 offset:512\n\t"
                             "ds_read_b64 %5, %19 offset:640\n\t"
                             "ds_read_b64 %6, %19 offset:768\n\t"
                             "ds_read_b64 %7, %19 offset:896\n\t"
                             "ds_read_b64 %8, %19 offset:1024\n\t"
                             "ds_read_b64 %9, %19 offset:1152\n\t"
                             "ds_read_b64 %10, %19 offset:1280\n\t"
                             "ds_read_b64 %11, %19 offset:1408\n\t"
                             "ds_read_b64 %12, %19 offset:1536\n\t"
                             "ds_read_b64 %13, %19 offset:1664\n\t"
                             "ds_read_b64 %14, %19 offset:1792\n\t"
                             "ds_read_b64 %15, %19 offset:1920\n\t"
                             "ds_read_b64 %16, %19 offset:2048\n\t"
                             "ds_read_b64 %17, %19 offset:2176\n\t"
                             "ds_read_b64 %18, %19 offset:2304\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(wv[0]), "=&v"(wv[1]), "=&v"(wv[2]), "=&v"(wv[3]), "=&v"(wv[4]), "=&v"(wv[5]), "=&v"(wv[6]), "=&v"(wv[7]), "=&v"(wv[8]), "=&v"(wv[9]), "=&v"(wv[10]), "=&v"(wv[11]), "=&v"(wv[12]), "=&v"(wv[13]), "=&v"(wv[14]), "=&v"(wv[15]), "=&v"(wv[16]), "=&v"(wv[17]), "=&v"(wv[18])
                             : "v"(a0) : "memory");
            }
#pragma unroll
            for (int i0 = 0; i0 < OUT; i0 += 4) {
                if (i0 < OUT - 4) {
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
                if (i0 < OUT - 4)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wv[i0 + 19]), "+v"(wv[i0 + 20]), "+v"(wv[i0 + 21]), "+v"(wv[i0 + 22]) :: "memory");
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = i0 + j;
                    *reinterpret_cast<float2*>(uo + (PHASE_MAJOR ? 32 * (i & 1) + (i >> 1) : 2 * i) * ROW * 8) = make_float2(acc[j].x, acc[j].y);
                }
            }
        };
        // ---- prologue: the whole span of the first tile; the second tile's new samples requested
        {
            float4* xb = reinterpret_cast<float4*>(&xs[0][0]);
            const uint64_t in0 = (uint64_t)t_lo * NEW;
            for (uint32_t q = (uint32_t)tf; q < (uint32_t)(SPAN / 2); q += NST) xb[q] = iq_pair_cvt<FMT>(load_pair(in0 + 2ull * q));
            if (!DMA && 1 < NTL) fetch(pre[1], t_lo + 1u);
        }
        lds_barrier();
        for (int it = 0; it < IT; it += 2) {
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
                const int i2 = it + hb;
                if (i2 < IT) {
                    if constexpr (DMA && SNOUT_SP_DMA == 2) { if (i2 + 1 < NTL) dma_tile(i2 + 1, w, std::integral_constant<int, kFirWaves>{}); }
                    if (i2 < NTL) {
                        // tile i2 + 2's samples: two tiles ahead into the set tile i2's came from; tile i2 + 1's (requested
                        // one tile ago) go to LDS behind this tile's FIR: a load has one tile time + the FIR to arrive
                        if (!DMA && i2 + 2 < NTL) fetch(pre[hb], t_lo + (uint32_t)i2 + 2u);
                        SP_STAMP(0);
#ifndef SNOUT_SP_STAGGER
#define SNOUT_SP_STAGGER 0
#endif
                        // Stagger (A/B switch): the upper half of the FIR waves stages the next tile BEFORE its FIR, so that the
                        // barrier does not release ten waves into the same burst of window reads and the two halves' LDS reads /
                        // FMAs / LDS writes interleave (stage() only reads this tile's input buffer and writes the next one's).
                        const bool stage_first = SNOUT_SP_STAGGER != 0 && w >= kFirWaves / 2;
                        if (stage_first && !DMA && i2 + 1 < NTL) stage(pre[hb ^ 1], hb ^ 1);
                        if (hb == 0) fir_tile(std::integral_constant<int, 0>{});
                        else         fir_tile(std::integral_constant<int, 1>{});
                        SP_STAMP(1);
                        if (!stage_first && !DMA && i2 + 1 < NTL) stage(pre[hb ^ 1], hb ^ 1);
                        SP_STAMP(4);
                    }
                    if constexpr (DMA && SNOUT_SP_DMA == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lds_barrier();
                    SP_STAMP(2);
                }
            }
        }
    } else if (M == 40 && W == 12 && w == 8) {
        // wave 8 shares its SIMD with FIR waves 0 and 4: it stays idle and only keeps the barriers' count
        for (int i = 0; i < IT + 1; i++) lds_barrier();
    } else {
        // =====================================================================================
        // FFT waves: a thread owns one output time of a 64-time block
        // =====================================================================================
        // Wave f takes the blocks b = f + 6 k (block b = 64-time half b & 1 of tile b >> 1): tile j = f / 2 + 3 k, half
        // f & 1; a block takes three tile times: Q1 (row, first stage) | barrier | Q2 | barrier | Q3 | barrier.
        const int f = (M == 40 && W == 12) ? (w < 8 ? w - kFirWaves : w - kFirWaves - 1) : w - kFirWaves, j0 = f / BPT, half = f % BPT;
        // The FFT is dependency chains; the FIR waves' 256 independent packed FMAs per tile are always ready and, being the
        // older waves, would win every arbitration: FFT waves issue first (priority, then age), the FIR fills their gaps.
#ifndef SNOUT_SP_PRIO_FFT
#define SNOUT_SP_PRIO_FFT 2
#endif
#ifndef SNOUT_SP_PRIO_FFT16
#define SNOUT_SP_PRIO_FFT16 SNOUT_SP_PRIO_FFT
#endif
        __builtin_amdgcn_s_setprio(M == 16 ? SNOUT_SP_PRIO_FFT16 : SNOUT_SP_PRIO_FFT);
        const float* const tw = M == 40 ? kTw40 : kTw16;
        const float c5_1 = kTw5[2], c5_2 = kTw5[4], s5_1 = -kTw5[3], s5_2 = -kTw5[5];
        // output time of this lane within its block
        const uint32_t mloc = PHASE_MAJOR ? 4u * (uint32_t)(l & 15) + (uint32_t)(l >> 4) : (uint32_t)l;
        int itc = 0;
        auto bar = [&]() {
            SP_STAMP(3);
            if constexpr (DMA && SNOUT_SP_DMA == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            itc++;
            if constexpr (DMA && SNOUT_SP_DMA == 1) { if (itc < NTL) dma_tile(itc, f, std::integral_constant<int, kFftWaves>{}); }       // tile time itc - 1 begins: tile itc's samples
            SP_STAMP(2);
        };
        bar();                                                       // the FIR waves' prologue
        for (int i = 0; i <= j0; i++) bar();                         // barrier j + 2 ends tile j's FIR

        if constexpr (M == 40) {
            // BTLE: state of the block whose last symbols wait for the next block's first output times.
            // Lane k < 40 holds the 64 hard bits of channel k.
            uint32_t pm_lo = 0, pm_hi = 0;
            uint64_t pend_m0 = 0;
            int pend_par = 0;
            bool pending = false;
            auto finalize = [&]() {
                if constexpr (BT) {
                    if (pending && l < M) {
                        const uint64_t nbits = n_out >= 4 ? n_out - 4 : 0;           // bits exist for m < n_out - 4
                        // the block behind this one is the next wave's (wave 0's NEXT block behind wave 5's)
                        const int fs = f == kFftWaves - 1 ? 0 : f + 1;
                        const int spar = f == kFftWaves - 1 ? pend_par ^ 1 : pend_par;
                        const float4* la = reinterpret_cast<const float4*>(&edge_y[f].last[l * 4]);
                        const float4* fi = reinterpret_cast<const float4*>(&edge_y[fs].first[spar][l * 4]);
                        const float4 l01 = la[0], l23 = la[1], f01 = fi[0], f23 = fi[1];
                        const uint32_t b0 = (l01.x * f01.y) > (f01.x * l01.y) ? 1u : 0u;
                        const uint32_t b1 = (l01.z * f01.w) > (f01.z * l01.w) ? 1u : 0u;
                        const uint32_t b2 = (l23.x * f23.y) > (f23.x * l23.y) ? 1u : 0u;
                        const uint32_t b3 = (l23.z * f23.w) > (f23.z * l23.w) ? 1u : 0u;
                        const uint32_t lo = (pm_lo & 0x7FFF7FFFu) | (b0 << 15) | (b1 << 31);
                        const uint32_t hi = (pm_hi & 0x7FFF7FFFu) | (b2 << 15) | (b3 << 31);
                        // symbols whose sample exists: m0 + 4 sy + j < nbits, a prefix of the 16 per phase
                        const uint32_t left = nbits > pend_m0 ? (uint32_t)(nbits - pend_m0 < 64u ? nbits - pend_m0 : 64u) : 0u;
                        uint16_t* dst = planes16 + ((uint64_t)l * A.plane_stride + (pend_m0 >> 8) * 4u) * 4u + (uint32_t)((pend_m0 & 255u) >> 6);
    #pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t cnt = left > (uint32_t)j ? (left - (uint32_t)j + 3u) >> 2 : 0u;     // <= 16
                            const uint32_t v = ((j < 2 ? lo : hi) >> (16 * (j & 1))) & 0xFFFFu;
                            dst[4 * j] = (uint16_t)(v & ((1u << cnt) - 1u));
                        }
                    }
                    pending = false;
                }
            };

            // channel of (k1, k2) in the prime-factor output map: k = k1 mod 8, k = k2 mod 5
            auto kch = [](int k1, int k2) { return (25 * k1 + 16 * k2) % 40; };
            int kblk = 0;
            for (int j = j0; j < NTL; j += 3, kblk++) {
                const uint32_t tile = t_lo + (uint32_t)j;
                const uint64_t m0b = (uint64_t)tile * T + 64u * (uint32_t)half;
                // (as scalars by force: kept as lane masks the two flags cost a v_cndmask / v_cmp pair at every use)
                const bool emit = __builtin_amdgcn_readfirstlane((int)(tile < t_end)) != 0;   // the tile behind the range only supplies its first four times
                const bool need = __builtin_amdgcn_readfirstlane((int)(emit || half == 0)) != 0;
                const int par = kblk & 1;
                // ---- Q1: the row and the 8-point DFTs over n1 (inputs in the prime-factor order (5 n1 + 8 n2) mod 40)
                cx Bv[M2][M1];
                if (need) {
                    const float4* rowp = reinterpret_cast<const float4*>(&us[j & 1][(64 * half + l) * ROW]);
                    cx u[M];
    #pragma unroll
                    for (int q = 0; q < M / 2; q++) {
                        const float4 v = rowp[q];
                        u[2 * q] = mk(v.x, v.y);
                        u[2 * q + 1] = mk(v.z, v.w);
                    }
    #pragma unroll
                    for (int n2 = 0; n2 < M2; n2++) {
                        cx a[M1], X[M1];
    #pragma unroll
                        for (int n1 = 0; n1 < M1; n1++) a[n1] = u[(M2 * n1 + M1 * n2) % M];    // prime-factor input map
                        dft8(a, X);
    #pragma unroll
                        for (int k1 = 0; k1 < M1; k1++) {
                            Bv[n2][k1] = X[k1];                                                   // 8 and 5 are coprime: no twiddles
                        }
                    }
                }
                bar();
                // ---- Q2, Q3: 5-point DFTs over n2 per k1, epilogue per channel k = (25 k1 + 16 k2) mod 40
                finalize();                       // the block before this one: its successor's first output times are there now
                uint32_t m_lo = 0, m_hi = 0;
                const bool edge = (l & 15) == 0 || (l & 15) == 15;
                float2* const edge_slot = BT ? ((l & 15) == 0 ? &edge_y[f].first[par][l >> 4] : &edge_y[f].last[l >> 4]) : nullptr;
                auto do_k1 = [&](int k1) {
                    cx b[M2], Y[M2];
    #pragma unroll
                    for (int n2 = 0; n2 < M2; n2++) b[n2] = Bv[n2][k1];
#if SNOUT_SP_PACKED
                    dft5(b, Y, pk2(c5_1, c5_2), pk2(s5_1, s5_2));
#else
                    dft5(b, Y, c5_1, c5_2, s5_1, s5_2);
#endif
                    if constexpr (BT) {
                        // bit[m] = (I[m] Q[m+4]) > (I[m+4] Q[m]); m + 4 is the next lane of the 16-lane row.  The
                        // factor (-1)^{km} is the same for m and m + 4 and cancels in both products.
    #pragma unroll
                        for (int k2 = 0; k2 < M2; k2++) {
                            const int k = kch(k1, k2);
                            const float qn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(im_of(Y[k2])), 0x101, 0xF, 0xF, true));
                            const float in = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(re_of(Y[k2])), 0x101, 0xF, 0xF, true));
                            const uint64_t mk = __builtin_amdgcn_ballot_w64((re_of(Y[k2]) * qn) > (in * im_of(Y[k2])));
                            m_lo = (uint32_t)__llvm_amdgcn_writelane((int)(uint32_t)mk, k, (int)m_lo);
                            m_hi = (uint32_t)__llvm_amdgcn_writelane((int)(uint32_t)(mk >> 32), k, (int)m_hi);
                        }
                        // the block's first four output times (lanes 0, 16, 32, 48) and its last four (15, 31, 47, 63): one
                        // predicated run of stores at constant offsets from the lane's slot
                        if (edge) {
    #pragma unroll
                            for (int k2 = 0; k2 < M2; k2++) edge_slot[kch(k1, k2) * 4] = make_float2(re_of(Y[k2]), im_of(Y[k2]));
                        }
                    } else {
                        const uint64_t mg = m0b + mloc;
                        if (mg < n_out) {
    #pragma unroll
                            for (int k2 = 0; k2 < M2; k2++) {
                                const int k = kch(k1, k2);
                                float vr = re_of(Y[k2]), vi = im_of(Y[k2]);
                                if ((k & 1) && (mg & 1)) { vr = -vr; vi = -vi; }
                                A.y[(uint64_t)k * A.y_stride + mg] = make_float2(vr, vi);
                            }
                        }
                    }
                };
                if (need) {
    #pragma unroll
                    for (int k1 = 0; k1 < M1 / 2; k1++) do_k1(k1);
                }
                bar();
                if (need) {
    #pragma unroll
                    for (int k1 = M1 / 2; k1 < M1; k1++) do_k1(k1);
                    if constexpr (BT) {
                        pm_lo = m_lo; pm_hi = m_hi; pend_m0 = m0b; pend_par = par; pending = emit;
                    }
                }
                bar();
            }
            // ---- drain: the block behind the last one is finished one barrier later; then keep step with the FIR waves
            if (itc < IT + 1) bar();
            finalize();
        } else {
            // =================================================================================
            // M = 16: 4 x 4 FFT in registers, then (802.15.4) the FM discriminator of the thread's 16 channels
            // =================================================================================
            // Wave f takes the blocks b = f + 12 k: tile j = f / 2 + 6 k, half f & 1; six tile times per block:
            //   Q1 row + FFT (+ y of the block's last output time into LDS for the block behind it) | Q2..Q5 four channels'
            //   discriminator values each: d[m] = fast_atan2f(y[m] conj y[m-1]) (zb_discrim.h), y[m-1] from the lane below
            //   (whole-wave DPP shift; lane 0: the block before) | Q6 the IIR sub-block sums S_j of the block's 64 values
            //   per channel, in the oracle's order (four partial sums of 16 sequential terms, S = (P0 + P1) + (P2 + P3)).
            // (Other layouts: BPT blocks per tile, wave f <-> tile f / BPT + PERIOD k, part f % BPT; with fewer tile times per
            // block the quarters share them: PERIOD 4 = Q1 | Q2 Q3 | Q4 Q5 | Q6, PERIOD 2 = Q1 | Q2 .. Q6.)
            float* const dl = &dls[ZB ? f : 0][0];
            const int part = f % BPT;
            int kblk = 0;
            for (int j = f / BPT; j < NTL; j += PERIOD, kblk++) {
                const uint32_t tile = t_lo + (uint32_t)j;
                const uint64_t m0b = (uint64_t)tile * T + 64u * (uint32_t)part;
                const bool emit = __builtin_amdgcn_readfirstlane((int)(tile >= t_begin)) != 0;   // (wave-uniform) the tile before the range only supplies y[m0 - 1]
                const uint64_t mg = m0b + mloc;
                // ---- Q1: the row and the FFT; y_k[m] = (-1)^{km} X[k]
                cx y[M];
                {
                    const float4* rowp = reinterpret_cast<const float4*>(&us[j & 1][(64 * part + l) * ROW]);
                    cx u[M];
#pragma unroll
                    for (int q = 0; q < M / 2; q++) {
                        const float4 v = rowp[q];
                        u[2 * q] = mk(v.x, v.y);
                        u[2 * q + 1] = mk(v.z, v.w);
                    }
                    cx Bv[M2][M1];
#pragma unroll
                    for (int n2 = 0; n2 < M2; n2++) {
                        cx a[M1], X[M1];
#pragma unroll
                        for (int n1 = 0; n1 < M1; n1++) a[n1] = u[M2 * n1 + n2];
                        dft4(a, X);
#pragma unroll
                        for (int k1 = 0; k1 < M1; k1++) {
                            const int jj = (n2 * k1) % M;
                            Bv[n2][k1] = jj == 0 ? X[k1] : mul_tw(X[k1], tw[2 * jj], tw[2 * jj + 1]);    // literals
                        }
                    }
#pragma unroll
                    for (int k1 = 0; k1 < M1; k1++) {
                        cx b[M2], Y[M2];
#pragma unroll
                        for (int n2 = 0; n2 < M2; n2++) b[n2] = Bv[n2][k1];
                        dft4(b, Y);
#pragma unroll
                        for (int k2 = 0; k2 < M2; k2++) {
                            const int k = k1 + M1 * k2;
                            float vr = re_of(Y[k2]), vi = im_of(Y[k2]);
                            if ((k & 1) && (mg & 1)) { vr = -vr; vi = -vi; }
                            y[k] = mk(vr, vi);
                        }
                    }
                }
                if constexpr (ZB) {
                    if (l == 63) {
#pragma unroll
                        for (int k = 0; k < M; k++) ylast[f][k] = make_float2(im_of(y[k]), re_of(y[k]));      // (im, re): see below
                    }
                }
                bar();
                if constexpr (ZB) {
                    // y[m0 - 1]: of the block before this one (the wave before; wave 11's previous block for wave 0), zero
                    // in front of the very first block
                    // (one unconditional broadcast read per channel: the row of zeros stands in where there is no block before)
                    const bool have_prev = j > 0 || part > 0;
                    const float2* yp = &ylast[!have_prev ? kFftWaves : (f == 0 ? kFftWaves - 1 : f - 1)][0];
                    const uint32_t left = n_out > m0b ? (uint32_t)(n_out - m0b < 64u ? n_out - m0b : 64u) : 0u;   // outputs of this block that exist
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        // lane 0's y[m - 1] of the quarter's four channels in one go (stored (im, re): the shifted values then
                        // sit in the register order the packed products take them in)
                        float2 pv4[4];
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) pv4[kk] = yp[4 * q + kk];
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const int k = 4 * q + kk;
                            const float2 pv = pv4[kk];
                            float2 p;
                            p.y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(pv.x), __float_as_int(im_of(y[k])), 0x138, 0xF, 0xF, false));
                            p.x = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(pv.y), __float_as_int(re_of(y[k])), 0x138, 0xF, 0xF, false));
#if SNOUT_ATAN_PAIR
                            float v = zb_discriminate_with(make_float2(re_of(y[k]), im_of(y[k])), p, AtanPairs{atan_p});
#else
                            float v = zb_discriminate(make_float2(re_of(y[k]), im_of(y[k])), p, atan_s);
#endif
                            if (mloc >= left) v = 0.0f;
                            if (emit) A.zb.d[(uint64_t)seg * A.segs.d_seg + (uint64_t)k * A.zb.d_stride + mg] = v;
                            dl[k * 65 + l] = v;
                        }
                        if (PERIOD == 6 || (PERIOD == 4 && (q & 1))) bar();
                    }
                    // S_j of this block's sub-block of every channel: lane <-> (channel l / 4, quarter l & 3)
                    {
                        const int qu = l & 3, kk = l >> 2;
                        double acc = 0.0;
#pragma unroll
                        for (int i = 0; i < 16; i++)
                            acc = acc + wts_s[63 - (16 * qu + i)] * (double)dl[kk * 65 + 16 * qu + i];
                        acc = acc + __shfl_down(acc, 1);
                        acc = acc + __shfl_down(acc, 2);
                        const uint64_t jsb = m0b >> 6;
                        if (emit && qu == 0 && jsb < A.zb.nsb) A.zb.S[(uint64_t)seg * A.segs.S_seg + (uint64_t)kk * A.zb.nsb + jsb] = acc;
                    }
                    bar();
                } else {
                    if (mg < n_out) {
#pragma unroll
                        for (int k = 0; k < M; k++) A.y[(uint64_t)k * A.y_stride + mg] = make_float2(re_of(y[k]), im_of(y[k]));
                    }
                    for (int q = 0; q < PERIOD - 1; q++) bar();
                }
            }
        }
        while (itc < IT + 1) bar();
    }
#ifdef SNOUT_MF_STAMPS
    if (l == 0 && blockIdx.x < 256) {
        st_acc[7] = __builtin_amdgcn_s_memtime() - st_t0;
        st_acc[6] = st_acc[7] * 100000ull / (__builtin_amdgcn_s_memrealtime() - st_r0 + 1ull);
        for (int k = 0; k < 8; k++) g_sp_stamps[(blockIdx.x * 16 + w) * 8 + k] = st_acc[k];
    }
#endif
}

// =============================================================================================
// Host side
// =============================================================================================
#ifdef SNOUT_MF_STAMPS
extern "C" int snout_debug_sp_stamps(unsigned long long* out, uint32_t n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp_stamps), (size_t)n * 8u, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -4;
}
#endif

uint32_t pfb_spec_tile(uint32_t M) { return M == 40 ? (uint32_t)sp::Layout<40, 16>::T : (uint32_t)sp::Layout<16, 16>::T; }

int pfb_spec_launch(uint32_t M, int mode, int fmt, int waves, uint32_t grid, hipStream_t st, const PfbMfArgs& a, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    // (hipExtLaunchKernelGGL: the stop event is the kernel's own completion signal -- no barrier packet behind the kernel)
#define SNOUT_SP(MM, MODE, WW)                                                                             \
    do {                                                                                                  \
        if (fmt == kFmtSc8) hipExtLaunchKernelGGL((pfb_spec<MM, MODE, kFmtSc8, WW>), dim3(grid), dim3(64 * WW), 0, st, ev_start, ev_stop, 0, a);        \
        else if (fmt == kFmtSc16) hipExtLaunchKernelGGL((pfb_spec<MM, MODE, kFmtSc16, WW>), dim3(grid), dim3(64 * WW), 0, st, ev_start, ev_stop, 0, a); \
        else hipExtLaunchKernelGGL((pfb_spec<MM, MODE, kFmtCf32, WW>), dim3(grid), dim3(64 * WW), 0, st, ev_start, ev_stop, 0, a);                      \
    } while (0)
    if (M == 40 && mode != kSpZb) {
        if (waves == 12) { if (mode == kSpBtle) SNOUT_SP(40, kSpBtle, 12); else SNOUT_SP(40, kSpIq, 12); }
        else             { if (mode == kSpBtle) SNOUT_SP(40, kSpBtle, 16); else SNOUT_SP(40, kSpIq, 16); }
    } else if (M == 16 && mode != kSpBtle) {
        if (mode == kSpZb) SNOUT_SP(16, kSpZb, 16); else SNOUT_SP(16, kSpIq, 16);
    } else {
        return SNOUT_EINVAL;
    }
#undef SNOUT_SP
    SNOUT_HIP(hipGetLastError());
    return 0;
}

}  // namespace snout
