// pfb_mfma.hip — the channelizer with its FIR on the matrix pipe and its FFT in registers (gfx950).
//
// Same arithmetic contract as pfb.hip / oracle/oracle_pfb.c (the channelizer replaces the one-channel hop of
// snout/core/radio.py:415, snout/util/btle.py:62: SURVEY.md §8d cfg #3/#4), other division of labour:
//
//   * ONE 1024-thread workgroup per CU walks a contiguous range of tiles of T = 128 output times; its 16 waves
//     are specialised.  Waves 0-7 ("FIR waves") run the polyphase FIR of tile i on the MATRIX pipe while waves
//     8-15 ("FFT waves") run the FFTs + slicer of tiles i-1 .. i-4 on the VECTOR pipe: the two pipes of a SIMD
//     issue side by side, so the tile time is max(FIR, FFT) instead of their sum (pfb.hip: 0.30 of the f32
//     peak with both on the VALU).  One s_barrier per tile; the FIR-output tile is double-buffered in LDS.
//
//   * FIR as a banded-Toeplitz product on v_mfma_f32_16x16x4_f32.  For one branch r the 128 output times of a
//     tile are four parity streams' worth of sliding dot products over z_{r,e}[q] = x[qM + eD + r]:
//         u_{2(16b+j)+e}[r] = sum_k z_{r,e}[16b + k] * h_r[k - j],   k = 0..31, taps outside [0,16) are 0.
//     D[i][j] = sum_k A[i][k] B[k][j] with rows i = (block b, parity e, re/im) = 16 data streams, columns
//     j = 16 consecutive outputs, B = the Toeplitz matrix of branch r's 16 taps (8 VGPRs, resident): a chain of
//     8 MFMAs yields all 128 outputs of one branch.  Half of B is structural zeros; the matrix pipe is otherwise
//     idle, and v_mfma_f32 is an exact fmaf chain in ascending k (tools/mfma_probe.hip), so the result equals
//     the oracle's 16-term chain bit for bit (a zero tap adds +-0; the oracle restates the chain with the zero
//     terms where that can matter: non-finite samples, results that are zero).
//     The A operand is one ds_read_b32 per lane and MFMA; the span is stored XOR-swizzled by its 16-position
//     block so that the 32 lanes of a read hit 32 banks (unswizzled: blocks 5 120 B apart, 4-way conflicts).
//
//   * FFT: a thread owns one output time: 20 ds_read_b128 of its row, the 8x5 (4x4) FFT of oracle_pfb.c entirely
//     in registers (680 VALU for M = 40, no LDS round trip, no barrier inside), then the epilogue on its 40
//     results.  BTLE: lane = 16 (m mod 4) + (m / 4 mod 16), so y[m+4] is the next lane of the DPP row and the
//     64 lanes' hard bits of one channel are four 16-symbol pieces of its plane words straight out of v_cmp.
//     A 64-time block is spread over four tile times (three barriers inside the straight-line code), so the
//     eight FFT waves always hold eight blocks in flight: the VALU work never sits on the critical path.
#include "common.h"
#include "iq_fmt.h"
#include <type_traits>
#include "pfb_tables.inc"

namespace snout {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
// v_writelane_b32 as the compiler's own instruction (this clang has no __builtin_amdgcn_writelane): as inline
// asm the hazard recogniser does not see it, and on gfx950 a VALU read of an SGPR needs two wait states behind
// the v_cmp that wrote it -- the asm form read stale masks wherever the scheduler put the two back to back.
extern "C" __device__ int __llvm_amdgcn_writelane(int, int, int) __asm("llvm.amdgcn.writelane");

#ifdef SNOUT_MF_STAMPS
// Diagnostic build only (tools/mf_stamps.py): shader cycles each wave spends in its phases, summed over the tiles
// of a workgroup: [block][wave][slot]; slot 7 = the wave's whole run, slot 6 = shader clock in kHz.
__device__ unsigned long long g_mf_stamps[256 * 16 * 8];
#define MF_STAMP(k)                                                              \
    do {                                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();           \
        st_acc[k] += now_ - st_last; st_last = now_;                             \
    } while (0)
#else
#define MF_STAMP(k) do { } while (0)
#endif

namespace mf {

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

// Workgroup barrier that orders LDS traffic only (vector-memory loads and stores stay in flight).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int M> struct Geom;
template <> struct Geom<40> { static constexpr int T = 128, M1 = 8, M2 = 5, ROW = 41, NB = 5; };
template <> struct Geom<16> { static constexpr int T = 128, M1 = 4, M2 = 4, ROW = 17, NB = 2; };
constexpr int kFirWaves = 8, kFftWaves = 8, kThreads = 64 * (kFirWaves + kFftWaves);

// XOR swizzle of the staged span: sample s (tile-relative) of position row P = s / M lives at s ^ swz(P).
// The lanes of one A-operand read differ in (k mod 2, parity e, re/im, block b); unswizzled the four blocks
// (16 positions = 16 M samples = a multiple of 128 bytes apart) fall on the same banks.
template <int M> __device__ __forceinline__ uint32_t swz_of(uint32_t P)
{
    if constexpr (M == 40) return (P >> 4) & 3u;                       // sample bits 0-1 = bank bits 1-2
    else return (((P >> 4) & 3u) << 1) | (P & 1u);                     // M = 16: also k mod 2 (16 samples = 128 B)
}

}  // namespace mf

// Output modes of the kernel
constexpr int kMfIq = 0, kMfBtle = 1;

// IMPL: how the front-end waves 0-7 compute the FIR.
//   kFirMfma: all eight on the matrix pipe (banded-Toeplitz chains), staging on the side
//   kFirValu: waves 0 .. M/8-1 with packed FMAs (the sliding dot product of pfb.hip: thread <-> branch, parity,
//             16 outputs), the other front-end waves only stage the input, two tiles ahead
constexpr int kFirMfma = 0, kFirValu = 1;

template <int M, int MODE, int FMT, int IMPL>
__global__ __launch_bounds__(mf::kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void pfb_mfma(const PfbMfArgs A)
{
    using namespace mf;
    using G = Geom<M>;
    constexpr int T = G::T, M1 = G::M1, M2 = G::M2, ROW = G::ROW, NB = G::NB, D = M / 2, P = 16;
    constexpr int SPAN = (T - 1) * D + M * P, NEW = T * D, OV = SPAN - NEW;
    constexpr int XS = SPAN + 2;                        // + one always-zero sample (the k = 31 operand)
    constexpr int NFIR = 64 * kFirWaves;                // threads that stage
    static_assert(M % kFirWaves == 0 && NB == M / kFirWaves, "branches per FIR wave");
    static_assert(SPAN % 8 == 0 || M == 40, "swizzle groups stay inside the span");
    static_assert(NEW % (4 * 16 * M) == 0, "the overlap keeps its swizzle when it moves to the front");
    static_assert((XS * 8) % 16 == 0 && OV % 2 == 0 && NEW % 2 == 0, "16-byte staging");
    constexpr bool BT = MODE == kMfBtle;
    constexpr bool PHASE_MAJOR = M == 40;               // FFT lane <-> output time (see the header)

    __shared__ float2 xs[2][XS];
    __shared__ float2 us[2][T * ROW];
    __shared__ float2 carry[BT ? kFftWaves : 1][2][2][BT ? (M / 2) * 4 : 1];   // [wave][block parity][first / last four times][channel of the wave][phase]

    const uint32_t seg = blockIdx.x / A.segs.wgs_per_seg, bid = blockIdx.x - seg * A.segs.wgs_per_seg;
    const void* __restrict__ x = A.segs.x[seg];
    uint16_t* planes16 = A.planes16 ? A.planes16 + (uint64_t)seg * A.segs.planes_seg : nullptr;
    const uint64_t n = A.n, n_out = A.n_out;
    const uint32_t n_tiles = A.n_tiles;

    const uint32_t t_begin = bid * A.tiles_per_wg;
    uint32_t t_end = t_begin + A.tiles_per_wg;
    if (t_end > n_tiles) t_end = n_tiles;
    if (t_begin >= t_end) return;
    // BTLE: the first four output times of the tile behind the range complete the range's last symbols
    const uint32_t t_stop = (BT && t_end < n_tiles) ? t_end + 1u : t_end;
    const int NTL = (int)(t_stop - t_begin);            // tiles this workgroup computes
    const int IT = NTL + 3;                              // barriers after the first one (pipeline drain included)

    const int t = threadIdx.x, w = t >> 6, l = t & 63;
#ifdef SNOUT_MF_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_last, st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    if (IMPL == kFirValu && w < kFirWaves) {
        using Raw = typename IqRaw<FMT>::pair;
        constexpr int NFW = M / 8;                                  // FIR waves: (branch, parity, 4 groups of 16 outputs)
        static_assert(NFW * 64 == 2 * M * 4 && T == 128, "FIR thread map");
        if (w < NFW) {
            // =================================================================================
            // FIR waves (vector pipe): thread <-> (branch r, output parity e, group grp): the 16 outputs
            // m = e + 2 (16 grp + i) of one branch are a sliding dot product over z[q] = x[r + e D + q M]
            // =================================================================================
            const int tf = w * 64 + l, r = tf % M, e = (tf / M) & 1, grp = tf / (2 * M);
            v2f hp[P / 2];
#pragma unroll
            for (int p = 0; p < P / 2; p++) hp[p] = v2f{A.proto[r + (2 * p) * M], A.proto[r + (2 * p + 1) * M]};
            const uint32_t rd = (uint32_t)((r + e * D + 16 * grp * M) * 8);                       // window start, bytes
            // row of output i: 64 (grp / 2) + 16 (e + 2 (i & 1)) + 8 (grp & 1) + i / 2  (phase-major rows, see the FFT waves)
            const uint32_t wr = PHASE_MAJOR ? (uint32_t)(((64 * (grp >> 1) + 16 * e + 8 * (grp & 1)) * ROW + r) * 8)
                                            : (uint32_t)(((e + 32 * grp) * ROW + r) * 8);
            auto fir_tile = [&](auto bufc) {
                constexpr int BUF = decltype(bufc)::value;
                const uint32_t a0 = (uint32_t)(uintptr_t)&xs[BUF][0] + rd;
                char* uo = reinterpret_cast<char*>(&us[BUF][0]) + wr;
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
                    for (int j = 0; j < 4; j++) {
                        const int i = i0 + j;
                        const int rowoff = PHASE_MAJOR ? (32 * (i & 1) + (i >> 1)) : 2 * i;
                        *reinterpret_cast<float2*>(uo + rowoff * ROW * 8) = make_float2(acc[j].x, acc[j].y);
                    }
                }
            };
            lds_barrier();                                           // the staging waves' prologue
            for (int it = 0; it < IT; it += 2) {
#pragma unroll
                for (int hb = 0; hb < 2; hb++) {
                    const int i2 = it + hb;
                    if (i2 < IT) {
                        if (i2 < NTL) {
                            MF_STAMP(0);
                            if (hb == 0) fir_tile(std::integral_constant<int, 0>{});
                            else         fir_tile(std::integral_constant<int, 1>{});
                            MF_STAMP(1);
                        }
                        lds_barrier();
                        MF_STAMP(2);
                    }
                }
            }
        } else {
            // =================================================================================
            // Staging waves: the input span of tile i + 1 is in LDS before the barrier that ends tile i; its new
            // samples were requested two tiles earlier into one of two register sets (a 2^24-sample-per-CU-second
            // stream: the latency of a load under that traffic is about one tile time)
            // =================================================================================
            constexpr int NST = 64 * (kFirWaves - NFW);
            const int ts = (w - NFW) * 64 + l;
            auto load_pair = [&](uint64_t g) -> Raw {                  // samples g, g+1 (g even), zero past n
                if (g + 1 < n) return iq_pair_raw<FMT>(x, g);
                Raw v = Raw{};
                if (g < n) v = iq_single_raw<FMT>(x, g);
                return v;
            };
            constexpr int NPRE = (NEW / 2 + NST - 1) / NST, NOV = (OV / 2 + NST - 1) / NST;
            Raw pre[2][NPRE];
            auto fetch = [&](Raw (&set)[NPRE], uint32_t tile) {
                const uint64_t in1 = (uint64_t)tile * NEW + OV;
                if (in1 + NEW <= n) {
#pragma unroll
                    for (int k = 0; k < NPRE; k++)
                        if (k * NST + NST <= NEW / 2 || ts + k * NST < NEW / 2) set[k] = iq_pair_raw<FMT>(x, in1 + 2ull * (uint64_t)(ts + k * NST));
                } else {
#pragma unroll
                    for (int k = 0; k < NPRE; k++)
                        if (k * NST + NST <= NEW / 2 || ts + k * NST < NEW / 2) set[k] = load_pair(in1 + 2ull * (uint64_t)(ts + k * NST));
                }
            };
            auto overlap = [&](int nb) {                                // xs[nb ^ 1][NEW ..] -> xs[nb][0 ..]
                float4* dst = reinterpret_cast<float4*>(&xs[nb][0]);
                const float4* src = reinterpret_cast<const float4*>(&xs[nb ^ 1][0]);
#pragma unroll
                for (int k = 0; k < NOV; k++)
                    if (ts + k * NST < OV / 2) dst[ts + k * NST] = src[ts + k * NST + NEW / 2];
            };
            auto stage = [&](Raw (&set)[NPRE], int nb) {
                float4* dst = reinterpret_cast<float4*>(&xs[nb][0]);
#pragma unroll
                for (int k = 0; k < NPRE; k++)
                    if (k * NST + NST <= NEW / 2 || ts + k * NST < NEW / 2) dst[OV / 2 + ts + k * NST] = iq_pair_cvt<FMT>(set[k]);
            };
            {
                float4* xb = reinterpret_cast<float4*>(&xs[0][0]);
                const uint64_t in0 = (uint64_t)t_begin * NEW;
                for (uint32_t q = (uint32_t)ts; q < (uint32_t)(SPAN / 2); q += NST) xb[q] = iq_pair_cvt<FMT>(load_pair(in0 + 2ull * q));
                if (1 < NTL) fetch(pre[1], t_begin + 1u);
            }
            lds_barrier();
            for (int it = 0; it < IT; it += 2) {
#pragma unroll
                for (int hb = 0; hb < 2; hb++) {
                    const int i2 = it + hb;
                    if (i2 < IT) {
                        if (i2 + 2 < NTL) fetch(pre[hb], t_begin + (uint32_t)i2 + 2u);
                        if (i2 + 1 < NTL) { overlap(hb ^ 1); stage(pre[hb ^ 1], hb ^ 1); }
                        MF_STAMP(0);
                        lds_barrier();
                        MF_STAMP(2);
                    }
                }
            }
        }
    } else if (w < kFirWaves) {
        // =====================================================================================
        // FIR waves: stage the input span, run the FIR on the matrix pipe
        // =====================================================================================
        const int jl = l & 15, kq = l >> 4;
        const int blk = jl >> 2, e = (jl >> 1) & 1, c = jl & 1;
        constexpr uint32_t MASK = M == 40 ? 3u : 7u;
        float tap[NB][8];
        uint32_t fa0[NB], fa3[NB], fa1[NB], fa7[NB];     // float index of the lane's k-step 0 / 3 / 4.. / 7 operand
#pragma unroll
        for (int bi = 0; bi < NB; bi++) {
            const int r = w * NB + bi;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int p = 4 * kk + kq - jl;
                tap[bi][kk] = (p >= 0 && p < P) ? A.proto[r + M * p] : 0.0f;
            }
            const uint32_t s0 = (uint32_t)((16 * blk + kq) * M + e * D + r);
            const uint32_t cy = (e * D + r) >= M ? 1u : 0u;
            const uint32_t low = s0 & MASK, hi = s0 & ~MASK, podd = (uint32_t)(kq + cy) & 1u;
            const uint32_t f0 = 2u * (hi | (low ^ swz_of<M>((uint32_t)(16 * blk) + podd))) + (uint32_t)c;
            const uint32_t f1 = 2u * (hi | (low ^ swz_of<M>((uint32_t)(16 * (blk + 1)) + podd))) + (uint32_t)c;
            fa0[bi] = f0;
            fa1[bi] = f1;
            fa3[bi] = (kq == 3 && cy) ? f1 : f0;         // 4 kk + kq + carry reaches 16 at kk = 3 for these lanes
            fa7[bi] = (kq == 3) ? (uint32_t)(2 * SPAN) : f1 + (uint32_t)(8 * M * 7);    // k = 31: the zero sample
        }
        // where this lane's four results of a chain go: rows m = 2 idx + e, idx = 16 (l / 16) + (l mod 16)
        uint32_t dw[2];
#pragma unroll
        for (int ee = 0; ee < 2; ee++) {
            const uint32_t m = 2u * (uint32_t)(16 * kq + jl) + (uint32_t)ee;
            const uint32_t rowpos = PHASE_MAJOR ? ((m & 64u) + 16u * (m & 3u) + ((m & 63u) >> 2)) : m;
            dw[ee] = rowpos * ROW;
        }

        // ---- staging roles
        const int ts = w * 64 + l;
        using Raw = typename IqRaw<FMT>::pair;
        auto load_pair = [&](uint64_t g) -> Raw {                  // samples g, g+1 (g even), zero past n
            if (g + 1 < n) return iq_pair_raw<FMT>(x, g);
            Raw v = Raw{};
            if (g < n) v = iq_single_raw<FMT>(x, g);
            return v;
        };
        auto pair_slot = [&](uint32_t q, uint32_t& swap) -> uint32_t {   // tile-relative pair q -> float4 slot
            const uint32_t Pq = (2u * q) / (uint32_t)M, sz = swz_of<M>(Pq);
            swap = sz & 1u;
            return q ^ (sz >> 1);
        };
        auto put_pair = [&](float4* xb, uint32_t slot, uint32_t swap, float4 v) {
            xb[slot] = swap ? make_float4(v.z, v.w, v.x, v.y) : v;
        };
        constexpr int NPRE = (NEW / 2 + NFIR - 1) / NFIR;          // new pairs per thread and tile (the last round partial)
        uint32_t pslot[NPRE], pswap[NPRE];
#pragma unroll
        for (int k = 0; k < NPRE; k++) pslot[k] = pair_slot((uint32_t)(OV / 2 + ts + k * NFIR), pswap[k]);
        uint32_t oswap;
        const uint32_t oslot = pair_slot((uint32_t)ts, oswap);     // overlap pair ts (same swizzle NEW samples further)
        (void)oswap;
        Raw pre[NPRE];
        auto fetch = [&](uint32_t tile) {                          // request tile's new samples
            const uint64_t in1 = (uint64_t)tile * NEW + OV;
            if (in1 + NEW <= n) {
#pragma unroll
                for (int k = 0; k < NPRE; k++)
                    if (k * NFIR + NFIR <= NEW / 2 || ts + k * NFIR < NEW / 2) pre[k] = iq_pair_raw<FMT>(x, in1 + 2ull * (uint64_t)(ts + k * NFIR));
            } else {
#pragma unroll
                for (int k = 0; k < NPRE; k++)
                    if (k * NFIR + NFIR <= NEW / 2 || ts + k * NFIR < NEW / 2) pre[k] = load_pair(in1 + 2ull * (uint64_t)(ts + k * NFIR));
            }
        };
        auto stage = [&](int nb) {                                  // registers -> xs[nb], overlap from xs[nb ^ 1]
            float4* dst = reinterpret_cast<float4*>(&xs[nb][0]);
            const float4* src = reinterpret_cast<const float4*>(&xs[nb ^ 1][0]);
#pragma unroll
            for (int k = 0; k < NPRE; k++)
                if (k * NFIR + NFIR <= NEW / 2 || ts + k * NFIR < NEW / 2) put_pair(dst, pslot[k], pswap[k], iq_pair_cvt<FMT>(pre[k]));
            if (ts < OV / 2) dst[oslot] = src[oslot + NEW / 2];
        };
        // ---- prologue: the whole span of the first tile, the zero samples, the second tile's request
        {
            float4* xb = reinterpret_cast<float4*>(&xs[0][0]);
            const uint64_t in0 = (uint64_t)t_begin * NEW;
            for (uint32_t q = (uint32_t)ts; q < (uint32_t)(SPAN / 2); q += NFIR) {
                uint32_t sw;
                const uint32_t slot = pair_slot(q, sw);
                put_pair(xb, slot, sw, iq_pair_cvt<FMT>(load_pair(in0 + 2ull * q)));
            }
            if (ts < 2) xs[ts][SPAN] = make_float2(0.0f, 0.0f);
            if (1 < NTL) fetch(t_begin + 1u);
        }
        lds_barrier();

        // One chain = the 8 MFMAs of one branch.  The operand reads of chain bi + 1 are issued BEFORE the MFMAs of
        // chain bi and waited for with a counted s_waitcnt (LDS returns in order: the 8 reads of the next chain and
        // the 2 result stores of this one may stay outstanding), so the matrix pipe never waits for an LDS round
        // trip.  (The compiler's own schedule of the plain C++ form put each read right in front of its MFMA:
        // three full lgkmcnt(0) stalls per chain.)  Reads as inline asm: offsets fold into the instruction.
        auto fir_tile = [&](auto bufc) {
            constexpr int BUF = decltype(bufc)::value;
            const uint32_t xb = (uint32_t)(uintptr_t)&xs[BUF][0];
            float2* uo = &us[BUF][0];
            auto reads = [&](float (&a)[8], int bi) {
                const uint32_t b0 = xb + 4u * fa0[bi], b3 = xb + 4u * fa3[bi], b1 = xb + 4u * fa1[bi], b7 = xb + 4u * fa7[bi];
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[0]) : "v"(b0), "n"(32 * M * 0));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[1]) : "v"(b0), "n"(32 * M * 1));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[2]) : "v"(b0), "n"(32 * M * 2));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[3]) : "v"(b3), "n"(32 * M * 3));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[4]) : "v"(b1), "n"(32 * M * 4));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[5]) : "v"(b1), "n"(32 * M * 5));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[6]) : "v"(b1), "n"(32 * M * 6));
                asm volatile("ds_read_b32 %0, %1" : "=v"(a[7]) : "v"(b7));
            };
            float a[2][8];
            reads(a[0], 0);
#pragma unroll
            for (int bi = 0; bi < NB; bi++) {
                float (&ac)[8] = a[bi & 1];
                if (bi + 1 < NB) {
                    reads(a[(bi + 1) & 1], bi + 1);
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ac[0]), "+v"(ac[1]), "+v"(ac[2]), "+v"(ac[3]), "+v"(ac[4]), "+v"(ac[5]), "+v"(ac[6]), "+v"(ac[7]) :: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ac[0]), "+v"(ac[1]), "+v"(ac[2]), "+v"(ac[3]), "+v"(ac[4]), "+v"(ac[5]), "+v"(ac[6]), "+v"(ac[7]) :: "memory");
                }
                v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int kk = 0; kk < 8; kk++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[kk], tap[bi][kk], acc, 0, 0, 0);
                const int r = w * NB + bi;
                uo[dw[0] + r] = make_float2(acc[0], acc[1]);
                uo[dw[1] + r] = make_float2(acc[2], acc[3]);
            }
        };

        for (int it = 0; it < IT; it += 2) {
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
                const int i2 = it + hb;
                if (i2 < IT) {
                    if (i2 < NTL) {
                        if (i2 + 1 < NTL) stage(hb ^ 1);
                        if (i2 + 2 < NTL) fetch(t_begin + (uint32_t)i2 + 2u);
                        MF_STAMP(0);
                        if (hb == 0) fir_tile(std::integral_constant<int, 0>{});
                        else         fir_tile(std::integral_constant<int, 1>{});
                        MF_STAMP(1);
                    }
                    lds_barrier();
                    MF_STAMP(2);
                }
            }
        }
    } else {
        // =====================================================================================
        // FFT waves: a thread owns one output time of a 64-time block and half of its channels
        // =====================================================================================
        // A block's 40-point FFTs hold 40 complex values per thread at their mid-point: 80 registers, which the
        // epilogue state on top does not fit into the 128 a 16-wave workgroup leaves per thread.  So TWO waves
        // share a block: wave kh = 0 takes the outputs k1 < M1/2 of the first stage, wave kh = 1 the others
        // (E[k] + T[k] / E[k] - T[k] of the same even / odd halves, which both compute: 48 instead of 56
        // operations per 8-point DFT), i.e. the channels k = k1 + M1 k2 of its k1 half.  Wave f: tile parity
        // f / 4, time half (f / 2) mod 2, channel half f mod 2; a block takes two tile times, Q1 | barrier | Q2.
        const int f = w - kFirWaves, j0 = f >> 2, half = (f >> 1) & 1, kh = f & 1;
        constexpr int H1 = M1 / 2, NCH = H1 * M2;                     // first-stage outputs / channels per wave
        const float* const tw = M == 40 ? kTw40 : kTw16;
        const float c5_1 = kTw5[2], c5_2 = kTw5[4], s5_1 = -kTw5[3], s5_2 = -kTw5[5];
        // output time of this lane within its block
        const uint32_t mloc = PHASE_MAJOR ? (4u * (uint32_t)(l & 15) + (uint32_t)(l >> 4)) : (uint32_t)l;
        int itc = 0;
#ifdef SNOUT_MF_STAMPS
        auto bar = [&]() { MF_STAMP(3); lds_barrier(); itc++; MF_STAMP(2); };
#else
        auto bar = [&]() { lds_barrier(); itc++; };
#endif
        bar();                                                       // the FIR waves' prologue
        for (int i = 0; i <= j0; i++) bar();

        // BTLE: state of the block whose last symbols wait for the next block's first output times.
        // Lane c < NCH holds the 64 hard bits of the wave's channel c = k1l + H1 k2 (k = k1l + H1 kh + M1 k2).
        uint32_t pm_lo = 0, pm_hi = 0;
        uint64_t pend_m0 = 0;
        int pend_par = 0;
        bool pending = false;
        auto finalize = [&]() {
            if constexpr (BT) {
                if (pending && l < NCH) {
                    const uint64_t nbits = n_out >= 4 ? n_out - 4 : 0;           // bits exist for m < n_out - 4
                    // the block's own last four output times and the first four of the block behind it: the wave
                    // two further (same tile, second half) or the first-half wave of the other tile parity
                    const int fs = half == 0 ? f + 2 : ((f & 3) - 2 + 4 * (1 - (f >> 2)));
                    const int spar = half == 0 ? pend_par : (j0 == 0 ? pend_par : pend_par ^ 1);
                    const float4* la = reinterpret_cast<const float4*>(&carry[f][pend_par][1][l * 4]);
                    const float4* fi = reinterpret_cast<const float4*>(&carry[fs][spar][0][l * 4]);
                    const float4 l01 = la[0], l23 = la[1], f01 = fi[0], f23 = fi[1];
                    const uint32_t b0 = (l01.x * f01.y) > (f01.x * l01.y) ? 1u : 0u;
                    const uint32_t b1 = (l01.z * f01.w) > (f01.z * l01.w) ? 1u : 0u;
                    const uint32_t b2 = (l23.x * f23.y) > (f23.x * l23.y) ? 1u : 0u;
                    const uint32_t b3 = (l23.z * f23.w) > (f23.z * l23.w) ? 1u : 0u;
                    const uint32_t lo = (pm_lo & 0x7FFF7FFFu) | (b0 << 15) | (b1 << 31);
                    const uint32_t hi = (pm_hi & 0x7FFF7FFFu) | (b2 << 15) | (b3 << 31);
                    // symbols whose sample exists: m0 + 4 sy + j < nbits, a prefix of the 16 per phase
                    const uint32_t left = nbits > pend_m0 ? (uint32_t)(nbits - pend_m0 < 64u ? nbits - pend_m0 : 64u) : 0u;
                    const int k = (l % H1) + H1 * kh + M1 * (l / H1);
                    uint16_t* dst = planes16 + ((uint64_t)k * A.plane_stride + (pend_m0 >> 8) * 4u) * 4u + (uint32_t)((pend_m0 & 255u) >> 6);
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

        auto block = [&](auto khc, int j) {
            constexpr int KH = decltype(khc)::value;
            const uint32_t tile = t_begin + (uint32_t)j;
            const uint64_t m0b = (uint64_t)tile * T + 64u * (uint32_t)half;
            const bool emit = tile < t_end;                          // the tile behind the range only supplies first4
            const bool need = emit || half == 0;
            const int par = (j >> 1) & 1;                            // which of the wave's two carry slots
            // ---- Q1: M1-point DFTs over n1 (this wave's half of the outputs) and the twiddles W_M^{n2 k1}
            cf Bv[M2][H1];
            if (need) {
                const float2* rowp = &us[j & 1][(64 * half + l) * ROW];
#pragma unroll
                for (int n2 = 0; n2 < M2; n2++) {
                    cf a[M1], X[H1];
#pragma unroll
                    for (int n1 = 0; n1 < M1; n1++) { const float2 v = rowp[M2 * n1 + n2]; a[n1] = cf{v.x, v.y}; }
                    if constexpr (M1 == 8) {
                        const float c = 0.70710678118654752440f;
                        const cf e[4] = {a[0], a[2], a[4], a[6]}, o[4] = {a[1], a[3], a[5], a[7]};
                        cf E[4], O[4], Tt[4];
                        dft4(e, E);
                        dft4(o, O);
                        Tt[0] = O[0];
                        Tt[1] = cf{(O[1].re + O[1].im) * c, (O[1].im - O[1].re) * c};
                        Tt[2] = cf{O[2].im, -O[2].re};
                        Tt[3] = cf{(O[3].im - O[3].re) * c, -((O[3].re + O[3].im) * c)};
#pragma unroll
                        for (int k = 0; k < 4; k++) X[k] = KH ? csub(E[k], Tt[k]) : cadd(E[k], Tt[k]);
                    } else {
                        // 4-point DFT: outputs 0, 1 (KH = 0) or 2, 3 (KH = 1)
                        const cf s0 = cadd(a[0], a[2]), s1 = csub(a[0], a[2]);
                        const cf s2 = cadd(a[1], a[3]), s3 = csub(a[1], a[3]);
                        if (KH == 0) { X[0] = cadd(s0, s2); X[1] = cf{s1.re + s3.im, s1.im - s3.re}; }
                        else         { X[0] = csub(s0, s2); X[1] = cf{s1.re - s3.im, s1.im + s3.re}; }
                    }
#pragma unroll
                    for (int k1l = 0; k1l < H1; k1l++) {
                        const int jj = (n2 * (k1l + H1 * KH)) % M;
                        Bv[n2][k1l] = jj == 0 ? X[k1l] : cmul_tw(X[k1l], tw[2 * jj], tw[2 * jj + 1]);   // literals
                    }
                }
            }
            bar();
            // ---- Q2: M2-point DFTs over n2 per k1, epilogue per channel k = k1 + M1 k2
            finalize();                       // the block before this one: its successor's first output times are there now
            if (need) {
                uint32_t m_lo = 0, m_hi = 0;
#pragma unroll
                for (int k1l = 0; k1l < H1; k1l++) {
                    const int k1 = k1l + H1 * KH;
                    cf b[M2], Y[M2];
#pragma unroll
                    for (int n2 = 0; n2 < M2; n2++) b[n2] = Bv[n2][k1l];
                    if constexpr (M2 == 5) dft5(b, Y, c5_1, c5_2, s5_1, s5_2); else dft4(b, Y);
                    if constexpr (BT) {
                        // bit[m] = (I[m] Q[m+4]) > (I[m+4] Q[m]); m + 4 is the next lane of the 16-lane row.  The
                        // factor (-1)^{km} is the same for m and m + 4 and cancels in both products.
#pragma unroll
                        for (int k2 = 0; k2 < M2; k2++) {
                            const float qn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Y[k2].im), 0x101, 0xF, 0xF, true));
                            const float in = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Y[k2].re), 0x101, 0xF, 0xF, true));
                            const uint64_t mk = __builtin_amdgcn_ballot_w64((Y[k2].re * qn) > (in * Y[k2].im));
                            m_lo = (uint32_t)__llvm_amdgcn_writelane((int)(uint32_t)mk, k1l + H1 * k2, (int)m_lo);
                            m_hi = (uint32_t)__llvm_amdgcn_writelane((int)(uint32_t)(mk >> 32), k1l + H1 * k2, (int)m_hi);
                        }
                        const int li = l & 15;
                        if (li == 0 || li == 15) {                   // first / last four output times of the block
#pragma unroll
                            for (int k2 = 0; k2 < M2; k2++)
                                carry[f][par][li == 15][(k1l + H1 * k2) * 4 + (l >> 4)] = make_float2(Y[k2].re, Y[k2].im);
                        }
                    } else {
                        const uint64_t mg = m0b + mloc;
                        if (mg < n_out) {
#pragma unroll
                            for (int k2 = 0; k2 < M2; k2++) {
                                const int k = k1 + M1 * k2;
                                cf v = Y[k2];
                                if ((k & 1) && (mg & 1)) { v.re = -v.re; v.im = -v.im; }
                                A.y[(uint64_t)k * A.y_stride + mg] = make_float2(v.re, v.im);
                            }
                        }
                    }
                }
                if constexpr (BT) {
                    pm_lo = m_lo; pm_hi = m_hi; pend_m0 = m0b; pend_par = par; pending = emit;
                }
            }
            bar();
        };
        for (int j = j0; j < NTL; j += 2) {
            if (kh == 0) block(std::integral_constant<int, 0>{}, j);
            else         block(std::integral_constant<int, 1>{}, j);
        }
        // ---- drain: one more barrier, then the last block's pending symbols; then keep step with the FIR waves
        if (itc < IT + 1) bar();
        finalize();
        while (itc < IT + 1) bar();
    }
#ifdef SNOUT_MF_STAMPS
    if (l == 0 && blockIdx.x < 256) {
        st_acc[7] = __builtin_amdgcn_s_memtime() - st_t0;
        st_acc[6] = st_acc[7] * 100000ull / (__builtin_amdgcn_s_memrealtime() - st_r0 + 1ull);
        for (int k = 0; k < 8; k++) g_mf_stamps[(blockIdx.x * 16 + w) * 8 + k] = st_acc[k];
    }
#endif
}

// =============================================================================================
// Host side
// =============================================================================================
#ifdef SNOUT_MF_STAMPS
extern "C" int snout_debug_mf_stamps(unsigned long long* out, uint32_t n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mf_stamps), (size_t)n * 8u, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -4;
}
#endif
int pfb_mfma_launch(uint32_t M, bool btle, int fmt, int impl, uint32_t grid, hipStream_t st, const PfbMfArgs& a)
{
#define SNOUT_MF(MM, MODE, IM)                                                                             \
    do {                                                                                                  \
        if (fmt == kFmtSc8) hipLaunchKernelGGL((pfb_mfma<MM, MODE, kFmtSc8, IM>), dim3(grid), dim3(mf::kThreads), 0, st, a);        \
        else if (fmt == kFmtSc16) hipLaunchKernelGGL((pfb_mfma<MM, MODE, kFmtSc16, IM>), dim3(grid), dim3(mf::kThreads), 0, st, a); \
        else hipLaunchKernelGGL((pfb_mfma<MM, MODE, kFmtCf32, IM>), dim3(grid), dim3(mf::kThreads), 0, st, a);                      \
    } while (0)
    if (M != 40) return SNOUT_EINVAL;
    if (impl == kFirMfma) { if (btle) SNOUT_MF(40, kMfBtle, kFirMfma); else SNOUT_MF(40, kMfIq, kFirMfma); }
    else                  { if (btle) SNOUT_MF(40, kMfBtle, kFirValu); else SNOUT_MF(40, kMfIq, kFirValu); }
#undef SNOUT_MF
    SNOUT_HIP(hipGetLastError());
    return 0;
}

}  // namespace snout
