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
// Work split per tile of T output times (one workgroup):
//   1. stage the (T-1)D + MP input samples in LDS (coalesced 8-B loads, read from HBM once; tiles
//      overlap by MP - D samples, served by L2),
//   2. FIR: thread <-> (branch r, output parity e, group g).  Outputs m = e + 2i of one branch are
//      a sliding dot product over the branch stream z[q] = x[r + eD + qM]: 23 LDS reads feed 8
//      outputs x 16 taps, taps live in registers -> FMA-bound, not LDS-bound,
//   3. FFT in two LDS passes (M = M1 M2): thread <-> (m, n2) does the M1-point DFT + twiddle,
//      thread <-> (m, k1) does the M2-point DFT and writes y_k[m] with m fastest (coalesced).
// LDS rows are padded to M + 1 complex so column walks hit distinct banks.
//
// Roofline: at M = 40 the stage needs ~181 flop per input sample (FIR 128 + FFT), i.e. ~23 flop/B:
// just above the f32 vector ridge (157 TFLOP/s / 8 TB/s = 20 flop/B) -> bound by f32 VALU issue.
// The FIR is NOT GEMM-shaped (a per-branch sliding correlation: a Toeplitz operand with 2 useful
// columns), so MFMA does not apply; see DESIGN.md.
#include "common.h"
#include "pfb_tables.inc"

namespace snout {

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

__device__ __forceinline__ void dft5(const cf b[5], cf X[5], const float* __restrict__ tw5)
{
#pragma unroll
    for (int k = 0; k < 5; k++) {
        cf acc = b[0];
#pragma unroll
        for (int n = 1; n < 5; n++) {
            const int j = (n * k) % 5;
            const float wr = tw5[2 * j], wi = tw5[2 * j + 1];
            acc.re = __builtin_fmaf(b[n].re, wr, acc.re);
            acc.re = __builtin_fmaf(-b[n].im, wi, acc.re);
            acc.im = __builtin_fmaf(b[n].re, wi, acc.im);
            acc.im = __builtin_fmaf(b[n].im, wr, acc.im);
        }
        X[k] = acc;
    }
}

template <int M> struct PfbGeom;
template <> struct PfbGeom<40> { static constexpr int T = 64,  M1 = 8, M2 = 5, NT = 320; };
template <> struct PfbGeom<16> { static constexpr int T = 128, M1 = 4, M2 = 4, NT = 256; };

template <int M>
__global__ __launch_bounds__(PfbGeom<M>::NT) void pfb_channelize(
    const float2* __restrict__ x, uint64_t n, uint64_t n_out, const float* __restrict__ proto,
    const float* __restrict__ twM, const float* __restrict__ tw5g, float2* __restrict__ y,
    uint64_t y_stride)
{
    using G = PfbGeom<M>;
    constexpr int T = G::T, M1 = G::M1, M2 = G::M2, NT = G::NT, D = M / 2, P = 16;
    constexpr int SPAN = (T - 1) * D + M * P;      // input samples one tile needs
    constexpr int ROW = M + 1;                     // padded LDS row, complex
    __shared__ float2 xs[SPAN];
    __shared__ float2 us[T * ROW];
    __shared__ float2 bs[T * ROW];
    __shared__ float tw_s[2 * M + 10];

    const int t = threadIdx.x;
    const uint64_t m0 = (uint64_t)blockIdx.x * T;
    const uint64_t in0 = m0 * D;

    // ---- 1. stage input + twiddles
    for (int i = t; i < SPAN; i += NT) {
        const uint64_t g = in0 + (uint64_t)i;
        xs[i] = g < n ? x[g] : make_float2(0.0f, 0.0f);
    }
    for (int i = t; i < 2 * M; i += NT) tw_s[i] = twM[i];
    if (t < 10) tw_s[2 * M + t] = tw5g[t];

    // ---- 2. FIR: thread <-> (r, e, g); outputs m = e + 2 (8g + i), i = 0..7
    const int r = t % M, e = (t / M) & 1, grp = t / (2 * M);
    float h[P];
#pragma unroll
    for (int p = 0; p < P; p++) h[p] = proto[r + p * M];
    __syncthreads();
    {
        const int base = r + e * D + (8 * grp) * M;       // tile-relative index of z[8g]
        float2 w[8 + P - 1];
#pragma unroll
        for (int q = 0; q < 8 + P - 1; q++) w[q] = xs[base + q * M];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float ar = 0.0f, ai = 0.0f;
#pragma unroll
            for (int p = 0; p < P; p++) {
                ar = __builtin_fmaf(h[p], w[i + p].x, ar);
                ai = __builtin_fmaf(h[p], w[i + p].y, ai);
            }
            const int m = e + 2 * (8 * grp + i);
            us[m * ROW + r] = make_float2(ar, ai);
        }
    }
    __syncthreads();

    // ---- 3a. M1-point DFTs over n1 for every (m, n2), then twiddle W_M^{n2 k1}
    for (int it = t; it < T * M2; it += NT) {
        const int m = it % T, n2 = it / T;
        cf a[M1], A[M1];
#pragma unroll
        for (int n1 = 0; n1 < M1; n1++) {
            const float2 v = us[m * ROW + M2 * n1 + n2];
            a[n1] = cf{v.x, v.y};
        }
        if constexpr (M1 == 8) dft8(a, A); else dft4(a, A);
#pragma unroll
        for (int k1 = 0; k1 < M1; k1++) {
            const int j = (n2 * k1) % M;
            const cf v = j ? cmul_tw(A[k1], tw_s[2 * j], tw_s[2 * j + 1]) : A[k1];
            bs[m * ROW + n2 * M1 + k1] = make_float2(v.re, v.im);
        }
    }
    __syncthreads();

    // ---- 3b. M2-point DFTs over n2 for every (m, k1); y_k[m] = (-1)^{km} X[k]
    for (int it = t; it < T * M1; it += NT) {
        const int m = it % T, k1 = it / T;
        cf b[M2], Y[M2];
#pragma unroll
        for (int n2 = 0; n2 < M2; n2++) {
            const float2 v = bs[m * ROW + n2 * M1 + k1];
            b[n2] = cf{v.x, v.y};
        }
        if constexpr (M2 == 5) dft5(b, Y, &tw_s[2 * M]); else dft4(b, Y);
        const uint64_t mg = m0 + (uint64_t)m;
        if (mg < n_out) {
#pragma unroll
            for (int k2 = 0; k2 < M2; k2++) {
                const int k = k1 + M1 * k2;
                cf v = Y[k2];
                if (k & (int)(mg & 1u)) { v.re = -v.re; v.im = -v.im; }
                y[(uint64_t)k * y_stride + mg] = make_float2(v.re, v.im);
            }
        }
    }
}

// =============================================================================================
// Host side
// =============================================================================================
static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

int PfbCtx::init(uint32_t M_)
{
    M = M_;
    if (M != 40 && M != 16) { set_last_error("channelizer supports M = 40 or 16, not %u", M); return SNOUT_EINVAL; }
    const float* proto = M == 40 ? kPfbProto40 : kPfbProto16;
    const float* tw = M == 40 ? kTw40 : kTw16;
    if (int rc = d_proto.ensure(M * 16 * 4)) return rc;
    if (int rc = d_tw.ensure(2 * M * 4)) return rc;
    if (int rc = d_tw5.ensure(10 * 4)) return rc;
    SNOUT_HIP(hipMemcpy(d_proto.p, proto, M * 16 * 4, hipMemcpyHostToDevice));
    SNOUT_HIP(hipMemcpy(d_tw.p, tw, 2 * M * 4, hipMemcpyHostToDevice));
    SNOUT_HIP(hipMemcpy(d_tw5.p, kTw5, 10 * 4, hipMemcpyHostToDevice));
    SNOUT_HIP(hipEventCreate(&ev_k0));
    SNOUT_HIP(hipEventCreate(&ev_k1));
    return 0;
}

void PfbCtx::destroy()
{
    d_proto.release(); d_tw.release(); d_tw5.release(); d_y.release();
    if (ev_k0) { (void)hipEventDestroy(ev_k0); (void)hipEventDestroy(ev_k1); ev_k0 = nullptr; }
}

uint64_t PfbCtx::n_out_for(uint64_t n) const
{
    const uint64_t L = (uint64_t)M * 16u, D = M / 2u;
    return n >= L ? (n - L) / D + 1u : 0u;
}

int PfbCtx::run(const float* d_iq, uint64_t n, hipStream_t st)
{
    n_out = n_out_for(n);
    y_stride = n_out + 64;
    if (int rc = d_y.ensure(y_stride * M * 8u)) return rc;
    if (n_out == 0) return 0;
    SNOUT_HIP(hipEventRecord(ev_k0, st));
    if (M == 40)
        hipLaunchKernelGGL(pfb_channelize<40>, dim3(cdiv(n_out, PfbGeom<40>::T)), dim3(PfbGeom<40>::NT),
                           0, st, (const float2*)d_iq, n, n_out, d_proto.as<float>(), d_tw.as<float>(),
                           d_tw5.as<float>(), d_y.as<float2>(), y_stride);
    else
        hipLaunchKernelGGL(pfb_channelize<16>, dim3(cdiv(n_out, PfbGeom<16>::T)), dim3(PfbGeom<16>::NT),
                           0, st, (const float2*)d_iq, n, n_out, d_proto.as<float>(), d_tw.as<float>(),
                           d_tw5.as<float>(), d_y.as<float2>(), y_stride);
    SNOUT_HIP(hipEventRecord(ev_k1, st));
    SNOUT_HIP(hipGetLastError());
    return 0;
}

}  // namespace snout
