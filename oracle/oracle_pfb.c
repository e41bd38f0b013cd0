/*
 * oracle_pfb.c — CPU statement of the polyphase FFT channelizer.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference has no channelizer: it observes one channel at a time and hops sequentially
 * (snout/core/radio.py:415, snout/util/btle.py:62).  The channelizer is the north-star addition
 * that lets one wideband capture feed every channel's demodulator at once (SURVEY §2.1, §8d
 * cfg #3/#4), so there is no upstream arithmetic to follow; this file IS the specification, and
 * the HIP kernel follows the same operation order so outputs are bit-identical f32.
 *
 * Definition (M channels, decimation D = M/2, P = 16 taps per branch, L = M P, prototype h):
 *   u_m[r] = sum_{p=0}^{P-1} h[r + pM] x[mD + r + pM]         fmaf chain over ascending p
 *            (oracle_pfb_block_order: the same sum in the term order of the experimental matrix-pipe kernel)
 *   X_m[k] = sum_r u_m[r] e^{-2 pi i k r / M}                  two-factor FFT specified below
 *   y_k[m] = (-1)^{k m} X_m[k]                                 (D = M/2 phase rotation)
 * for m = 0 .. n_out-1, n_out = (n - L)/D + 1.  y_k is channel k (centre k fs/M) at 2 fs/M.
 *
 * FFT, M = 16 (M1 = M2 = 4, Cooley-Tukey): n = M2 n1 + n2, k = k1 + M1 k2;
 *   A[n2][k1] = DFT_M1 over n1 (radix-2 DIT butterflies, exact +-1/+-i, sqrt(1/2) as one f32)
 *   B[n2][k1] = A[n2][k1] W_M^{n2 k1}        re = fmaf(a,c,-(b d)), im = fmaf(a,d,b c)
 *   X[k1 + M1 k2] = DFT_M2 over n2           butterflies
 * FFT, M = 40 (8 x 5, coprime: Good-Thomas prime-factor mapping, no twiddle factors -- since round 4):
 *   n = (5 n1 + 8 n2) mod 40,  k = (25 k1 + 16 k2) mod 40  (k = k1 mod 8, k = k2 mod 5), so n k = 5 n1 k1 + 8 n2 k2 mod 40 and
 *   A[n2][k1] = DFT_8 over n1 of u[(5 n1 + 8 n2) mod 40]    (dft8 below)
 *   X[(25 k1 + 16 k2) mod 40] = DFT_5 over n2 of A[n2][k1]  (real-factor form, dft5 below)
 *   (Rounds 1-3 specified the Cooley-Tukey form with the 28 twiddle products W_40^{n2 k1}; the two are the same DFT and differ
 *   in the last bits.  The Cooley-Tukey form stays available -- oracle_pfb_legacy_fft() -- as the specification of the two
 *   kernels kept as A/B partners in libsnout_rx_ab.so, pfb.hip and pfb_mfma.hip.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "pfb_tables.inc"

typedef struct { float re, im; } cf;

static inline cf cadd(cf a, cf b) { cf r = {a.re + b.re, a.im + b.im}; return r; }
static inline cf csub(cf a, cf b) { cf r = {a.re - b.re, a.im - b.im}; return r; }
static inline cf cmul_tw(cf a, float c, float d)
{
    cf r;
    r.re = fmaf(a.re, c, -(a.im * d));
    r.im = fmaf(a.re, d, a.im * c);
    return r;
}

static void dft4(const cf b[4], cf X[4])
{
    const cf s0 = cadd(b[0], b[2]), s1 = csub(b[0], b[2]);
    const cf s2 = cadd(b[1], b[3]), s3 = csub(b[1], b[3]);
    X[0] = cadd(s0, s2);
    X[2] = csub(s0, s2);
    X[1].re = s1.re + s3.im; X[1].im = s1.im - s3.re;
    X[3].re = s1.re - s3.im; X[3].im = s1.im + s3.re;
}

static void dft8(const cf a[8], cf X[8])
{
    const float c = 0.70710678118654752440f;
    cf e[4] = {a[0], a[2], a[4], a[6]}, o[4] = {a[1], a[3], a[5], a[7]}, E[4], O[4], T[4];
    dft4(e, E);
    dft4(o, O);
    T[0] = O[0];
    T[1].re = (O[1].re + O[1].im) * c; T[1].im = (O[1].im - O[1].re) * c;
    T[2].re = O[2].im;                 T[2].im = -O[2].re;
    T[3].re = (O[3].im - O[3].re) * c; T[3].im = -((O[3].re + O[3].im) * c);
    for (int k = 0; k < 4; k++) { X[k] = cadd(E[k], T[k]); X[k + 4] = csub(E[k], T[k]); }
}

/* 5-point DFT by its real-factor symmetry (36 operations instead of 80), fixed order:
 *   t1 = b1+b4, t2 = b2+b3, t3 = b1-b4, t4 = b2-b3,  X0 = (b0+t1)+t2
 *   a1 = fma(C2,t2, fma(C1,t1,b0)),  a2 = fma(C1,t2, fma(C2,t1,b0))         C = cos(2 pi j/5)
 *   s1 = fma(S2,t4, S1*t3),          s2 = fma(-S1,t4, S2*t3)                S = sin(2 pi j/5)
 *   X1 = a1 - i s1, X4 = a1 + i s1, X2 = a2 - i s2, X3 = a2 + i s2 */
static void dft5(const cf b[5], cf X[5])
{
    const float C1 = kTw5[2], C2 = kTw5[4], S1 = -kTw5[3], S2 = -kTw5[5];
    const cf t1 = cadd(b[1], b[4]), t2 = cadd(b[2], b[3]), t3 = csub(b[1], b[4]), t4 = csub(b[2], b[3]);
    cf a1, a2, s1, s2;
    X[0] = cadd(cadd(b[0], t1), t2);
    a1.re = fmaf(C2, t2.re, fmaf(C1, t1.re, b[0].re)); a1.im = fmaf(C2, t2.im, fmaf(C1, t1.im, b[0].im));
    a2.re = fmaf(C1, t2.re, fmaf(C2, t1.re, b[0].re)); a2.im = fmaf(C1, t2.im, fmaf(C2, t1.im, b[0].im));
    s1.re = fmaf(S2, t4.re, S1 * t3.re);  s1.im = fmaf(S2, t4.im, S1 * t3.im);
    s2.re = fmaf(-S1, t4.re, S2 * t3.re); s2.im = fmaf(-S1, t4.im, S2 * t3.im);
    X[1].re = a1.re + s1.im; X[1].im = a1.im - s1.re;
    X[4].re = a1.re - s1.im; X[4].im = a1.im + s1.re;
    X[2].re = a2.re + s2.im; X[2].im = a2.im - s2.re;
    X[3].re = a2.re - s2.im; X[3].im = a2.im + s2.re;
}

/* the shipped specification: prime-factor 8 x 5 */
static void fft40(const cf* u, cf* X)
{
    cf A[5][8];
    for (int n2 = 0; n2 < 5; n2++) {
        cf a[8];
        for (int n1 = 0; n1 < 8; n1++) a[n1] = u[(5 * n1 + 8 * n2) % 40];
        dft8(a, A[n2]);
    }
    for (int k1 = 0; k1 < 8; k1++) {
        cf b[5], Y[5];
        for (int n2 = 0; n2 < 5; n2++) b[n2] = A[n2][k1];
        dft5(b, Y);
        for (int k2 = 0; k2 < 5; k2++) X[(25 * k1 + 16 * k2) % 40] = Y[k2];
    }
}

/* rounds 1-3: Cooley-Tukey 8 x 5 with twiddles (the A/B partners' specification) */
static int g_legacy_fft = 0;
void oracle_pfb_legacy_fft(int on) { g_legacy_fft = on != 0; }

static void fft40_ct(const cf* u, cf* X)
{
    cf B[5][8];
    for (int n2 = 0; n2 < 5; n2++) {
        cf a[8], A[8];
        for (int n1 = 0; n1 < 8; n1++) a[n1] = u[5 * n1 + n2];
        dft8(a, A);
        for (int k1 = 0; k1 < 8; k1++) {
            const int j = (n2 * k1) % 40;
            B[n2][k1] = j ? cmul_tw(A[k1], kTw40[2 * j], kTw40[2 * j + 1]) : A[k1];
        }
    }
    for (int k1 = 0; k1 < 8; k1++) {
        cf b[5], Y[5];
        for (int n2 = 0; n2 < 5; n2++) b[n2] = B[n2][k1];
        dft5(b, Y);
        for (int k2 = 0; k2 < 5; k2++) X[k1 + 8 * k2] = Y[k2];
    }
}

static void fft16(const cf* u, cf* X)
{
    cf B[4][4];
    for (int n2 = 0; n2 < 4; n2++) {
        cf a[4], A[4];
        for (int n1 = 0; n1 < 4; n1++) a[n1] = u[4 * n1 + n2];
        dft4(a, A);
        for (int k1 = 0; k1 < 4; k1++) {
            const int j = (n2 * k1) % 16;
            B[n2][k1] = j ? cmul_tw(A[k1], kTw16[2 * j], kTw16[2 * j + 1]) : A[k1];
        }
    }
    for (int k1 = 0; k1 < 4; k1++) {
        cf b[4], Y[4];
        for (int n2 = 0; n2 < 4; n2++) b[n2] = B[n2][k1];
        dft4(b, Y);
        for (int k2 = 0; k2 < 4; k2++) X[k1 + 4 * k2] = Y[k2];
    }
}

uint64_t oracle_pfb_nout(uint64_t n, uint32_t M)
{
    const uint64_t L = (uint64_t)M * 16u, D = M / 2u;
    return n >= L ? (n - L) / D + 1u : 0u;
}

const float* oracle_pfb_proto(uint32_t M) { return M == 40 ? kPfbProto40 : (M == 16 ? kPfbProto16 : NULL); }

/* The M = 40 FIR in the term order of the EXPERIMENTAL matrix-pipe kernel (snout_amd/csrc/pfb_mfma.hip, selected with
 * SNOUT_PFB_IMPL=mfma; the shipped kernel pfb_spec.hip runs the 16-term chain above).  There the 128 output times of a
 * tile are, per branch r and output parity e, four blocks of 16 consecutive outputs idx = 16 b + j of the stream
 * z[q] = x[128 D tile + q M + e D + r]; a block is ONE product over the 32 window positions k = 0..31 with the banded
 * Toeplitz matrix of the branch's taps:
 *     u = fmaf(z[16 b + k], t, u)  for ascending k,   t = h[r + (k - j) M] if 0 <= k - j < 16, else +0,
 * the operand of k = 31 (no output has a tap there) being +0 instead of a sample; samples past the end of the segment
 * read as +0.  A zero tap contributes (sample x 0) = +-0, which leaves a non-zero finite sum untouched, so this equals
 * the 16-term chain unless a sample of the window is not finite (NaN / Inf x 0 = NaN reaches all 16 outputs of the
 * block) or the result is a zero (whose sign the +-0 terms can change).  oracle_pfb_block_order() runs the 16-term
 * chain and comes here in exactly those two cases. */
static void fir40_block_order(const float* iq, uint64_t n, uint64_t m, uint32_t r, const float* h, float* re, float* im)
{
    const uint32_t M = 40, D = 20, T = 128;
    const uint64_t tile = m / T;
    const uint32_t mt = (uint32_t)(m % T), e = mt & 1u, idx = mt >> 1, b = idx >> 4, j = idx & 15u;
    float ar = 0.0f, ai = 0.0f;
    for (uint32_t k = 0; k < 32; k++) {
        const uint64_t s = tile * (uint64_t)(T * D) + (uint64_t)(16u * b + k) * M + e * D + r;
        float xr = 0.0f, xi = 0.0f;
        if (k < 31 && s < n) { xr = iq[2 * s]; xi = iq[2 * s + 1]; }
        const float t = (k >= j && k - j < 16u) ? h[r + (k - j) * M] : 0.0f;
        ar = fmaf(xr, t, ar);
        ai = fmaf(xi, t, ai);
    }
    *re = ar; *im = ai;
}

/* y: [M][y_stride] interleaved complex floats (2 floats per sample) */
static int pfb_impl(const float* iq, uint64_t n, uint32_t M, float* y, uint64_t y_stride, int block_order);
int oracle_pfb(const float* iq, uint64_t n, uint32_t M, float* y, uint64_t y_stride) { return pfb_impl(iq, n, M, y, y_stride, 0); }
/* the experimental matrix-pipe kernel's term order (M = 40 only differs) */
int oracle_pfb_block_order(const float* iq, uint64_t n, uint32_t M, float* y, uint64_t y_stride) { return pfb_impl(iq, n, M, y, y_stride, 1); }

static int pfb_impl(const float* iq, uint64_t n, uint32_t M, float* y, uint64_t y_stride, int block_order)
{
    if (M != 40 && M != 16) return -1;
    const float* h = oracle_pfb_proto(M);
    const uint32_t D = M / 2, P = 16;
    const uint64_t n_out = oracle_pfb_nout(n, M);
    /* any sample that is not finite?  (exponent all ones; one pass, only the M = 40 term order cares) */
    int nonfinite = 0;
    if (M == 40 && block_order) {
        const uint32_t* w = (const uint32_t*)iq;
        uint32_t seen = 0;
#pragma omp parallel for schedule(static) reduction(|:seen)
        for (int64_t i = 0; i < (int64_t)(2 * n); i++) seen |= ((w[i] & 0x7F800000u) == 0x7F800000u);
        nonfinite = seen != 0;
    }
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < (int64_t)n_out; m++) {
        cf u[40], X[40];
        const float* x = iq + 2ull * (uint64_t)m * D;
        for (uint32_t r = 0; r < M; r++) {
            float ar = 0.0f, ai = 0.0f;
            for (uint32_t p = 0; p < P; p++) {
                const float c = h[r + p * M];
                ar = fmaf(c, x[2 * (r + p * M)], ar);
                ai = fmaf(c, x[2 * (r + p * M) + 1], ai);
            }
            if (M == 40 && block_order && (nonfinite || ar == 0.0f || ai == 0.0f)) fir40_block_order(iq, n, (uint64_t)m, r, h, &ar, &ai);
            u[r].re = ar; u[r].im = ai;
        }
        if (M == 40) { if (g_legacy_fft) fft40_ct(u, X); else fft40(u, X); } else fft16(u, X);
        for (uint32_t k = 0; k < M; k++) {
            cf v = X[k];
            if (k & (uint32_t)m & 1u) { v.re = -v.re; v.im = -v.im; }
            y[2 * (k * y_stride + (uint64_t)m)] = v.re;
            y[2 * (k * y_stride + (uint64_t)m) + 1] = v.im;
        }
    }
    return 0;
}

/* ---- wideband receivers: channelize, then the per-channel oracle on every bin --------------- */
static int32_t btle_rf_to_channel(uint32_t k)
{
    if (k == 0) return 37;
    if (k == 12) return 38;
    if (k == 39) return 39;
    if (k >= 1 && k <= 11) return (int32_t)k - 1;
    return (int32_t)k - 2;
}

/* bin b of the M = 40 channelizer (centre 2442 MHz) carries RF index (b + 20) mod 40 */
uint32_t oracle_btle_bin_channel(uint32_t bin) { return (uint32_t)btle_rf_to_channel((bin + 20u) % 40u); }
/* bin b of the M = 16 channelizer carries synthetic 802.15.4 channel 11 + (b + 8) mod 16 */
uint32_t oracle_zigbee_bin_channel(uint32_t bin) { return 11u + (bin + 8u) % 16u; }

int oracle_wideband_segment(const float* iq, uint64_t n, uint64_t first_index, uint32_t proto,
                            uint32_t access_addr, uint32_t crc_init, uint32_t threshold,
                            uint32_t core, uint32_t warmup, snout_pkt* out, uint64_t cap,
                            uint64_t* n_out)
{
    *n_out = 0;
    const uint32_t M = proto == 0 ? 40u : 16u;
    const uint64_t nc = oracle_pfb_nout(n, M);
    if (nc == 0) return 0;
    float* y = (float*)malloc((size_t)M * nc * 2u * sizeof(float));
    if (!y) return -3;
    int rc = oracle_pfb(iq, n, M, y, nc);
    /* the channels are independent: one OpenMP task per bin, each into its own buffer, joined in
     * bin order (the all-cores leg of bench.py's cpu_baseline; with one thread this is the plain
     * loop over the bins) */
    (void)oracle_fast_atan2f(0.0f, 1.0f);       /* lazily built tables: before the threads start */
    snout_pkt* part[40] = {0};
    uint64_t got[40] = {0}, pcap[40];
    int prc[40] = {0};
    for (uint32_t b = 0; b < M; b++) {
        pcap[b] = nc / 256u + 64u;
        part[b] = (snout_pkt*)malloc((size_t)pcap[b] * sizeof(snout_pkt));
        if (!part[b]) rc = -3;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < (int)M; b++) {
        if (rc) continue;
        const float* yc = y + 2ull * (uint64_t)b * nc;
        for (;;) {
            if (proto == 0)
                prc[b] = oracle_btle_segment(yc, nc, first_index, oracle_btle_bin_channel((uint32_t)b), access_addr,
                                             crc_init, part[b], pcap[b], &got[b], NULL, 0, NULL);
            else
                prc[b] = oracle_zigbee_segment(yc, nc, first_index, oracle_zigbee_bin_channel((uint32_t)b), threshold,
                                               core, warmup, part[b], pcap[b], &got[b]);
            if (prc[b] != -5 || got[b] <= pcap[b]) break;
            pcap[b] = got[b] + 64u;                       /* capacity: grow and redo this bin */
            snout_pkt* np = (snout_pkt*)realloc(part[b], (size_t)pcap[b] * sizeof(snout_pkt));
            if (!np) { prc[b] = -3; break; }
            part[b] = np;
        }
    }
    uint64_t total = 0;
    for (uint32_t b = 0; b < M; b++) {
        if (!rc && prc[b] && prc[b] != -5) rc = prc[b];
        if (!rc && part[b]) {
            for (uint64_t i = 0; i < got[b] && i < pcap[b]; i++)
                if (total + i < cap) out[total + i] = part[b][i];
            total += got[b];
        }
        free(part[b]);
    }
    free(y);
    *n_out = total;
    return rc ? rc : (total > cap ? -5 : 0);
}

/* ---- threads ---------------------------------------------------------------------------------- */
#ifdef _OPENMP
#include <omp.h>
int  oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
int  oracle_hw_threads(void) { return omp_get_num_procs(); }
#else
int  oracle_set_threads(int n) { (void)n; return 1; }
int  oracle_hw_threads(void) { return 1; }
#endif

/* Overlapping segments of one narrowband capture in parallel (one OpenMP task per segment), the
 * all-cores form of the single-channel receivers: segment s covers [s seg, (s+1) seg + overlap) and
 * reports the packets that start before (s+1) seg.  out is NOT sorted across segments beyond the
 * segment order; returns the total count in *n_out (-5 if it exceeds cap). */
int oracle_narrowband_parallel(const float* iq, uint64_t n, uint32_t proto, uint32_t channel,
                               uint32_t access_addr, uint32_t crc_init, uint32_t threshold, uint32_t core,
                               uint32_t warmup, uint64_t seg, uint64_t overlap, snout_pkt* out, uint64_t cap,
                               uint64_t* n_out)
{
    *n_out = 0;
    if (seg == 0) return -1;
    const uint64_t ns = (n + seg - 1) / seg;
    (void)oracle_fast_atan2f(0.0f, 1.0f);       /* lazily built tables: before the threads start */
    uint64_t* cnt = (uint64_t*)calloc(ns ? ns : 1, sizeof(uint64_t));
    snout_pkt** part = (snout_pkt**)calloc(ns ? ns : 1, sizeof(snout_pkt*));
    if (!cnt || !part) { free(cnt); free(part); return -3; }
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t s = 0; s < (int64_t)ns; s++) {
        const uint64_t a = (uint64_t)s * seg;
        const uint64_t b = a + seg + overlap < n ? a + seg + overlap : n;
        const uint64_t pc = (b - a) / 256u + 64u;
        snout_pkt* buf = (snout_pkt*)malloc((size_t)pc * sizeof(snout_pkt));
        uint64_t got = 0;
        int r = -3;
        if (buf)
            r = proto == 0 ? oracle_btle_segment(iq + 2 * a, b - a, a, channel, access_addr, crc_init, buf, pc, &got, NULL, 0, NULL)
                           : oracle_zigbee_segment(iq + 2 * a, b - a, a, channel, threshold, core, warmup, buf, pc, &got);
        if (r && r != -5) {
#pragma omp critical
            rc = r;
        }
        uint64_t keep = 0;
        for (uint64_t i = 0; buf && i < got && i < pc; i++)
            if (buf[i].sample_index < a + seg || (uint64_t)s + 1 == ns) buf[keep++] = buf[i];
        part[s] = buf;
        cnt[s] = keep;
    }
    uint64_t total = 0;
    for (uint64_t s = 0; s < ns; s++) {
        for (uint64_t i = 0; i < cnt[s]; i++)
            if (total + i < cap) out[total + i] = part[s][i];
        total += cnt[s];
        free(part[s]);
    }
    free(cnt);
    free(part);
    *n_out = total;
    return rc ? rc : (total > cap ? -5 : 0);
}

/* Overlapping segments of one WIDEBAND capture in parallel: one OpenMP task per segment, each the
 * whole single-threaded chain (channelizer + every channel's receiver; the inner parallel loops run
 * serially inside a task).  seg must be a multiple of M (keeps the channelizer's phase).  The
 * all-cores leg of bench.py's cpu_baseline for cfg #3 / #4. */
int oracle_wideband_parallel(const float* iq, uint64_t n, uint32_t proto, uint32_t access_addr,
                             uint32_t crc_init, uint32_t threshold, uint32_t core, uint32_t warmup,
                             uint64_t seg, snout_pkt* out, uint64_t cap, uint64_t* n_out)
{
    *n_out = 0;
    const uint32_t M = proto == 0 ? 40u : 16u, D = M / 2u;
    if (seg == 0 || seg % M) return -1;
    const uint64_t overlap = (proto == 0 ? 1600ull : 17024ull + 2048ull) * D + 16ull * M;
    const uint64_t ns = (n + seg - 1) / seg;
    (void)oracle_fast_atan2f(0.0f, 1.0f);
    uint64_t* cnt = (uint64_t*)calloc(ns ? ns : 1, sizeof(uint64_t));
    snout_pkt** part = (snout_pkt**)calloc(ns ? ns : 1, sizeof(snout_pkt*));
    if (!cnt || !part) { free(cnt); free(part); return -3; }
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t s = 0; s < (int64_t)ns; s++) {
        const uint64_t a = (uint64_t)s * seg;
        const uint64_t b = a + seg + overlap < n ? a + seg + overlap : n;
        const uint64_t pc = (b - a) / 1024u + 256u;
        snout_pkt* buf = (snout_pkt*)malloc((size_t)pc * sizeof(snout_pkt));
        uint64_t got = 0;
        int r = buf ? oracle_wideband_segment(iq + 2 * a, b - a, a / D, proto, access_addr, crc_init, threshold,
                                              core, warmup, buf, pc, &got) : -3;
        if (r) {
#pragma omp critical
            rc = r;
        }
        uint64_t keep = 0;
        for (uint64_t i = 0; buf && i < got && i < pc; i++)
            if (buf[i].sample_index < (a + seg) / D || (uint64_t)s + 1 == ns) buf[keep++] = buf[i];
        part[s] = buf;
        cnt[s] = keep;
    }
    uint64_t total = 0;
    for (uint64_t s = 0; s < ns; s++) {
        for (uint64_t i = 0; i < cnt[s]; i++)
            if (total + i < cap) out[total + i] = part[s][i];
        total += cnt[s];
        free(part[s]);
    }
    free(cnt);
    free(part);
    *n_out = total;
    return rc ? rc : (total > cap ? -5 : 0);
}
