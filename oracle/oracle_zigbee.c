/*
 * oracle_zigbee.c — CPU restatement of the IEEE 802.15.4 (Zigbee) receive chain.
 * TEST INFRASTRUCTURE ONLY (see oracle_btle.c for the rules).
 *
 * PARITY UNPINNED: the topology and every parameter come from the reference's flowgraph
 *   snout/modulations/Zigbee/hackrf/Zigbee_rx/top_block.py
 *     :73  analog.quadrature_demod_cf(1)
 *     :52  filter.single_pole_iir_filter_ff(0.00016, 1)   :70 blocks.sub_ff(1)   (:84-86,:89 wiring)
 *     :69  digital.clock_recovery_mm_ff(2, 0.000225, 0.5, 0.03, 0.0002)
 *     :67  ieee802_15_4.packet_sink(10)
 * but the block arithmetic lives in GNU Radio 3.7.13.5 and bastibl/gr-ieee802-15-4 (unpinned,
 * installed by PyBOMBS, Makefile:40-41), neither present in /root/reference nor importable here.
 * The blocks are restated from their published algorithms.  The 8-tap
 * MMSE interpolator bank (GNU Radio interpolator_taps.h) is not available offline and is
 * REGENERATED (oracle/mmse_taps.inc, tools/gen_mmse_taps.py): low-order bits differ from GNU
 * Radio's table.
 *
 * Segmentation (stated deviation): the stream is cut into lanes.  Lane i owns the
 * core [i*core, (i+1)*core) of the channel's samples; it starts `warmup` samples early with all
 * loop state at its initial value and reports a frame only if the chip that first matched the
 * preamble lies in its core; it runs past its core only to finish such a frame.  The GPU path
 * processes the identical lanes, so both sides compute the same thing.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static const float kMmseTaps[129][8] = {
#include "mmse_taps.inc"
};

/* RX correlator words of gr-ieee802-15-4 (FM-domain chip words, MSB = first chip). */
static const uint32_t kChipMap[16] = {
    1618456172u, 1309113062u, 1826650030u, 1724778362u, 778887287u, 2061946375u, 2007919840u,
    125494990u,  529027475u,  838370585u,  320833617u,  422705285u, 1368596360u, 85537272u,
    139563807u,  2021988657u};

const uint32_t* oracle_zb_chip_map(void) { return kChipMap; }
const float* oracle_zb_mmse_taps(void) { return &kMmseTaps[0][0]; }

/* ---- a4: quadrature_demod_cf(1) with GNU Radio's table-driven fast_atan2f ------------------- */
static float g_atan_tab[257];
static int g_atan_ready = 0;
static void atan_init(void)
{
    if (g_atan_ready) return;
    for (int i = 0; i < 257; i++) g_atan_tab[i] = (float)atan((double)i / 255.0);
    g_atan_ready = 1;
}

float oracle_fast_atan2f(float y, float x)
{
    atan_init();
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
        base = g_atan_tab[k];
        base += (g_atan_tab[k + 1] - g_atan_tab[k]) * a;
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

/* d[t] = fast_atan2f(Im(x[t] conj x[t-1]), Re(...)), x[-1] = 0 */
void oracle_zb_discrim(const float* iq, uint64_t n, float* d)
{
    atan_init();
    /* (pointwise: with more than one OpenMP thread set -- oracle_set_threads, never inside another parallel region's
     * threads -- the samples are shared out; the values do not depend on it) */
#pragma omp parallel for schedule(static)
    for (uint64_t t = 0; t < n; t++) {
        const float pr = t ? iq[2 * t - 2] : 0.0f, pi = t ? iq[2 * t - 1] : 0.0f;
        const float ar = iq[2 * t], ai = iq[2 * t + 1];
        const float re = ar * pr + ai * pi;
        const float im = ai * pr - ar * pi;
        float a = oracle_fast_atan2f(im, re);
        if (!(fabsf(a) <= 4.0f)) a = 0.0f;      /* non-finite input: defined as 0 (stated deviation) */
        d[t] = a;
    }
}

/* ---- lane state ----------------------------------------------------------------------------- */
typedef struct {
    /* a5 */
    double lp;
    /* a6 */
    float mu, omega, last;
    /* a7 */
    int state;          /* 0 search, 1 have_sync, 2 have_header */
    uint32_t shift;
    int preamble_cnt, chip_cnt, packet_byte, byte_index, packetlen, packetlen_cnt, payload_cnt;
    unsigned lqi, lqi_cnt;
    uint64_t trigger;   /* input sample index at which the first preamble symbol matched */
    uint64_t sync;      /* `at` of the chip that completed the SFD (state 0 -> 1) */
    int chip_err;       /* the last return to search came from a symbol without a chip word within the threshold */
    uint8_t pkt[128];
} lane_t;

static inline unsigned popc(uint32_t v) { return (unsigned)__builtin_popcount(v); }

static void enter_search(lane_t* s)
{
    s->state = 0; s->shift = 0; s->preamble_cnt = 0; s->chip_cnt = 0; s->packet_byte = 0; s->chip_err = 0;
}

static int decode_chips(lane_t* s, unsigned threshold)
{
    int best = 0xFF;
    unsigned min_t = 33;
    for (int i = 0; i < 16; i++) {
        unsigned t = popc((s->shift & 0x7FFFFFFEu) ^ (kChipMap[i] & 0x7FFFFFFEu));
        if (t < min_t) { best = i; min_t = t; }
    }
    if (min_t < threshold) {
        if (s->lqi_cnt < 8) { s->lqi += 32 - min_t; s->lqi_cnt++; }
        return best & 0xF;
    }
    return 0xFF;
}

uint16_t oracle_crc16_154(const uint8_t* d, int n)
{
    uint16_t c = 0;
    for (int i = 0; i < n; i++) {
        c ^= d[i];
        for (int k = 0; k < 8; k++) c = (c & 1) ? (uint16_t)((c >> 1) ^ 0x8408) : (uint16_t)(c >> 1);
    }
    return c;
}

/* Feed one chip (soft value) into the packet sink. Returns 1 when a frame completed. */
static int sink_chip(lane_t* s, float chip, uint64_t at, unsigned th)
{
    s->shift = (s->shift << 1) | (chip > 0.0f ? 1u : 0u);
    switch (s->state) {
    case 0:
        if (s->preamble_cnt > 0) s->chip_cnt++;
        if (s->preamble_cnt == 0) {
            if (popc((s->shift & 0x7FFFFFFEu) ^ (kChipMap[0] & 0x7FFFFFFEu)) < th) {
                s->preamble_cnt = 1;
                s->trigger = at;
            }
        } else if (s->chip_cnt == 32) {
            s->chip_cnt = 0;
            if (s->packet_byte == 0) {
                if (popc((s->shift & 0x7FFFFFFEu) ^ (kChipMap[0] & 0x7FFFFFFEu)) <= th) {
                    s->preamble_cnt++;
                } else if (popc((s->shift & 0x7FFFFFFEu) ^ (kChipMap[7] & 0x7FFFFFFEu)) <= th) {
                    s->packet_byte = 7 << 4;
                } else {
                    enter_search(s);
                }
            } else {
                if (popc((s->shift & 0x7FFFFFFEu) ^ (kChipMap[10] & 0x7FFFFFFEu)) <= th) {
                    s->packet_byte |= 0xA;
                    s->sync = at;
                    s->state = 1; s->packetlen_cnt = 0; s->packet_byte = 0; s->byte_index = 0;
                    s->lqi = 0; s->lqi_cnt = 0;
                } else {
                    enter_search(s);
                }
            }
        }
        return 0;
    case 1:
        s->chip_cnt++;
        if (s->chip_cnt == 32) {
            s->chip_cnt = 0;
            int c = decode_chips(s, th);
            if (c == 0xFF) { enter_search(s); s->chip_err = 1; return 0; }
            if (s->byte_index == 0) s->packet_byte = c; else s->packet_byte |= c << 4;
            s->byte_index++;
            if (s->byte_index % 2 == 0) {
                int len = s->packet_byte;
                if (len <= 127) {
                    s->state = 2; s->packetlen = len; s->payload_cnt = 0; s->packet_byte = 0;
                    s->byte_index = 0;
                } else {
                    enter_search(s);
                }
            }
        }
        return 0;
    default:
        s->chip_cnt = (s->chip_cnt + 1) % 32;
        if (s->chip_cnt == 0) {
            int c = decode_chips(s, th);
            if (c == 0xFF) { enter_search(s); s->chip_err = 1; return 0; }
            if (s->byte_index == 0) s->packet_byte = c; else s->packet_byte |= c << 4;
            s->byte_index++;
            if (s->byte_index % 2 == 0) {
                s->pkt[s->packetlen_cnt++] = (uint8_t)s->packet_byte;
                s->payload_cnt++;
                s->byte_index = 0;
                if (s->payload_cnt >= s->packetlen) return 1;   /* caller publishes + enter_search */
            }
        }
        return 0;
    }
}

/*
 * Lanes.  The clock-recovery loop is a feedback loop, so the segment is cut into
 * lanes: lane l of a channel covers the core [l core, (l+1) core) and starts its IIR + M&M `warmup`
 * samples early (IIR state carried in exactly, see below; M&M from its initial state).  A lane
 * produces every chip whose interpolator window starts before its core end.
 *
 * Stitching.  Each chip has the time key T = 128 * (absolute window start) + rint(128 mu)  (the
 * interpolation instant in 1/128 sample).  Lane 0 owns all its chips.  For lane l+1, the candidates
 * are its chips whose window start lies in [core_start - 3, core_start + 5]; f0 = the first
 * candidate with T >= E_l (the chip after the last candidate if none), where E_l = T(last chip of
 * lane l) + 128, one sample = half a chip period later (E_l = 128 * core_end if lane l produced no
 * chip).  The discriminator output of O-QPSK/MSK is close to rectangular, so the M&M detector has
 * a wide dead zone and two loops locked onto the same signal may sit up to a sample apart, where
 * time alone cannot tell which chip is "the next one".  The chip VALUES decide: for s in (0, -1,
 * +1) the 48 chips of lane l+1 ending at chip f0+s-1 are compared with the last 48 chips of lane l;
 * the shift with the most agreements wins (a later shift must be strictly better; a shift whose
 * 48 chips are not all available is skipped, and no comparison is made if lane l has fewer than 48
 * chips).  Lane l+1 owns its chips from index f0+s.  The owned chips of consecutive lanes thus
 * join into ONE chip stream per channel without a repeated or a missing chip.
 *
 * Sinks.  gr-ieee802-15-4's packet sink runs over the stitched stream once per lane: it starts
 * (searching, register cleared) ORACLE_ZB_SINK_WARM chips before the lane's first owned chip (or
 * at the start of the stream) and stops when it is idle at or past the next lane's first owned
 * chip.  Sinks that start at different chips fall into step as soon as both are searching with a
 * full register; until then they may first match different symbols of one preamble, but they find
 * the start-of-frame delimiter at the same chip, so a frame is reported by the lane that owns the
 * chip completing its SFD.
 *
 * Resolve (what makes the lanes' frames those of ONE sequential sink).  The sequential sink is busy
 * from the chip that first matched a preamble symbol (its trigger) to the chip that completes the
 * frame, and searches again only from the next chip: it cannot report a frame that begins inside
 * another one.  A lane's sink that starts inside a frame can (it false-syncs on payload symbols).
 * So the candidate frames of all lanes, in the order of their SFD chips, are passed through the
 * sequential rule: a frame is kept iff its trigger chip lies after the last chip of the frame kept
 * before it.
 *
 * sample_index.  Which preamble symbol a sink matches first depends on where it started, the SFD
 * chip does not: the record's sample_index is the window start of the chip 319 chips before the
 * one that completes the SFD (= the first chip of a regular 8-symbol preamble + 2-symbol SFD; chip
 * 0 if the stream is shorter), the same for every sink that finds the frame.
 * With core >= n (one lane) this is the reference's sequential receiver.
 */
#define ORACLE_ZB_SINK_WARM 512u

/*
 * Frame repair (round 5; what makes the lanes' frames those of the ONE sequential loop on dense traffic).
 * At two samples per chip the M&M loop has two stable lock points one sample (half a chip) apart.  A loop that meets a
 * frame at its preamble takes the one that decodes and keeps it to the end of the frame; a lane that starts inside a
 * payload takes whichever is nearer to its starting phase, and when that is the other one the stitched stream slips
 * half a chip at the seam: the sink gives the frame up at the first symbol with >= threshold chip errors
 * (profiles/r4_lane_residual.md: 4-7 % of the sequential receiver's frames on cfg #4's traffic, all of them frames
 * with a seam behind their SFD).  The sequential loop has no seams.  So a frame that a lane's sink
 *   - has synchronised to (SFD found at a chip the lane owns) and then
 *   - gives up at a symbol (state 1 or 2, no chip word within the threshold), or completes with a bad FCS,
 *   - at a chip that a LATER lane owns (a seam lies between the SFD and that chip),
 * is received again the way the sequential receiver receives it -- by the loop that met its preamble going ON instead
 * of handing over: lane l's loop continues from its state at its core end (every z the value lane l's own filter
 * recurrence gives), lane l's sink from its state at the seam (taken when it crosses the next lane's first owned chip
 * while busy, state 1 or 2), over the samples of lane l + 1.  Where lane l + 2 takes over -- from half a chip before the
 * window start of its first owned chip -- the sink returns to the stitched stream (bounded work: one lane's samples; a
 * second wrong hand-over inside one frame is rare and loses the frame).  What that run decodes replaces what the lanes
 * decoded: a record if the frame completes (SNOUT_PKT_ZB_REPAIRED), nothing if it is given up again -- then the loop that
 * met the preamble loses it as well.  For the sequential rule (Resolve) the repaired frame occupies the stream chips
 * from the trigger found before to the chip it ends at; it takes its place in the lane's records where the lane's sink
 * stopped.
 * One lane (core >= n) has no seams and repairs nothing: the reference's receiver itself.
 */

static int g_zb_repair = 1;
/* test / analysis knob: 0 = the lanes alone (rounds 1-4), 1 = with frame repair (the product's default) */
void oracle_zb_set_repair(int on) { g_zb_repair = on; }

typedef struct {
    uint64_t n_chips;        /* chips produced by the lane's M&M */
    uint64_t first_owned;    /* index of the first owned chip (<= n_chips) */
    uint64_t t_last;         /* key of the last chip (valid if n_chips) */
} lane_chips_t;

/* The M&M recurrence of one loop: state + one step.  z holds the DC-removed samples in a ring of 16. */
typedef struct {
    double lp;
    float mu, omega, last;
    float ring[16];
    uint64_t ii, z_next;
} mm_t;

static void mm_start(mm_t* m, uint64_t iir_from, double lp_init, uint64_t ii)
{
    memset(m, 0, sizeof(*m));
    m->lp = lp_init; m->mu = 0.5f; m->omega = 2.0f; m->last = 0.0f;
    m->ii = ii; m->z_next = iir_from;
}

/* one chip: the interpolated sample at (ii, mu); advances the loop.  Caller guarantees ii + 8 <= n. */
static inline float mm_step(mm_t* m, const float* d, uint64_t n, int* imu_out, float* soft_z, uint64_t soft_base,
                            uint64_t soft_cap)
{
    const double alpha = 0.00016, one_minus = 1.0 - 0.00016;
    const float omega_mid = 2.0f, gain_omega = 0.000225f, gain_mu = 0.03f;
    const float omega_lim = omega_mid * 0.0002f;
    while (m->z_next < m->ii + 8 && m->z_next < n) {
        const float x = d[m->z_next];
        m->lp = alpha * (double)x + one_minus * m->lp;
        const float z = x - (float)m->lp;
        m->ring[m->z_next & 15] = z;
        if (soft_z && m->z_next - soft_base < soft_cap) soft_z[m->z_next - soft_base] = z;
        m->z_next++;
    }
    float win[8];
    for (int k = 0; k < 8; k++) win[k] = m->ring[(m->ii + k) & 15];
    const int imu = (int)rintf(m->mu * 128.0f);
    float acc = 0.0f;
    for (int k = 0; k < 8; k++) acc = fmaf(kMmseTaps[imu][k], win[7 - k], acc);
    const float o = acc;
    *imu_out = imu;
    const float mm = (m->last < 0.0f ? -1.0f : 1.0f) * o - (o < 0.0f ? -1.0f : 1.0f) * m->last;
    m->last = o;
    m->omega = m->omega + gain_omega * mm;
    {
        const float x = m->omega - omega_mid;
        const float c = 0.5f * (fabsf(x + omega_lim) - fabsf(x - omega_lim));
        m->omega = omega_mid + c;
    }
    m->mu = m->mu + m->omega + gain_mu * mm;
    const float fl = floorf(m->mu);
    m->ii += fl >= 1.0f ? (uint64_t)(int)fl : 1u;     /* 1..3 for finite input */
    m->mu = m->mu - fl;
    return o;
}

/* a5 + a6 of one lane; appends (bit, pos) of ALL its chips to bits/pos/key (capacity ensured by the
 * caller: (core + warmup) chips at most).  soft_z / soft_chips: optional taps. */
static uint64_t mm_lane(const float* d, uint64_t n, uint64_t core_start, uint64_t core_len,
                        uint64_t warmup, double lp_init, uint8_t* bits, uint64_t* pos, uint64_t* key,
                        float* soft_z, float* soft_chips, uint64_t soft_cap, mm_t* end_state)
{
    const uint64_t s0 = core_start > warmup ? core_start - warmup : 0;
    const uint64_t core_end = core_start + core_len;
    mm_t m;
    mm_start(&m, s0, lp_init, s0);
    uint64_t chips = 0;
    while (m.ii < core_end && m.ii + 8 <= n) {
        const uint64_t at = m.ii;
        int imu;
        const float o = mm_step(&m, d, n, &imu, soft_z, s0, soft_cap);
        if (soft_chips && chips < soft_cap) soft_chips[chips] = o;
        if (bits) { bits[chips] = o > 0.0f; pos[chips] = at; key[chips] = at * 128u + (uint64_t)imu; }
        chips++;
    }
    if (end_state) *end_state = m;
    return chips;
}

/*
 * IIR carry-in.  GNU Radio's single-pole IIR runs over the continuous stream, so when a frame
 * arrives its DC estimate has long converged (time constant 1/alpha = 6250 samples).  A lane that
 * started the recurrence from zero would see the uncorrected CFO offset for its first thousands of
 * samples.  The linear recurrence is therefore carried across lanes exactly as a blocked evaluation
 * of the same filter (all in double, fixed operation order, the GPU does the same):
 *   S_j   = (P0 + P1) + (P2 + P3),  P_p = sum in sequence, from 0, over k = 16p .. 16p+15 of
 *           w[63-k] * d[64 j + k],  w[m] = alpha (1-alpha)^m                  (zero-state response)
 *   L_i   = fold over the sub-blocks of lane block i:  L = D64 * L + S_j
 *   lp_in[l] = fold over the lane blocks i = max(0, l - W) .. l-1, from 0:  lp = Dblk_i * lp + L_i
 * where lane block i = [s0_i, s0_{i+1}), s0_i = max(0, i core - warmup), D64 = (1-alpha)^64,
 * Dblk_i = D64^(sub-blocks of the block), powers formed by repeated multiplication, and
 * W = ceil(2^18 / core) lane blocks: what lies further back has decayed by (1-alpha)^(2^18) < 2^-60
 * and is dropped, which makes every lane's carry-in an independent, fixed-length computation.
 * Requires core and warmup to be multiples of 64.
 */
void oracle_zb_iir_tables(double w[64], double* d64)
{
    const double alpha = 0.00016, one_minus = 1.0 - 0.00016;
    double v = alpha;
    for (int m = 0; m < 64; m++) { w[m] = v; v = v * one_minus; }
    double p = 1.0;
    for (int m = 0; m < 64; m++) p = p * one_minus;
    *d64 = p;
}

static double pow_rep(double base, uint64_t e)
{
    double p = 1.0;
    for (uint64_t k = 0; k < e; k++) p = p * base;
    return p;
}

/* lp_in[l] for every lane l of one channel; returns malloc'ed array of n_lanes doubles */
double* oracle_zb_iir_carry(const float* d, uint64_t n, uint32_t core, uint32_t warmup, uint64_t n_lanes)
{
    double w[64], d64;
    oracle_zb_iir_tables(w, &d64);
    const uint64_t nsb = (n + 63) / 64;
    double* S = (double*)malloc((nsb ? nsb : 1) * sizeof(double));
#pragma omp parallel for schedule(static)
    for (uint64_t j = 0; j < nsb; j++) {
        double P[4];
        for (int p = 0; p < 4; p++) {
            double acc = 0.0;
            for (int k = 16 * p; k < 16 * p + 16; k++) {
                const uint64_t t = 64 * j + (uint64_t)k;
                acc = acc + w[63 - k] * (double)(t < n ? d[t] : 0.0f);
            }
            P[p] = acc;
        }
        S[j] = (P[0] + P[1]) + (P[2] + P[3]);
    }
    double* lp_in = (double*)malloc((n_lanes ? n_lanes : 1) * sizeof(double));
    const double dcore = pow_rep(d64, core / 64u);
    const double dfirst = pow_rep(d64, core > warmup ? (core - warmup) / 64u : 0u);
    double* L = (double*)malloc((n_lanes ? n_lanes : 1) * sizeof(double));
#pragma omp parallel for schedule(static)
    for (uint64_t l = 0; l < n_lanes; l++) {
        const uint64_t b0 = l == 0 ? 0 : ((uint64_t)l * core - warmup) / 64u;      /* first sub-block */
        const uint64_t b1 = ((uint64_t)(l + 1) * core - warmup) / 64u;             /* one past last   */
        double a = 0.0;
        for (uint64_t j = b0; j < b1; j++) a = d64 * a + (j < nsb ? S[j] : 0.0);
        L[l] = a;
    }
    const uint64_t W = ((1u << 18) + core - 1) / core;
#pragma omp parallel for schedule(static)
    for (uint64_t l = 0; l < n_lanes; l++) {
        double lp = 0.0;
        for (uint64_t i = l > W ? l - W : 0; i < l; i++) lp = (i == 0 ? dfirst : dcore) * lp + L[i];
        lp_in[l] = lp;
    }
    free(L);
    free(S);
    return lp_in;
}

/* Developer aid (tools/ only): when set, channel_lanes copies its stitched chip stream, the chips' window starts and the
 * lanes' first owned chips here. */
uint8_t* g_dbg_bits = NULL; uint64_t* g_dbg_pos = NULL; uint64_t* g_dbg_first = NULL; uint64_t g_dbg_cap = 0, g_dbg_n = 0, g_dbg_lanes = 0;

/* Frame repair: lane l's loop (state `m` at its core end) and its sink (state `s` at the seam, stream chip q0) go on with
 * the frame the sink is busy with -- over the next lane's samples; where the lane after that takes over (hand_pos = window
 * start of ITS first owned chip, stream chip hand_q; hand_q = 0: there is none) the sink returns to the stitched stream.
 * Returns 1 if the frame completes (s holds it, *end_chip = the stream chip it ends at), 0 if it is given up. */
static int repair_frame(const float* d, uint64_t n, mm_t m, uint64_t q0, uint32_t th, lane_t* s,
                        uint64_t hand_pos, uint64_t hand_q, const uint8_t* sb, uint64_t total, uint64_t* end_chip)
{
    uint64_t c = 0;
    int handed = 0;
    while (m.ii + 8 <= n) {
        /* the chip this step would produce lies at key 128 ii + rint(128 mu); from half a chip before the next-but-one
         * lane's first owned chip (taken at the middle of its window-start sample) the chips are that lane's */
        if (hand_q && m.ii * 128u + (uint64_t)(int)rintf(m.mu * 128.0f) + 128u >= hand_pos * 128u + 64u) { handed = 1; break; }
        int imu;
        const float o = mm_step(&m, d, n, &imu, NULL, 0, 0);
        const int done = sink_chip(s, o, q0 + c, th);
        c++;
        if (done) { *end_chip = q0 + c - 1u; return 1; }
        if (s->state == 0) return 0;        /* given up (a symbol without a chip word within the threshold, or a length > 127) */
    }
    if (!handed) return 0;                  /* the capture ends inside the frame */
    for (uint64_t q = hand_q; q < total; q++) {
        const int done = sink_chip(s, sb[q] ? 1.0f : -1.0f, q, th);
        if (done) { *end_chip = q; return 1; }
        if (s->state == 0) return 0;
    }
    return 0;
}

/* One channel: lanes -> stitched chip stream -> lane-local sinks. */
static void channel_lanes(const float* d, uint64_t n, uint64_t first_index, uint32_t channel,
                          uint32_t threshold, uint32_t core, uint32_t warmup,
                          snout_pkt* out, uint64_t cap, uint64_t* n_out)
{
    const uint64_t n_lanes = (n + core - 1) / core;
    double* lp_in = oracle_zb_iir_carry(d, n, core, warmup, n_lanes);
    const uint64_t lane_cap = (uint64_t)core + warmup + 16;
    /* the lanes' loops are independent: ZB_BLK of them at a time, shared out over the OpenMP threads when more than one is
     * set (a single-channel capture; the wideband receivers run one channel per thread instead), then stitched in order */
    const uint64_t ZB_BLK = lane_cap >= (1u << 20) ? 1u : ((1u << 22) / lane_cap ? (1u << 22) / lane_cap : 1u);
    const uint64_t blk_lanes = n_lanes < ZB_BLK ? (n_lanes ? n_lanes : 1) : ZB_BLK;
    uint8_t* LB = (uint8_t*)malloc(lane_cap * blk_lanes);
    uint64_t* LPOS = (uint64_t*)malloc(lane_cap * blk_lanes * 8);
    uint64_t* LKEY = (uint64_t*)malloc(lane_cap * blk_lanes * 8);
    uint64_t* NC = (uint64_t*)malloc(blk_lanes * 8);
    uint8_t* sb = (uint8_t*)malloc(n + 16 + 4096);      /* stitched stream: at most one chip per sample */
    uint64_t* spos = (uint64_t*)malloc((n + 16 + 4096) * 8);
    uint64_t* o = (uint64_t*)malloc((n_lanes + 1) * 8); /* stream offset of every lane's first owned chip */
    mm_t* ends = (mm_t*)malloc((n_lanes + 1) * sizeof(mm_t)); /* every lane's loop at its core end */
    uint64_t* seam = (uint64_t*)malloc((n_lanes + 1) * 8); /* per lane: XOR of its 48 chips before the seam with lane l-1's last 48 (bit 0 = the last chip) */
    uint64_t total = 0;
    uint64_t E = 0, prev_nc = 0, prev_hist = 0;
    for (uint64_t l = 0; l < n_lanes; l++) {
        const uint64_t cs = l * core, ce = cs + core;
        if (l % blk_lanes == 0) {
            const uint64_t nb = n_lanes - l < blk_lanes ? n_lanes - l : blk_lanes;
#pragma omp parallel for schedule(dynamic, 4)
            for (uint64_t i = 0; i < nb; i++)
                NC[i] = mm_lane(d, n, (l + i) * core, core, warmup, lp_in[l + i], LB + i * lane_cap, LPOS + i * lane_cap,
                                LKEY + i * lane_cap, NULL, NULL, 0, &ends[l + i]);
        }
        const uint8_t* lb = LB + (l % blk_lanes) * lane_cap;
        const uint64_t* lpos = LPOS + (l % blk_lanes) * lane_cap;
        const uint64_t* lkey = LKEY + (l % blk_lanes) * lane_cap;
        const uint64_t nc = NC[l % blk_lanes];
        uint64_t f = 0;
        seam[l] = l ? 0xFFFFFFFFFFFFull : 0;                        /* no comparison made: nothing verified */
        if (l > 0) {
            uint64_t c0 = 0;
            while (c0 < nc && lpos[c0] + 3 < cs) c0++;              /* first candidate */
            uint64_t c1 = c0;
            while (c1 < nc && lpos[c1] <= cs + 5) c1++;             /* one past the last candidate */
            uint64_t f0 = c0;
            while (f0 < c1 && lkey[f0] < E) f0++;                   /* first candidate with T >= E */
            f = f0;
            if (prev_nc >= 48 && c1 > c0) {
                const uint64_t c_end = c1 - 1;
                uint64_t H = 0;                                     /* chips c_end-63 .. c_end, MSB first */
                for (uint64_t j = (c_end >= 63 ? c_end - 63 : 0); j <= c_end; j++) H = (H << 1) | lb[j];
                static const int shifts[3] = {0, -1, 1};
                int best = -1;
                for (int k = 0; k < 3; k++) {
                    const int64_t e = (int64_t)f0 + shifts[k] - 1;  /* chip aligned with lane l's last chip */
                    if (e < 47 || e > (int64_t)c_end) continue;
                    const uint64_t own = (H >> (c_end - (uint64_t)e)) & 0xFFFFFFFFFFFFull;
                    const int agree = 48 - __builtin_popcountll(own ^ (prev_hist & 0xFFFFFFFFFFFFull));
                    if (agree > best) { best = agree; f = (uint64_t)((int64_t)f0 + shifts[k]); seam[l] = own ^ (prev_hist & 0xFFFFFFFFFFFFull); }
                }
            }
        }
        o[l] = total;
        for (uint64_t j = f; j < nc; j++) { sb[total] = lb[j]; spos[total] = lpos[j]; total++; }
        E = nc ? lkey[nc - 1] + 128u : ce * 128u;
        prev_nc = nc;
        prev_hist = 0;                                              /* last 64 chips of the lane, MSB first */
        for (uint64_t j = (nc >= 64 ? nc - 64 : 0); j < nc; j++) prev_hist = (prev_hist << 1) | lb[j];
    }
    o[n_lanes] = total;
    if (g_dbg_bits) {
        g_dbg_n = total < g_dbg_cap ? total : g_dbg_cap;
        memcpy(g_dbg_bits, sb, g_dbg_n); memcpy(g_dbg_pos, spos, g_dbg_n * 8);
        g_dbg_lanes = n_lanes;
        if (g_dbg_first) memcpy(g_dbg_first, o, (n_lanes + 1) * 8);
    }
    /* The sinks of the lanes are independent (each reads the stitched stream; a frame repair reads the samples): they run
     * over the OpenMP threads when more than one is set and leave their candidate frames, in order, per lane; the
     * sequential rule (Resolve) then passes over the candidates in lane order. */
    typedef struct { uint64_t trigger, end_chip; snout_pkt pkt; } cand_t;
    cand_t** cands = (cand_t**)calloc(n_lanes ? n_lanes : 1, sizeof(cand_t*));
    uint32_t* ncand = (uint32_t*)calloc(n_lanes ? n_lanes : 1, sizeof(uint32_t));
#pragma omp parallel for schedule(dynamic, 16)
    for (uint64_t l = 0; l < n_lanes; l++) {
        lane_t s;
        memset(&s, 0, sizeof(s));
        enter_search(&s);
        uint64_t q_start = o[l] > ORACLE_ZB_SINK_WARM ? o[l] - ORACLE_ZB_SINK_WARM : 0;
        lane_t snap;
        int have_snap = 0;
        for (uint64_t q = q_start; q < total; q++) {
            if (s.state == 0 && s.preamble_cnt == 0 && q >= o[l + 1]) break;   /* idle past the lane */
            if (q == o[l + 1] && s.state != 0) { snap = s; have_snap = 1; }     /* busy with a synchronised frame at the seam */
            const int before = s.state;
            const int done = sink_chip(&s, sb[q] ? 1.0f : -1.0f, q, threshold);  /* trigger = chip index */
            const int gave_up = !done && before != 0 && s.state == 0 && s.chip_err;
            if (!done && !gave_up) continue;
            /* a frame belongs to the lane that owns the chip completing its SFD: sinks may first
             * match different preamble symbols, but they all find the SFD at the same chip */
            if (s.sync < o[l] || s.sync >= o[l + 1]) { if (done) enter_search(&s); continue; }
            lane_t rs;
            const lane_t* fin = &s;
            const uint64_t idx = spos[s.sync >= 319u ? s.sync - 319u : 0u];
            uint64_t end_chip = q;
            int crc_ok = 0;
            if (done && s.packetlen_cnt >= 3) {
                const uint16_t c = oracle_crc16_154(s.pkt, s.packetlen_cnt - 2);
                crc_ok = ((c & 0xFF) == s.pkt[s.packetlen_cnt - 2]) && ((c >> 8) == s.pkt[s.packetlen_cnt - 1]);
            }
            if (g_zb_repair && have_snap && q >= o[l + 1] && (gave_up || !crc_ok || g_zb_repair == 2)) {
                rs = snap;
                uint64_t rend = 0;
                /* lane l + 2 takes over again if it owns chips (hand_q > 0 then: it is not the channel's first lane) */
                const int hb = l + 2 < n_lanes && o[l + 2] < o[l + 3] && core <= (1u << 23);     /* (lane-relative keys in 32 bits) */
                const int ok = repair_frame(d, n, ends[l], o[l + 1], threshold, &rs, hb ? spos[o[l + 2]] : 0u, hb ? o[l + 2] : 0u,
                                            sb, total, &rend);
                if (!ok) { if (done) enter_search(&s); continue; }
                fin = &rs;
                end_chip = rend;
                crc_ok = 0;
                if (rs.packetlen_cnt >= 3) {
                    const uint16_t c = oracle_crc16_154(rs.pkt, rs.packetlen_cnt - 2);
                    crc_ok = ((c & 0xFF) == rs.pkt[rs.packetlen_cnt - 2]) && ((c >> 8) == rs.pkt[rs.packetlen_cnt - 1]);
                }
            } else if (gave_up) {
                continue;
            }
            if ((ncand[l] & (ncand[l] - 1u)) == 0u)        /* grow at 0, 1, 2, 4, 8 ... */
                cands[l] = (cand_t*)realloc(cands[l], (size_t)(ncand[l] ? 2u * ncand[l] : 1u) * sizeof(cand_t));
            cand_t* cd = &cands[l][ncand[l]++];
            cd->trigger = s.trigger;
            cd->end_chip = end_chip;
            snout_pkt* p = &cd->pkt;
            memset(p, 0, sizeof(*p));
            p->sample_index = first_index + idx;
            p->proto = 1;
            p->channel = (uint16_t)channel;
            p->len = (uint16_t)fin->packetlen_cnt;
            unsigned scaled = (fin->lqi / 8) << 3;
            p->lqi = (uint8_t)(scaled >= 256 ? 255 : scaled);
            p->aux = (uint32_t)l;
            /* SNOUT_PKT_ZB_SEAM_DISAGREED.  Seams inside the frame = hand-overs from one lane's timing loop to
             * the next between the trigger chip and the last chip.  The two loops ran side by side over the 48
             * chips before the seam (seam[m]: XOR of their decisions, bit 0 = the last chip before it); if they
             * decided any of those that belong to the frame differently, the frame's chips depend on which
             * loop is asked -- the one sequential loop's may differ too.  A repaired frame (SNOUT_PKT_ZB_REPAIRED)
             * has no seams. */
            if (fin == &rs) p->flags |= 8u;
            else {
                /* the seams o[m] in (trigger, q]: o[] ascends and the SFD chip is lane l's, so they are lanes
                 * l, l - 1, ... while o[m] > trigger and l + 1, l + 2, ... while o[m] <= q */
                uint64_t m0 = l;
                while (m0 > 1 && o[m0 - 1] > s.trigger) m0--;
                if (m0 < 1) m0 = 1;
                for (uint64_t m = m0; m < n_lanes && o[m] <= q; m++) {
                    if (o[m] <= s.trigger) continue;
                    const uint64_t inside = o[m] - s.trigger;       /* compared chips at or after the trigger */
                    const uint64_t mask = inside >= 48 ? 0xFFFFFFFFFFFFull : ((1ull << inside) - 1ull);
                    if (seam[m] & mask) p->flags |= 4u;
                }
            }
            memcpy(p->bytes, fin->pkt, (size_t)fin->packetlen_cnt);
            p->crc_ok = (uint8_t)crc_ok;
            if (done) enter_search(&s);
        }
    }
    /* resolve: the sequential sink is busy until the end of the frame it kept last */
    uint64_t busy_end = 0;          /* last chip of the frame kept last (sequential rule) */
    int have_kept = 0;
    for (uint64_t l = 0; l < n_lanes; l++) {
        for (uint32_t k = 0; k < ncand[l]; k++) {
            const cand_t* cd = &cands[l][k];
            if (have_kept && cd->trigger <= busy_end) continue;
            have_kept = 1;
            busy_end = cd->end_chip;
            if (*n_out < cap) out[*n_out] = cd->pkt;
            (*n_out)++;
        }
        free(cands[l]);
    }
    free(cands); free(ncand);
    free(ends); free(seam); free(o); free(spos); free(sb); free(NC); free(LKEY); free(LPOS); free(LB); free(lp_in);
}

int oracle_zigbee_segment(const float* iq, uint64_t n, uint64_t first_index, uint32_t channel,
                          uint32_t threshold, uint32_t core, uint32_t warmup,
                          snout_pkt* out, uint64_t cap, uint64_t* n_out)
{
    *n_out = 0;
    if (n < 9) return 0;
    if (core % 64u || warmup % 64u || warmup >= core) return -1;
    float* d = (float*)malloc(n * sizeof(float));
    if (!d) return -3;
    oracle_zb_discrim(iq, n, d);
    channel_lanes(d, n, first_index, channel, threshold, core, warmup, out, cap, n_out);
    free(d);
    return *n_out > cap ? -5 : 0;
}

/* Soft taps of lane `lane`: z (DC-removed, from the lane start) and the M&M output chips. */
int oracle_zigbee_lane_soft(const float* iq, uint64_t n, uint32_t core, uint32_t warmup,
                            uint32_t lane, uint32_t threshold, float* z, float* chips,
                            uint64_t cap, uint64_t* n_chips)
{
    (void)threshold;
    float* d = (float*)malloc((n ? n : 1) * sizeof(float));
    if (!d) return -3;
    oracle_zb_discrim(iq, n, d);
    const uint64_t n_lanes = (n + core - 1) / core;
    double* lp_in = oracle_zb_iir_carry(d, n, core, warmup, n_lanes);
    *n_chips = mm_lane(d, n, (uint64_t)lane * core, core, warmup, lane < n_lanes ? lp_in[lane] : 0.0,
                       NULL, NULL, NULL, z, chips, cap, NULL);
    free(lp_in);
    free(d);
    return 0;
}
