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
    float pr = 0.0f, pi = 0.0f;
    for (uint64_t t = 0; t < n; t++) {
        const float ar = iq[2 * t], ai = iq[2 * t + 1];
        const float re = ar * pr + ai * pi;
        const float im = ai * pr - ar * pi;
        float a = oracle_fast_atan2f(im, re);
        if (!(fabsf(a) <= 4.0f)) a = 0.0f;      /* non-finite input: defined as 0 (stated deviation) */
        d[t] = a;
        pr = ar;
        pi = ai;
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
    uint8_t pkt[128];
} lane_t;

static inline unsigned popc(uint32_t v) { return (unsigned)__builtin_popcount(v); }

static void enter_search(lane_t* s)
{
    s->state = 0; s->shift = 0; s->preamble_cnt = 0; s->chip_cnt = 0; s->packet_byte = 0;
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
            if (c == 0xFF) { enter_search(s); return 0; }
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
            if (c == 0xFF) { enter_search(s); return 0; }
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

typedef struct {
    uint64_t n_chips;        /* chips produced by the lane's M&M */
    uint64_t first_owned;    /* index of the first owned chip (<= n_chips) */
    uint64_t t_last;         /* key of the last chip (valid if n_chips) */
} lane_chips_t;

/* ANALYSIS SWITCH ORACLE_ZB_EXPERIMENT_TWOSTART (tools/lane_gaps_r4.py only): samples by which the next mm_lane call starts its
 * loop later (0 or 1); per thread, channel_lanes runs one channel per thread */
static _Thread_local int g_exp_start_shift = 0;

/* a5 + a6 of one lane; appends (bit, pos) of ALL its chips to bits/pos/key (capacity ensured by the
 * caller: (core + warmup) chips at most).  soft_z / soft_chips: optional taps. */
static uint64_t mm_lane(const float* d, uint64_t n, uint64_t core_start, uint64_t core_len,
                        uint64_t warmup, double lp_init, uint8_t* bits, uint64_t* pos, uint64_t* key,
                        float* soft_z, float* soft_chips, uint64_t soft_cap)
{
    const uint64_t s0 = core_start > warmup ? core_start - warmup : 0;
    const uint64_t core_end = core_start + core_len;
    const double alpha = 0.00016, one_minus = 1.0 - 0.00016;
    const float omega_mid = 2.0f, gain_omega = 0.000225f, gain_mu = 0.03f;
    const float omega_lim = omega_mid * 0.0002f;
    double lp = lp_init;
    float mu = 0.5f, omega = 2.0f, last = 0.0f;
    float win[8], ring[16];
    uint64_t ii = s0, z_next = s0, chips = 0;
    float warm_gain = 1.0f;
    { const char* e = s0 ? getenv("ORACLE_ZB_EXPERIMENT_WARMGAIN") : NULL; if (e) warm_gain = (float)atof(e); }
    if (s0 && g_exp_start_shift) ii += (uint64_t)g_exp_start_shift;     /* ORACLE_ZB_EXPERIMENT_TWOSTART, see channel_lanes */
    {   /* ANALYSIS SWITCH (tools/lane_residual_r4.py only): start the lane's loop this many quarter samples later */
        const char* e = s0 ? getenv("ORACLE_ZB_EXPERIMENT_PHASE0") : NULL;
        if (e) { const int q = atoi(e); ii += (uint64_t)(q / 4); mu = 0.5f + 0.25f * (float)(q % 4); if (mu >= 1.0f) { mu -= 1.0f; ii++; } }
    }
    while (ii < core_end && ii + 8 <= n) {
        while (z_next < ii + 8 && z_next < n) {
            const float x = d[z_next];
            lp = alpha * (double)x + one_minus * lp;
            const float z = x - (float)lp;
            ring[z_next & 15] = z;
            if (soft_z && z_next - s0 < soft_cap) soft_z[z_next - s0] = z;
            z_next++;
        }
        for (int k = 0; k < 8; k++) win[k] = ring[(ii + k) & 15];
        const int imu = (int)rintf(mu * 128.0f);
        float acc = 0.0f;
        for (int k = 0; k < 8; k++) acc = fmaf(kMmseTaps[imu][k], win[7 - k], acc);
        const float o = acc;
        if (soft_chips && chips < soft_cap) soft_chips[chips] = o;
        if (bits) { bits[chips] = o > 0.0f; pos[chips] = ii; key[chips] = ii * 128u + (uint64_t)imu; }
        chips++;
        const float mm = (last < 0.0f ? -1.0f : 1.0f) * o - (o < 0.0f ? -1.0f : 1.0f) * last;
        last = o;
        omega = omega + gain_omega * mm;
        {
            const float x = omega - omega_mid;
            const float c = 0.5f * (fabsf(x + omega_lim) - fabsf(x - omega_lim));
            omega = omega_mid + c;
        }
        mu = mu + omega + (ii < core_start ? gain_mu * warm_gain : gain_mu) * mm;
        const float fl = floorf(mu);
        ii += fl >= 1.0f ? (uint64_t)(int)fl : 1u;     /* 1..3 for finite input */
        mu = mu - fl;
    }
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
    for (uint64_t l = 0; l < n_lanes; l++) {
        const uint64_t b0 = l == 0 ? 0 : ((uint64_t)l * core - warmup) / 64u;      /* first sub-block */
        const uint64_t b1 = ((uint64_t)(l + 1) * core - warmup) / 64u;             /* one past last   */
        double a = 0.0;
        for (uint64_t j = b0; j < b1; j++) a = d64 * a + (j < nsb ? S[j] : 0.0);
        L[l] = a;
    }
    const uint64_t W = ((1u << 18) + core - 1) / core;
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

/* One channel: lanes -> stitched chip stream -> lane-local sinks. */
static void channel_lanes(const float* d, uint64_t n, uint64_t first_index, uint32_t channel,
                          uint32_t threshold, uint32_t core, uint32_t warmup,
                          snout_pkt* out, uint64_t cap, uint64_t* n_out)
{
    const uint64_t n_lanes = (n + core - 1) / core;
    double* lp_in = oracle_zb_iir_carry(d, n, core, warmup, n_lanes);
    const uint64_t lane_cap = (uint64_t)core + warmup + 16;
    uint8_t* lb = (uint8_t*)malloc(lane_cap);
    uint64_t* lpos = (uint64_t*)malloc(lane_cap * 8);
    uint64_t* lkey = (uint64_t*)malloc(lane_cap * 8);
    uint8_t* sb = (uint8_t*)malloc(n + 16 + 4096);      /* stitched stream: at most one chip per sample */
    uint64_t* spos = (uint64_t*)malloc((n + 16 + 4096) * 8);
    uint64_t* o = (uint64_t*)malloc((n_lanes + 1) * 8); /* stream offset of every lane's first owned chip */
    uint64_t* seam = (uint64_t*)malloc((n_lanes + 1) * 8); /* per lane: XOR of its 48 chips before the seam with lane l-1's last 48 (bit 0 = the last chip) */
    uint64_t total = 0;
    uint64_t E = 0, prev_nc = 0, prev_hist = 0;
    /* ANALYSIS SWITCH (tools/lane_residual.py only): verified hand-over.  Every lane runs P samples past its core end;
     * the next lane takes over at the first chip from which R consecutive chips of both loops agree in value and in
     * time (keys within half a sample), or at the end of the post-roll if they never do. */
    const char* pr_env = getenv("ORACLE_ZB_EXPERIMENT_POSTROLL");
    if (pr_env != NULL && atoi(pr_env) > 0) {
        const uint64_t P = (uint64_t)atoi(pr_env);
        const uint64_t R = getenv("ORACLE_ZB_EXPERIMENT_RUN") ? (uint64_t)atoi(getenv("ORACLE_ZB_EXPERIMENT_RUN")) : 48u;
        const uint64_t cap2 = lane_cap + P;
        uint8_t* pb = (uint8_t*)malloc(cap2); uint64_t* ppos = (uint64_t*)malloc(cap2 * 8); uint64_t* pkey = (uint64_t*)malloc(cap2 * 8);
        uint8_t* cb = (uint8_t*)malloc(cap2); uint64_t* cpos = (uint64_t*)malloc(cap2 * 8); uint64_t* ckey = (uint64_t*)malloc(cap2 * 8);
        uint64_t pnc = 0, pfrom = 0;
        for (uint64_t l = 0; l < n_lanes; l++) {
            const uint64_t cs = l * core;
            const uint64_t nc = mm_lane(d, n, cs, core + P, warmup, lp_in[l], cb, cpos, ckey, NULL, NULL, 0);
            uint64_t cfrom = 0;
            seam[l] = 0;
            if (l > 0) {
                uint64_t c0 = 0;
                while (c0 < nc && cpos[c0] + 3 < cs) c0++;
                uint64_t i = pfrom, run = 0, last_i = 0, pto = pnc;
                int found = 0;
                for (uint64_t j = c0; j < nc; j++) {
                    while (i < pnc && pkey[i] + 64 < ckey[j]) i++;
                    if (i >= pnc) break;
                    const int ok = pkey[i] <= ckey[j] + 64 && pb[i] == cb[j];
                    if (ok && run > 0 && i == last_i + 1) run++; else run = ok ? 1 : 0;
                    last_i = i;
                    if (run >= R) { found = 1; pto = i + 1; cfrom = j + 1; break; }
                }
                if (!found) {
                    seam[l] = 0xFFFFFFFFFFFFull;
                    const uint64_t Ee = pnc ? pkey[pnc - 1] + 128u : cs * 128u;
                    cfrom = c0;
                    while (cfrom < nc && ckey[cfrom] < Ee) cfrom++;
                }
                o[l - 1] = total;
                for (uint64_t j = pfrom; j < pto; j++) { sb[total] = pb[j]; spos[total] = ppos[j]; total++; }
            }
            { uint8_t* t8 = pb; pb = cb; cb = t8; uint64_t* t = ppos; ppos = cpos; cpos = t; t = pkey; pkey = ckey; ckey = t; }
            pnc = nc; pfrom = cfrom;
        }
        if (n_lanes) { o[n_lanes - 1] = total; for (uint64_t j = pfrom; j < pnc; j++) { sb[total] = pb[j]; spos[total] = ppos[j]; total++; } }
        free(pb); free(ppos); free(pkey); free(cb); free(cpos); free(ckey);
    } else {
    /* ANALYSIS SWITCH (tools/lane_gaps_r4.py only; never set in tests or by the product's parity runs): lane boundaries in the
     * gaps between frames.  b[l] = the sample nearest l core (within half a core) at which the discriminator output has looked
     * like noise for G samples (mean |d| over them above 1.2: a frame's MSK gives 0.8, noise pi/2, a neighbour's leakage more);
     * no such sample: l core.  A lane starts its loop `warmup` samples before b[l] with the sequential filter's own state. */
    uint64_t* bnd = NULL;
    double* lp_seq = NULL;
    const char* gap_env = getenv("ORACLE_ZB_EXPERIMENT_GAPS");
    if (gap_env != NULL && atoi(gap_env) > 0 && n_lanes > 1) {
        const uint64_t G = (uint64_t)atoi(gap_env);
        bnd = (uint64_t*)malloc((n_lanes + 1) * 8);
        lp_seq = (double*)malloc(n_lanes * 8);
        uint8_t* gap = (uint8_t*)calloc(n + 1, 1);              /* gap[t]: d[t-G+1 .. t] is noise */
        double acc = 0.0;
        for (uint64_t t = 0; t < n; t++) {
            acc += fabs((double)d[t]);
            if (t >= G) acc -= fabs((double)d[t - G]);
            gap[t] = t + 1 >= G && acc > 1.2 * (double)G;
        }
        bnd[0] = 0; bnd[n_lanes] = n;
        uint64_t moved = 0;
        for (uint64_t l = 1; l < n_lanes; l++) {
            const uint64_t c = l * core;
            uint64_t best = c;
            for (uint64_t k = 0; k < core / 2; k++) {
                if (c + k < n && gap[c + k]) { best = c + k; break; }
                if (c >= k + 1 && c - k > bnd[l - 1] + 64 && gap[c - k]) { best = c - k; break; }
            }
            if (best <= bnd[l - 1] + 64) best = c > bnd[l - 1] + 64 ? c : bnd[l - 1] + 64;
            if (best > n) best = n;
            moved += best != c;
            bnd[l] = best;
        }
        free(gap);
        const double alpha = 0.00016, one_minus = 1.0 - 0.00016;
        double lp = 0.0;
        uint64_t l = 1;
        lp_seq[0] = 0.0;
        for (uint64_t t = 0; t < n && l < n_lanes; t++) {
            while (l < n_lanes && (bnd[l] > warmup ? bnd[l] - warmup : 0) == t) lp_seq[l++] = lp;
            lp = alpha * (double)d[t] + one_minus * lp;
        }
        while (l < n_lanes) lp_seq[l++] = lp;
        if (getenv("ORACLE_ZB_EXPERIMENT_GAPS_VERBOSE")) fprintf(stderr, "gaps: %llu of %llu boundaries moved\n", (unsigned long long)moved, (unsigned long long)(n_lanes - 1));
        free(lb); free(lpos); free(lkey);
        const uint64_t cap2 = 2 * (uint64_t)core + warmup + 80;
        lb = (uint8_t*)malloc(cap2); lpos = (uint64_t*)malloc(cap2 * 8); lkey = (uint64_t*)malloc(cap2 * 8);
    }
    for (uint64_t l = 0; l < n_lanes; l++) {
        const uint64_t cs = bnd ? bnd[l] : l * core, ce = bnd ? bnd[l + 1] : cs + core;
        if (bnd && ce <= cs) { o[l] = total; seam[l] = 0; continue; }
        /* ANALYSIS SWITCH: the timing loop of a lane that starts inside a frame can hang at the half-chip point for hundreds of
         * chips (Mueller & Mueller's hang-up).  Run the warm-up from two starts one sample (half a chip) apart and keep the one
         * with the wider eye: sum |interpolated sample| over the last E chips before the core. */
        if (l > 0 && getenv("ORACLE_ZB_EXPERIMENT_TWOSTART") != NULL) {
            const uint64_t Ew = (uint64_t)atoi(getenv("ORACLE_ZB_EXPERIMENT_TWOSTART"));
            float* sc = (float*)malloc((warmup + 64) * sizeof(float));
            double eye[2] = {0.0, 0.0};
            for (int c = 0; c < 2; c++) {
                g_exp_start_shift = c;
                const uint64_t s0w = cs > warmup ? cs - warmup : 0;
                const uint64_t k = mm_lane(d, n, s0w, cs - s0w, 0, bnd ? lp_seq[l] : lp_in[l], NULL, NULL, NULL, NULL, sc, warmup + 64);
                for (uint64_t j = k > Ew ? k - Ew : 0; j < k; j++) eye[c] += fabs((double)sc[j]);
            }
            free(sc);
            g_exp_start_shift = eye[1] > eye[0] ? 1 : 0;
        }
        const uint64_t nc = mm_lane(d, n, cs, ce - cs, warmup, bnd ? lp_seq[l] : lp_in[l], lb, lpos, lkey, NULL, NULL, 0);
        g_exp_start_shift = 0;
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
    free(bnd); free(lp_seq);
    }
    o[n_lanes] = total;
    if (g_dbg_bits) {
        g_dbg_n = total < g_dbg_cap ? total : g_dbg_cap;
        memcpy(g_dbg_bits, sb, g_dbg_n); memcpy(g_dbg_pos, spos, g_dbg_n * 8);
        g_dbg_lanes = n_lanes;
        if (g_dbg_first) memcpy(g_dbg_first, o, (n_lanes + 1) * 8);
    }
    uint64_t busy_end = 0;          /* last chip of the frame kept last (sequential rule) */
    int have_kept = 0;
    for (uint64_t l = 0; l < n_lanes; l++) {
        lane_t s;
        memset(&s, 0, sizeof(s));
        enter_search(&s);
        uint64_t q_start = o[l] > ORACLE_ZB_SINK_WARM ? o[l] - ORACLE_ZB_SINK_WARM : 0;
        /* ANALYSIS SWITCH (tools/lane_residual.py; never set in tests or by the product's parity runs): the sequential
         * sink is busy with the frame kept last until its last chip and searches again, register cleared, from the
         * next one; a lane's sink that starts inside that frame can false-sync on payload symbols, and a phantom's
         * bogus length can run over the next real frame.  With the switch the lane's sink starts behind the kept
         * frame instead: what that rule would recover of the lanes' residual. */
        if (getenv("ORACLE_ZB_EXPERIMENT_RESTART") != NULL && have_kept && q_start <= busy_end) q_start = busy_end + 1;
        for (uint64_t q = q_start; q < total; q++) {
            if (s.state == 0 && s.preamble_cnt == 0 && q >= o[l + 1]) break;   /* idle past the lane */
            const int done = sink_chip(&s, sb[q] ? 1.0f : -1.0f, q, threshold);  /* trigger = chip index */
            /* a frame belongs to the lane that owns the chip completing its SFD: sinks may first
             * match different preamble symbols, but they all find the SFD at the same chip */
            if (done && (s.sync < o[l] || s.sync >= o[l + 1])) { enter_search(&s); continue; }
            if (done) {
                /* resolve: the sequential sink is busy until the end of the frame it kept last */
                if (have_kept && s.trigger <= busy_end) { enter_search(&s); continue; }
                have_kept = 1;
                busy_end = q;
                if (*n_out < cap) {
                    snout_pkt* p = &out[*n_out];
                    memset(p, 0, sizeof(*p));
                    p->sample_index = first_index + spos[s.sync >= 319u ? s.sync - 319u : 0u];
                    p->proto = 1;
                    p->channel = (uint16_t)channel;
                    p->len = (uint16_t)s.packetlen_cnt;
                    unsigned scaled = (s.lqi / 8) << 3;
                    p->lqi = (uint8_t)(scaled >= 256 ? 255 : scaled);
                    p->aux = (uint32_t)l;
                    /* SNOUT_PKT_ZB_SEAM_DISAGREED.  Seams inside the frame = hand-overs from one lane's timing loop to
                     * the next between the trigger chip and the last chip.  The two loops ran side by side over the 48
                     * chips before the seam (seam[m]: XOR of their decisions, bit 0 = the last chip before it); if they
                     * decided any of those that belong to the frame differently, the frame's chips depend on which
                     * loop is asked -- the one sequential loop's may differ too. */
                    for (uint64_t m = 1; m < n_lanes; m++) {
                        if (o[m] <= s.trigger || o[m] > q) continue;
                        const uint64_t inside = o[m] - s.trigger;       /* compared chips at or after the trigger */
                        const uint64_t mask = inside >= 48 ? 0xFFFFFFFFFFFFull : ((1ull << inside) - 1ull);
                        if (seam[m] & mask) p->flags |= 4u;
                    }
                    memcpy(p->bytes, s.pkt, (size_t)s.packetlen_cnt);
                    if (s.packetlen_cnt >= 3) {
                        uint16_t c = oracle_crc16_154(s.pkt, s.packetlen_cnt - 2);
                        p->crc_ok = (uint8_t)(((c & 0xFF) == s.pkt[s.packetlen_cnt - 2]) &&
                                              ((c >> 8) == s.pkt[s.packetlen_cnt - 1]));
                    }
                }
                (*n_out)++;
                enter_search(&s);
            }
        }
    }
    free(seam); free(o); free(spos); free(sb); free(lkey); free(lpos); free(lb); free(lp_in);
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
                       NULL, NULL, NULL, z, chips, cap);
    free(lp_in);
    free(d);
    return 0;
}
