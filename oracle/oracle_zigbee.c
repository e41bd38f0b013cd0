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
 * One lane.  d = discriminator output of the whole segment (n samples).  Returns packets via
 * out/n_out (appending).  soft_z / soft_chips: optional taps (lane-relative, from the lane start).
 */
static void run_lane(const float* d, uint64_t n, uint64_t core_start, uint64_t core_len,
                     uint64_t warmup, unsigned th, uint32_t channel, uint64_t first_index,
                     uint32_t lane_id, snout_pkt* out, uint64_t cap, uint64_t* n_out,
                     float* soft_z, float* soft_chips, uint64_t soft_cap, uint64_t* n_chips,
                     double lp_init)
{
    const uint64_t s0 = core_start > warmup ? core_start - warmup : 0;
    const uint64_t core_end = core_start + core_len;
    const double alpha = 0.00016, one_minus = 1.0 - 0.00016;
    const float omega_mid = 2.0f, gain_omega = 0.000225f, gain_mu = 0.03f;
    const float omega_lim = omega_mid * 0.0002f;
    lane_t s;
    memset(&s, 0, sizeof(s));
    s.mu = 0.5f; s.omega = 2.0f; s.last = 0.0f;
    s.lp = lp_init;       /* IIR state carried in from the samples before the lane (see below) */
    enter_search(&s);
    /* z is produced lazily: z_have = number of lane-relative samples filtered so far */
    float win[8];
    uint64_t ii = s0;               /* absolute index of the interpolator window start */
    uint64_t z_next = s0;           /* next sample to run through the IIR */
    /* ring of filtered samples, big enough for the 8-tap window */
    float ring[16];
    uint64_t chips = 0;
    while (ii + 8 <= n) {
        while (z_next < ii + 8) {
            const float x = d[z_next];
            s.lp = alpha * (double)x + one_minus * s.lp;
            const float z = x - (float)s.lp;
            ring[z_next & 15] = z;
            if (soft_z && z_next - s0 < soft_cap) soft_z[z_next - s0] = z;
            z_next++;
        }
        for (int k = 0; k < 8; k++) win[k] = ring[(ii + k) & 15];
        const int imu = (int)rintf(s.mu * 128.0f);
        float acc = 0.0f;
        for (int k = 0; k < 8; k++) acc = fmaf(kMmseTaps[imu][k], win[7 - k], acc);
        const float o = acc;
        if (soft_chips && chips < soft_cap) soft_chips[chips] = o;
        chips++;
        const float mm = (s.last < 0.0f ? -1.0f : 1.0f) * o - (o < 0.0f ? -1.0f : 1.0f) * s.last;
        s.last = o;
        s.omega = s.omega + gain_omega * mm;
        {
            const float x = s.omega - omega_mid;
            const float c = 0.5f * (fabsf(x + omega_lim) - fabsf(x - omega_lim));
            s.omega = omega_mid + c;
        }
        s.mu = s.mu + s.omega + gain_mu * mm;
        const float fl = floorf(s.mu);
        const uint64_t at = ii;
        ii += (uint64_t)(int)fl;
        s.mu = s.mu - fl;

        const int was_idle = (s.state == 0 && s.preamble_cnt == 0);
        const int done = sink_chip(&s, o, at, th);
        if (was_idle && s.preamble_cnt == 1 && s.trigger >= core_end) break;   /* next lane's */
        if (done) {
            if (s.trigger >= core_start && s.trigger < core_end) {
                if (*n_out < cap) {
                    snout_pkt* p = &out[*n_out];
                    memset(p, 0, sizeof(*p));
                    p->sample_index = first_index + s.trigger;
                    p->proto = 1;
                    p->channel = (uint16_t)channel;
                    p->len = (uint16_t)s.packetlen_cnt;
                    unsigned scaled = (s.lqi / 8) << 3;
                    p->lqi = (uint8_t)(scaled >= 256 ? 255 : scaled);
                    p->aux = lane_id;
                    memcpy(p->bytes, s.pkt, (size_t)s.packetlen_cnt);
                    if (s.packetlen_cnt >= 3) {
                        uint16_t c = oracle_crc16_154(s.pkt, s.packetlen_cnt - 2);
                        p->crc_ok = (uint8_t)(((c & 0xFF) == s.pkt[s.packetlen_cnt - 2]) &&
                                              ((c >> 8) == s.pkt[s.packetlen_cnt - 1]));
                    }
                }
                (*n_out)++;
            }
            enter_search(&s);
        }
        if (s.state == 0 && s.preamble_cnt == 0 && ii >= core_end) break;
    }
    if (n_chips) *n_chips = chips;
}

/*
 * IIR carry-in.  GNU Radio's single-pole IIR runs over the continuous stream, so when a frame
 * arrives its DC estimate has long converged (time constant 1/alpha = 6250 samples).  A lane that
 * started the recurrence from zero would see the uncorrected CFO offset for its first thousands of
 * samples.  The linear recurrence is therefore carried across lanes exactly as a blocked evaluation
 * of the same filter (all in double, fixed operation order, the GPU does the same):
 *   S_j   = sum over the 64 samples of sub-block j of  w[63-k] * d[64 j + k],  w[m] = alpha (1-alpha)^m,
 *           terms summed pairwise: v[i] += v[i+off] for off = 32,16,8,4,2,1   (zero-state response)
 *   L_i   = fold over the sub-blocks of lane block i:  L = D64 * L + S_j
 *   lp_in[i+1] = Dblk_i * lp_in[i] + L_i ,  lp_in[0] = 0
 * where lane block i = [s0_i, s0_{i+1}), s0_i = max(0, i core - warmup), D64 = (1-alpha)^64 and
 * Dblk_i = D64^(sub-blocks of the block), powers formed by repeated multiplication.
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
        double v[64];
        for (int k = 0; k < 64; k++) {
            const uint64_t t = 64 * j + (uint64_t)k;
            v[k] = w[63 - k] * (double)(t < n ? d[t] : 0.0f);
        }
        for (int off = 32; off >= 1; off >>= 1)
            for (int i = 0; i < off; i++) v[i] = v[i] + v[i + off];
        S[j] = v[0];
    }
    double* lp_in = (double*)malloc((n_lanes ? n_lanes : 1) * sizeof(double));
    const double dcore = pow_rep(d64, core / 64u);
    const double dfirst = pow_rep(d64, core > warmup ? (core - warmup) / 64u : 0u);
    double lp = 0.0;
    for (uint64_t l = 0; l < n_lanes; l++) {
        lp_in[l] = lp;
        const uint64_t b0 = l == 0 ? 0 : ((uint64_t)l * core - warmup) / 64u;      /* first sub-block */
        const uint64_t b1 = ((uint64_t)(l + 1) * core - warmup) / 64u;             /* one past last   */
        double L = 0.0;
        for (uint64_t j = b0; j < b1; j++) L = d64 * L + (j < nsb ? S[j] : 0.0);
        lp = (l == 0 ? dfirst : dcore) * lp + L;
    }
    free(S);
    return lp_in;
}

int oracle_zigbee_segment(const float* iq, uint64_t n, uint64_t first_index, uint32_t channel,
                          uint32_t threshold, uint32_t core, uint32_t warmup,
                          snout_pkt* out, uint64_t cap, uint64_t* n_out)
{
    *n_out = 0;
    if (n < 9) return 0;
    float* d = (float*)malloc(n * sizeof(float));
    if (!d) return -3;
    oracle_zb_discrim(iq, n, d);
    if (core % 64u || warmup % 64u || warmup >= core) { free(d); return -1; }
    const uint64_t n_lanes = (n + core - 1) / core;
    double* lp_in = oracle_zb_iir_carry(d, n, core, warmup, n_lanes);
    for (uint64_t l = 0; l < n_lanes; l++)
        run_lane(d, n, l * core, core, warmup, threshold, channel, first_index, (uint32_t)l, out,
                 cap, n_out, NULL, NULL, 0, NULL, lp_in[l]);
    free(lp_in);
    free(d);
    return *n_out > cap ? -5 : 0;
}

/* Soft taps of lane `lane`: z (DC-removed, from the lane start) and the M&M output chips. */
int oracle_zigbee_lane_soft(const float* iq, uint64_t n, uint32_t core, uint32_t warmup,
                            uint32_t lane, uint32_t threshold, float* z, float* chips,
                            uint64_t cap, uint64_t* n_chips)
{
    float* d = (float*)malloc((n ? n : 1) * sizeof(float));
    if (!d) return -3;
    oracle_zb_discrim(iq, n, d);
    snout_pkt tmp[64];
    uint64_t cnt = 0;
    const uint64_t n_lanes = (n + core - 1) / core;
    double* lp_in = oracle_zb_iir_carry(d, n, core, warmup, n_lanes);
    run_lane(d, n, (uint64_t)lane * core, core, warmup, threshold, 0, 0, lane, tmp, 64, &cnt, z, chips,
             cap, n_chips, lane < n_lanes ? lp_in[lane] : 0.0);
    free(lp_in);
    free(d);
    return 0;
}
