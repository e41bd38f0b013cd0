/*
 * oracle_btle.c — CPU restatement of the BTLE receive path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (libsnout_rx.so) never does.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the third-party program `btle_rx`
 * (jkulskis/BTLE fork of JiaoXianjun/BTLE, commit unpinned, submodule vendor/BTLE is EMPTY in
 * /root/reference; .gitmodules:1-6, Makefile:45-49).  The reference holds no golden vector for
 * it.  This file restates the published algorithm and is anchored on the
 * reference's call site and consumer only:
 *   - invocation  snout/util/btle.py:63-68   (-c CH -g 6 -a 8e89bed6 -k 555555)
 *   - output line snout/core/message.py:214-215, parsed at :226-236
 *
 * Algorithm (per capture segment, state reset at segment start):
 *   bit[n]  = (I[n]*Q[n+4]) > (I[n+4]*Q[n])           4 samples/symbol, each product rounded to
 *                                                     f32 separately (no FMA contraction)
 *   w[n]    = last 32 bits of phase n%4 ending at n, oldest bit in the LSB
 *   hit     = smallest n >= resume+124 with w[n]==AA  (AA is sent LSB first)
 *   header  = 2 bytes at stride 4 after the hit, LSB first, de-whitened (x^7+x^4+1, seed ch|0x40)
 *   reject  payload length outside [6,37]  -> resume after the header
 *   payload = len+3 bytes, de-whitened; CRC24 poly 0x00065B init crc_init over header+payload
 *   resume  = first sample after the CRC
 * Deviation from upstream, stated: a packet whose payload does not fit in the segment is skipped
 * (search resumes after its header) instead of ending the search; upstream waits for more samples
 * from its ring buffer there.  Overlapping segments cover such packets.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "oracle.h"

/* ---- whitening: 7-bit LFSR x^7+x^4+1; position 0 = 1, positions 1..6 = channel MSB..LSB ---- */
void oracle_btle_whiten_seq(uint32_t channel, uint8_t* out, int nbytes)
{
    uint8_t p[7];
    p[0] = 1;
    for (int i = 0; i < 6; i++) p[1 + i] = (channel >> (5 - i)) & 1;
    for (int b = 0; b < nbytes; b++) {
        uint8_t v = 0;
        for (int k = 0; k < 8; k++) {
            uint8_t o = p[6];
            v |= (uint8_t)(o << k);                 /* LSB first */
            uint8_t n4 = p[3] ^ o;
            p[6] = p[5]; p[5] = p[4]; p[4] = n4; p[3] = p[2]; p[2] = p[1]; p[1] = p[0]; p[0] = o;
        }
        out[b] = v;
    }
}

/* ---- CRC24, bitwise: register bit 23 is sent first; data bits enter LSB first -------------- */
uint32_t oracle_btle_crc24(const uint8_t* data, int n, uint32_t init)
{
    uint32_t r = init & 0xFFFFFFu;
    for (int i = 0; i < n; i++) {
        uint8_t d = data[i];
        for (int k = 0; k < 8; k++, d >>= 1) {
            uint32_t t = (r >> 23) & 1u;
            r = (r << 1) & 0xFFFFFFu;
            if (t != (uint32_t)(d & 1)) r ^= 0x00065Bu;
        }
    }
    return r;
}

/* The three CRC bytes as they appear in the LSB-first packed byte stream. */
void oracle_btle_crc_bytes(uint32_t r, uint8_t out[3])
{
    for (int b = 0; b < 3; b++) {
        uint8_t v = 0;
        for (int k = 0; k < 8; k++) v |= (uint8_t)(((r >> (23 - (8 * b + k))) & 1u) << k);
        out[b] = v;
    }
}

/* Second, table-driven CRC (reflected form) used to cross-check the bitwise one in tests. */
uint32_t oracle_btle_crc24_table(const uint8_t* data, int n, uint32_t init)
{
    /* reflected register: bit i of rr = bit (23-i) of r */
    static uint32_t tab[256];
    static int ready = 0;
    if (!ready) {
        /* reflected polynomial of 0x00065B over 24 bits = 0xDA6000 */
        for (int i = 0; i < 256; i++) {
            uint32_t c = (uint32_t)i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? ((c >> 1) ^ 0xDA6000u) : (c >> 1);
            tab[i] = c;
        }
        ready = 1;
    }
    uint32_t rr = 0;
    for (int i = 0; i < 24; i++) rr |= ((init >> (23 - i)) & 1u) << i;
    for (int i = 0; i < n; i++) rr = (rr >> 8) ^ tab[(rr ^ data[i]) & 0xFFu];
    uint32_t r = 0;
    for (int i = 0; i < 24; i++) r |= ((rr >> i) & 1u) << (23 - i);
    return r;
}

/* ---- a1: hard bits -------------------------------------------------------------------------- */
/* bits[n] for n in [0, n_samples-4); returns number of bits written. */
uint64_t oracle_btle_bits(const float* iq, uint64_t n_samples, uint8_t* bits)
{
    if (n_samples < 5) return 0;
    uint64_t nb = n_samples - 4;
    for (uint64_t n = 0; n < nb; n++) {
        float a = iq[2 * n] * iq[2 * (n + 4) + 1];       /* I0*Q1 */
        float b = iq[2 * (n + 4)] * iq[2 * n + 1];       /* I1*Q0 */
        bits[n] = (uint8_t)(a > b);
    }
    return nb;
}

static inline uint8_t demod_byte(const uint8_t* bits, uint64_t at)
{
    uint8_t v = 0;
    for (int k = 0; k < 8; k++) v |= (uint8_t)(bits[at + 4u * (uint64_t)k] << k);
    return v;
}

/* every n in [124, nb) whose phase word equals the access address, ascending */
uint64_t oracle_btle_all_hits(const uint8_t* bits, uint64_t nb, uint32_t access_addr,
                              uint64_t* hits, uint64_t cap)
{
    uint32_t w[4] = {0, 0, 0, 0};
    uint64_t cnt = 0;
    for (uint64_t n = 0; n < nb; n++) {
        unsigned j = (unsigned)(n & 3u);
        w[j] = (w[j] >> 1) | ((uint32_t)bits[n] << 31);
        if (n >= 124 && w[j] == access_addr) {
            if (cnt < cap) hits[cnt] = n;
            cnt++;
        }
    }
    return cnt;
}

/* ---- a1+a2: sequential search / decode over one segment ------------------------------------- */
int oracle_btle_segment(const float* iq, uint64_t n_samples, uint64_t first_sample_index,
                        uint32_t channel, uint32_t access_addr, uint32_t crc_init,
                        snout_pkt* out, uint64_t cap, uint64_t* n_out,
                        uint64_t* hits_out, uint64_t hits_cap, uint64_t* n_hits_out)
{
    /* hits_out receives the ACCEPTED-or-examined hits of the sequential chain (the ones the
     * search actually stopped at); oracle_btle_all_hits() lists every match of every phase. */
    *n_out = 0;
    if (n_hits_out) *n_hits_out = 0;
    if (n_samples < 5) return 0;
    uint8_t* bits = (uint8_t*)malloc(n_samples);
    if (!bits) return -3;
    uint64_t nb = oracle_btle_bits(iq, n_samples, bits);
    uint8_t wh[42];
    oracle_btle_whiten_seq(channel, wh, 42);

    /* per-phase running 32-bit words */
    uint32_t w[4] = {0, 0, 0, 0};
    uint64_t resume = 0;            /* first sample the current search may use */
    uint64_t n = 0, n_found = 0, n_hits = 0;
    /* (re)start: words must be rebuilt from `resume`; track how many bits each phase has */
    uint32_t have[4] = {0, 0, 0, 0};
    while (n < nb) {
        unsigned j = (unsigned)(n & 3u);
        w[j] = (w[j] >> 1) | ((uint32_t)bits[n] << 31);
        if (have[j] < 32) have[j]++;
        if (have[j] == 32 && w[j] == access_addr) {
            /* every hit of every phase is reported to hits_out (a1 tap), accepted or not */
            if (hits_out && n_hits < hits_cap) hits_out[n_hits] = n;
            n_hits++;
            uint64_t hdr = n + 4;
            uint64_t next = 0;
            if (hdr + 4u * 15u < nb) {
                uint8_t rec[42];
                rec[0] = demod_byte(bits, hdr) ^ wh[0];
                rec[1] = demod_byte(bits, hdr + 32) ^ wh[1];
                unsigned plen = rec[1] & 0x3Fu;
                next = hdr + 64;
                if (plen >= 6 && plen <= 37) {
                    unsigned total = 2 + plen + 3;
                    if (hdr + 4u * (8u * (uint64_t)total - 1u) < nb) {
                        for (unsigned b = 2; b < total; b++)
                            rec[b] = demod_byte(bits, hdr + 32u * (uint64_t)b) ^ wh[b];
                        uint32_t crc = oracle_btle_crc24(rec, 2 + (int)plen, crc_init);
                        uint8_t cb[3];
                        oracle_btle_crc_bytes(crc, cb);
                        int ok = cb[0] == rec[2 + plen] && cb[1] == rec[3 + plen] && cb[2] == rec[4 + plen];
                        if (n_found < cap) {
                            snout_pkt* p = &out[n_found];
                            memset(p, 0, sizeof(*p));
                            p->sample_index = first_sample_index + (n - 124);
                            p->proto = 0;
                            p->channel = (uint16_t)channel;
                            p->len = (uint16_t)total;
                            p->crc_ok = (uint8_t)ok;
                            p->pdu_type = rec[0] & 0x0F;
                            p->flags = (uint8_t)(((rec[0] >> 6) & 1) | (((rec[0] >> 7) & 1) << 1));
                            p->aux = j;
                            memcpy(p->bytes, rec, total);
                        }
                        n_found++;
                        next = hdr + 32u * (uint64_t)total;
                    }
                }
            } else {
                next = n + 1;   /* header does not fit: nothing later fits either */
            }
            /* restart the search at `next`: all four phase words start empty there */
            resume = next;
            n = resume;
            have[0] = have[1] = have[2] = have[3] = 0;
            w[0] = w[1] = w[2] = w[3] = 0;
            continue;
        }
        n++;
    }
    free(bits);
    *n_out = n_found;
    if (n_hits_out) *n_hits_out = n_hits;
    (void)resume;
    return n_found > cap ? -5 : 0;
}
