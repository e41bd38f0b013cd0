// formats.cpp — host-side consumer contracts of the receive path (no GPU involved).
//
//  * snout_btle_format_line : the stdout grammar of `btle_rx` that the reference parses at
//    snout/core/message.py:205-237 (example line at :214) — 11 space-separated tokens, the last
//    being "CRC0\n" for a good CRC (:226).
//  * snout_rftap_encap      : what rftap.rftap_encap(2, 195, '') emits for a PDU whose meta holds
//    {lqi, qual=lqi/255.0} (top_block.py:53,80-83; epy_block_0.py:20-24): RFtap header with the
//    DLT and QUAL fields, then the MPDU.  One UDP datagram each (socket_pdu, top_block.py:71).
//  * channel plans.
#include "common.h"
#include <stdio.h>
#include <math.h>

static const char* kAdvPduName[16] = {
    "ADV_IND", "ADV_DIRECT_IND", "ADV_NONCONN_IND", "SCAN_REQ", "SCAN_RSP", "CONNECT_REQ",
    "ADV_SCAN_IND", "RESERVED0", "RESERVED1", "RESERVED2", "RESERVED3", "RESERVED4",
    "RESERVED5", "RESERVED6", "RESERVED7", "RESERVED8"};

static int put_hex_rev(char* d, size_t cap, size_t at, const uint8_t* b, int n)
{   // n bytes, most significant (last on air) first
    for (int i = n - 1; i >= 0; i--) {
        if (at + 2 >= cap) return -1;
        snprintf(d + at, 3, "%02x", b[i]);
        at += 2;
    }
    return (int)at;
}

static int put_hex(char* d, size_t cap, size_t at, const uint8_t* b, int n)
{
    for (int i = 0; i < n; i++) {
        if (at + 2 >= cap) return -1;
        snprintf(d + at, 3, "%02x", b[i]);
        at += 2;
    }
    return (int)at;
}

extern "C" {

int snout_btle_format_line(const snout_pkt* p, double fs_hz, double t0_epoch, uint32_t pkt_number,
                           uint32_t access_addr, char* dst, size_t cap)
{
    if (!p || !dst || cap < 64 || p->proto != SNOUT_PROTO_BTLE || p->len < 2 + 6 + 3 || fs_hz <= 0)
        return SNOUT_EINVAL;
    const double t = t0_epoch + (double)p->sample_index / fs_hz;
    const long long sec = (long long)floor(t);
    long long usec = (long long)floor((t - (double)sec) * 1e6 + 0.5);
    long long s2 = sec;
    if (usec >= 1000000) { usec -= 1000000; s2 += 1; }
    const int plen = p->len - 5;                       // payload length field
    const uint8_t* pay = p->bytes + 2;
    const int type = p->pdu_type & 0x0F;
    int at = snprintf(dst, cap, "%lld.%06lld Pkt%u Ch%u AA:%08x ADV_PDU_t%d:%s T%d R%d PloadL%d ",
                      s2, usec, pkt_number, (unsigned)p->channel, access_addr, type,
                      kAdvPduName[type], p->flags & 1, (p->flags >> 1) & 1, plen);
    if (at < 0 || (size_t)at >= cap) return SNOUT_EOVERFLOW;
    auto lit = [&](const char* s) -> bool {
        size_t n = strlen(s);
        if ((size_t)at + n >= cap) return false;
        memcpy(dst + at, s, n);
        at += (int)n;
        return true;
    };
    bool ok = true;
    switch (type) {
        case 1:   // ADV_DIRECT_IND: two addresses
            ok = lit("A0:") && (at = put_hex_rev(dst, cap, at, pay, 6)) >= 0 && lit(" A1:") &&
                 (at = put_hex_rev(dst, cap, at, pay + 6, plen >= 12 ? 6 : 0)) >= 0;
            break;
        case 3:   // SCAN_REQ
            ok = lit("ScanA:") && (at = put_hex_rev(dst, cap, at, pay, 6)) >= 0 && lit(" AdvA:") &&
                 (at = put_hex_rev(dst, cap, at, pay + 6, plen >= 12 ? 6 : 0)) >= 0;
            break;
        case 5:   // CONNECT_REQ: many tokens upstream; Snout drops it (token count != 11)
            ok = lit("InitA:") && (at = put_hex_rev(dst, cap, at, pay, 6)) >= 0 && lit(" AdvA:") &&
                 (at = put_hex_rev(dst, cap, at, pay + 6, plen >= 12 ? 6 : 0)) >= 0 &&
                 lit(" LLData:") && (at = put_hex(dst, cap, at, pay + 12, plen > 12 ? plen - 12 : 0)) >= 0 &&
                 lit(" -");
            break;
        default:  // ADV_IND / ADV_NONCONN_IND / ADV_SCAN_IND / SCAN_RSP / reserved
            ok = lit("AdvA:") && (at = put_hex_rev(dst, cap, at, pay, 6)) >= 0 && lit(" Data:") &&
                 (at = put_hex(dst, cap, at, pay + 6, plen - 6)) >= 0;
            break;
    }
    if (!ok || at < 0) return SNOUT_EOVERFLOW;
    int n = snprintf(dst + at, cap - at, " CRC%d\n", p->crc_ok ? 0 : 1);
    if (n < 0 || (size_t)(at + n) >= cap) return SNOUT_EOVERFLOW;
    return at + n;
}

int snout_rftap_encap(const snout_pkt* p, uint8_t* dst, size_t cap)
{
    if (!p || !dst || p->proto != SNOUT_PROTO_ZIGBEE) return SNOUT_EINVAL;
    const size_t total = 16u + p->len;
    if (cap < total) return SNOUT_EOVERFLOW;
    // magic, header length in 32-bit words, flags: bit0 DLT, bit7 QUAL
    static const uint8_t hdr[8] = {'R', 'F', 't', 'a', 4, 0, 0x81, 0x00};
    memcpy(dst, hdr, 8);
    const uint32_t dlt = 195;                       // LINKTYPE_IEEE802_15_4 (with FCS)
    memcpy(dst + 8, &dlt, 4);                       // little-endian host (x86-64)
    const float qual = (float)p->lqi / 255.0f;      // epy_block_0.py:22
    memcpy(dst + 12, &qual, 4);
    memcpy(dst + 16, p->bytes, p->len);
    return (int)total;
}

void snout_zigbee_lane_shape(uint64_t n_channels, uint32_t* core, uint32_t* warmup)
{
    // One shape for every call since ABI 3 (round 5): what a capture decodes to must not depend on how it is cut into
    // submissions.  ABI 4 (round 6): warm-up 3072 instead of 1024.  On 6 998 distinct frames of cfg #4 / #5's dense traffic
    // (profiles/r6_fidelity.md, GPU, against one sequential lane per channel): 6144 / 1024 loses 0.64 % of the sequential
    // loop's frames and reports 1.10 % that it misses, 6144 / 3072 0.30 % + 0.74 % for + 3-4 % of cfg #4's and cfg #5's step time (a fresh
    // loop's phase agrees with the sequential loop's at 60 % of the seams after 1 024 samples, at ~85 % after 3 072);
    // longer cores leave the GPU fewer lanes than it has SIMDs (8192 / 4096: 0.43 % + 0.46 %, + 12 %; 16384 / 8192:
    // 0.19 % + 0.23 %, + 35 %), shorter ones hand over more often.
    // That is the WIDEBAND handles' shape (16 adjacent channels out of one channelizer: a transmitting neighbour 2 MHz
    // either side drags a young loop).  A narrowband handle (n_channels <= 1: one channel as an SDR's own filter delivers
    // it) keeps ABI 3's 6144 / 1024: its traffic has no such neighbours (bench line, 544 distinct frames: 1 lost + 0 extra
    // with 1024, 0 + 0 with 3072), and its lanes -- 163 000 per 1e9 samples -- are throughput-bound, so the longer warm-up
    // would cost it 9 % (4.29 -> 4.67 ms per 1e9 samples).  Either way ONE shape per handle, whatever the size of a call.
    if (core) *core = 6144u;
    if (warmup) *warmup = n_channels <= 1u ? 1024u : 3072u;
}

double snout_zigbee_center_hz(uint32_t channel)
{
    return 1000000.0 * (2400.0 + 5.0 * ((double)channel - 10.0));
}

double snout_btle_center_hz(uint32_t ch)
{
    if (ch == 37) return 2402e6;
    if (ch == 38) return 2426e6;
    if (ch == 39) return 2480e6;
    if (ch <= 10) return (2404.0 + 2.0 * ch) * 1e6;
    if (ch <= 36) return (2428.0 + 2.0 * (ch - 11)) * 1e6;
    return 0.0;
}

int32_t snout_btle_rf_to_channel(uint32_t k)
{   // RF index k: centre 2402 + 2k MHz
    if (k == 0) return 37;
    if (k == 12) return 38;
    if (k == 39) return 39;
    if (k >= 1 && k <= 11) return (int32_t)k - 1;
    if (k >= 13 && k <= 38) return (int32_t)k - 2;
    return -1;
}

}  // extern "C"
