/*
 * snout_rx.h — C ABI of libsnout_rx.so, the MI355X (gfx950) IQ -> packets receive path.
 *
 * This is the drop-in boundary for the two process boundaries the reference crosses on its
 * receive hot path:
 *
 *   BTLE   : the `btle_rx` child process started at  snout/util/btle.py:53,63-69
 *            (argv `-c CH -g 6 -a 8e89bed6 -k 555555`), whose stdout lines are parsed by
 *            snout/core/message.py:205-237.  -> snout_rx_create(proto=SNOUT_PROTO_BTLE) +
 *            snout_rx_process*() + snout_btle_format_line().
 *   Zigbee : the GNU Radio flowgraph snout/modulations/Zigbee/hackrf/Zigbee_rx/top_block.py:52-89
 *            (quadrature_demod_cf -> x - single_pole_iir(x) -> clock_recovery_mm_ff ->
 *            ieee802_15_4.packet_sink(10) -> epy_block_0 -> rftap_encap(2,195,'') -> UDP 52002).
 *            -> snout_rx_create(proto=SNOUT_PROTO_ZIGBEE) + snout_rx_process*() +
 *            snout_rftap_encap().
 *
 * Conventions: plain pointers and sizes only (no torch / HIP types in signatures; a HIP stream is
 * passed as void*).  Every function returns 0 on success or a negative SNOUT_E* code.  The caller
 * owns every buffer it passes.  A handle is not thread-safe; different handles are independent.
 * No callbacks.  The library never falls back to a CPU path: without a usable gfx950 device
 * snout_rx_create() fails with SNOUT_ENODEV.
 */
#ifndef SNOUT_RX_H
#define SNOUT_RX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: the default 802.15.4 lane shape of a WIDEBAND handle is 6144 / 3072 (warm-up 1024 before, and still for a narrowband
 *    handle; snout_zigbee_lane_shape takes the handle's n_channels): half the frames lost against the one sequential
 *    receiver for + 3-4 % step time on the wideband 802.15.4 workloads (profiles/r6_fidelity.md); a client built against 3 fails the
 *    handshake instead of decoding another frame set
 * 3: the 802.15.4 frame repair (snout_pkt.flags SNOUT_PKT_ZB_REPAIRED) and ONE default lane shape (then 6144 / 1024), whatever the size
 *    of a call (snout_zigbee_lane_shape ignores its argument): the records of a capture no longer depend on how it is cut
 *    into submissions, and the default decode loses <= 1 % of the one sequential receiver's frames on dense traffic and
 *    finds 1-2 % that it misses (measured per run: bench.py frames_lost_vs_sequential)
 * 2: snout_rx_pack_last_records(skip, longest_dev), SNOUT_CFG_RECORDS_ON_DEVICE, batches of up to 64 segments,
 *    snout_pkt.flags SNOUT_PKT_ZB_SEAM_DISAGREED, snout_zigbee_lane_shape (the default 802.15.4 lane shape depends on the
 *    size of the call), the bench aid moved to snout_bench.h -- a client built against version 1 fails the handshake in
 *    snout_rx_create instead of decoding a different frame set or missing a symbol */
#define SNOUT_ABI_VERSION 4u

/* protocols (snout/core/protocols/__init__.py:2-27 names them BTLE / ZIGBEE) */
#define SNOUT_PROTO_BTLE   0u
#define SNOUT_PROTO_ZIGBEE 1u

/* input sample formats (snout_rx_cfg.sample_format).  Integer samples stand for the cf32 samples
 * v * 2^-7 (sc8) / v * 2^-15 (sc16): conversion and scale are exact in fp32, so results are bit-
 * identical to processing the converted capture as cf32. */
#define SNOUT_FMT_CF32 0u   /* interleaved float32 I,Q (numpy.complex64, gr_complex), 8 B/sample       */
#define SNOUT_FMT_SC8  1u   /* interleaved int8 I,Q: HackRF transfers, the input of upstream btle_rx
                               (SURVEY Appendix A.1), 2 B/sample                                     */
#define SNOUT_FMT_SC16 2u   /* interleaved int16 I,Q (USRP sc16 captures), 4 B/sample                */

/* snout_rx_cfg.flags */
#define SNOUT_CFG_KEEP_CHANNEL_IQ 1u  /* a wideband handle keeps channel IQ in HBM (unfused kernels;
                                         needed for the SNOUT_STAGE_CHAN_IQ tap).  Default: the
                                         channelizer feeds the BTLE bit planes / the 802.15.4
                                         discriminator rows directly                              */
#define SNOUT_CFG_RECORDS_ON_DEVICE 2u /* pipelined entry points: the records of a collected segment stay in
                                         device memory; snout_rx_collect_view hands out their COUNT and a NULL
                                         pointer, snout_rx_last_records_dev / snout_rx_pack_last_records the
                                         records.  For a consumer on the GPU (the multi-GPU gather): without it
                                         every submission's records are also downloaded to pinned host memory */

/* error codes */
#define SNOUT_OK          0
#define SNOUT_EINVAL     -1   /* bad argument / configuration                      */
#define SNOUT_ENODEV     -2   /* no usable HIP device (no CPU fallback exists)     */
#define SNOUT_ENOMEM     -3   /* device or host allocation failed                  */
#define SNOUT_EHIP       -4   /* a HIP runtime call failed (see snout_last_error)  */
#define SNOUT_EOVERFLOW  -5   /* more hits / packets than the configured capacity  */
#define SNOUT_ERANGE     -6   /* segment too long for 32-bit in-segment indices    */

/* snout_rx_soft() stage selectors (test-only taps of soft intermediates)   */
#define SNOUT_STAGE_BTLE_BITS     0u  /* a1: hard bits, one uint8 (0/1) per sample, as floats  */
#define SNOUT_STAGE_CHAN_IQ       1u  /* PFB output, channel-major interleaved cf32            */
#define SNOUT_STAGE_ZB_DISCRIM    2u  /* a4: quadrature_demod_cf output                        */
#define SNOUT_STAGE_ZB_DCREMOVED  3u  /* a5: x - single_pole_iir(x)                            */
#define SNOUT_STAGE_ZB_CHIPS      4u  /* a6: clock_recovery_mm_ff output (lane 0 of channel 0) */

typedef struct snout_rx_cfg {
    uint32_t abi_version;     /* SNOUT_ABI_VERSION                                              */
    uint32_t proto;           /* SNOUT_PROTO_*                                                  */
    uint32_t n_channels;      /* 1: input is one narrowband channel at 4 Msps (no channelizer).
                                 M>1: wideband input, M-branch polyphase channelizer, 2x oversampled
                                 (decimation M/2): BTLE M=40 (80 Msps), Zigbee M=16 (32 Msps)    */
    uint32_t taps_per_branch; /* P of the PFB prototype (M*P taps); 0 -> 16                     */
    uint32_t channel;         /* n_channels==1: protocol channel number of the input
                                 (BTLE 0..39 -> whitening seed, `-c` of btle.py:64;
                                  Zigbee 11..26, top_block.py:94-96). Ignored for wideband.     */
    uint32_t access_addr;     /* BTLE `-a` (btle.py:66); 0 -> 0x8E89BED6                         */
    uint32_t crc_init;        /* BTLE `-k` (btle.py:67); 0 -> 0x555555                           */
    uint32_t chip_threshold;  /* packet_sink(threshold) (top_block.py:67); 0 -> 10               */
    uint32_t zb_core;         /* Zigbee lane core length in channel samples; 0 (with zb_warmup != 0) -> 2048 */
    uint32_t zb_warmup;       /* Zigbee lane warm-up before its core, channel samples; 0 (with zb_core != 0) -> 512;
                                 multiples of 64, warm-up < core; the timing loop needs >= 256.
                                 BOTH 0 (the default): core 6144 / warm-up 3072 for a wideband handle, 6144 / 1024 for a
                                 narrowband one (snout_zigbee_lane_shape(n_channels)),
                                 the same for every call of every handle.  The decoded frame set is a function
                                 of the shape and of where the calls cut the capture;
                                 zb_core >= the call's channel samples is the reference's one sequential loop  */
    uint32_t max_hits;        /* capacity for candidate hits per call; 0 -> auto                 */
    int32_t  device;          /* HIP device ordinal; <0 -> current device                        */
    uint32_t flags;           /* SNOUT_CFG_* bits                                                */
    uint32_t sample_format;   /* SNOUT_FMT_* of every iq pointer handed to this handle (0 = cf32)  */
    uint32_t batch_segments;  /* segments one snout_rx_submit_batch_dev call may carry; 0 -> 1.
                                 > 1: wideband handles (n_channels > 1) only, at most 64          */
    uint32_t reserved_cus;    /* wideband handles: compute units the channelizer's persistent grid leaves free
                                 (0: none).  The channelizer holds one 16-wave workgroup with all of a CU's LDS
                                 on every CU it runs on, so nothing else runs beside it there: a caller with
                                 concurrent work on other streams that must not wait for a launch to end --
                                 the RCCL kernels and the record download of a multi-GPU gather -- reserves
                                 a few CUs (8 = one per XCD costs the channelizer 3 %)                */
} snout_rx_cfg;

/* snout_pkt.flags of an 802.15.4 record: the clock recovery runs in lanes (cfg.zb_core); where one lane's timing
 * loop hands over to the next INSIDE this frame, the two loops had decided some of the frame's last 48 chips before
 * the hand-over differently -- the frame's chips depend on which loop is asked, so the reference's one sequential
 * loop may have decided this frame differently too.  Never set with one lane per channel. */
#define SNOUT_PKT_ZB_SEAM_DISAGREED 0x04u
/* An 802.15.4 frame that the lanes' sink gave up (or finished with a bad FCS) behind a hand-over and that was received
 * again by ONE timing loop from the lane that found its SFD to its last chip, as the reference's sequential loop
 * receives every frame (frame repair).  Never set with one lane per channel. */
#define SNOUT_PKT_ZB_REPAIRED 0x08u

/* One decoded packet. Fixed 160 bytes so records can be gathered across ranks as flat bytes. */
typedef struct snout_pkt {
    uint64_t sample_index;    /* BTLE: first sample of the access address; Zigbee: window start of the
                                 chip 319 chips before the one that completes the SFD, i.e. the first
                                 chip of a regular preamble + SFD (the same for every sink that finds the
                                 frame; the packet sink itself reports no position). In channel samples,
                                 plus first_sample_index of the call.                            */
    uint32_t proto;           /* SNOUT_PROTO_*                                                   */
    uint16_t channel;         /* protocol channel number (BTLE 0..39, Zigbee 11..26)             */
    uint16_t len;             /* valid bytes in bytes[]: BTLE 2+payload+3 (header, payload, CRC),
                                 Zigbee PSDU length (FCS included, not checked — as packet_sink)  */
    uint8_t  crc_ok;          /* BTLE: 1 if CRC24 matches (btle_rx prints CRC0). Zigbee: FCS-16 ok */
    uint8_t  lqi;             /* Zigbee LQI as packet_sink computes it; BTLE 0                   */
    uint8_t  pdu_type;        /* BTLE header & 0x0F                                              */
    uint8_t  flags;           /* BTLE: TxAdd | RxAdd<<1.  Zigbee: SNOUT_PKT_ZB_SEAM_DISAGREED | SNOUT_PKT_ZB_REPAIRED */
    uint32_t aux;             /* BTLE: sample phase 0..3 of the hit; Zigbee: lane id             */
    uint8_t  bytes[136];
} snout_pkt;

/* Device-time breakdown of the last snout_rx_process*() call, measured with HIP events on the
 * stream the kernels were launched on. */
typedef struct snout_rx_prof {
    float    ms_total;        /* first kernel start -> records landed in host memory             */
    float    ms_dominant;     /* the dominant streaming kernel alone                             */
    uint32_t dominant_launches;
    uint32_t n_hits;          /* candidate hits before resolution (BTLE) / lanes (Zigbee)        */
    uint64_t bytes_algorithmic; /* 8 B (cf32; sc8 2, sc16 4) x input samples + 160 B x packets */
    char     dominant_name[48];
} snout_rx_prof;

typedef struct snout_rx snout_rx;

int  snout_rx_create (const snout_rx_cfg* cfg, snout_rx** out);
void snout_rx_destroy(snout_rx* h);

/* One capture segment, synchronous. iq = interleaved (re,im) in the handle's sample format (cf32
 * unless cfg.sample_format says otherwise), 16-byte aligned, n_samples complex samples.
 * Packets are written in ascending (channel, sample_index) order. *n_out receives the number of
 * packets found (may exceed cap -> SNOUT_EOVERFLOW, first cap records valid).
 *
 * Wideband handles (n_channels > 1): iq is the wideband capture; records carry the protocol
 * channel of their bin and sample_index counts CHANNEL samples (4 Msps), first_sample_index is
 * added as given (pass it in channel samples).
 *
 * snout_rx_process      : iq in HOST memory; copied H->D (PCIe-inclusive path).
 * snout_rx_process_dev  : iq already resident in DEVICE memory (HBM); hip_stream is a hipStream_t
 *                         (NULL = the legacy default stream). out is HOST memory. */
int  snout_rx_process    (snout_rx* h, const void* iq_host, uint64_t n_samples,
                          uint64_t first_sample_index, snout_pkt* out, uint64_t cap, uint64_t* n_out);
int  snout_rx_process_dev(snout_rx* h, const void* iq_dev, uint64_t n_samples,
                          uint64_t first_sample_index, void* hip_stream,
                          snout_pkt* out, uint64_t cap, uint64_t* n_out);

/* Pipelined form: up to three segments may be in flight.  submit enqueues every kernel of a segment
 * on hip_stream and returns without waiting; collect waits for the oldest submitted segment and
 * hands out its records (collect_view: a pointer into the handle's pinned buffer, valid until three
 * more submits; NULL and the count alone with SNOUT_CFG_RECORDS_ON_DEVICE).  The record D2H of segment i runs on an internal copy stream and overlaps the
 * kernels of segment i+1.  iq_dev must stay valid and unchanged until its collect returns. */
int  snout_rx_submit_dev  (snout_rx* h, const void* iq_dev, uint64_t n_samples,
                           uint64_t first_sample_index, void* hip_stream);
/* Several capture segments of EQUAL length as one submission (handle created with cfg.batch_segments
 * >= count): short segments leave most of the GPU idle inside the lane-serial 802.15.4 kernels (a 2^24-
 * sample wideband segment is 256 waves of clock recovery on 256 CUs) and pay the channelizer's prologue per
 * launch; a batch runs them side by side in the same launches (48 segments of 2^24 samples at the rate of one
 * 8e8-sample segment).
 * The records of the batch come out of ONE collect, ordered by (segment, channel, sample_index), each
 * with its own segment's first_sample_index[k] added; records whose sample_index is below
 * min_sample_index[k] are dropped (NULL: none) -- a sharded scan leaves what a segment finds in its
 * pre-roll to the segment before it (replaces `snout/core/radio.py:415`'s one-channel-at-a-time loop
 * together with snout_rx_submit_dev).  Same results as `count` single submissions. */
int  snout_rx_submit_batch_dev(snout_rx* h, const void* const* iq_devs, uint32_t count, uint64_t n_samples,
                               const uint64_t* first_sample_index, const uint64_t* min_sample_index,
                               void* hip_stream);
/* 1 if the oldest submitted segment has finished, 0 if not yet (or nothing is pending), negative on error:
 * lets one host thread drive several handles without blocking on one.  After a 1 the collect does not wait for
 * the device -- unless the segment exceeded a capacity (hits per chunk, frames per lane, candidates): collect
 * then grows it and runs the segment again before it returns, as the synchronous entry points do. */
int  snout_rx_poll        (snout_rx* h);
int  snout_rx_collect     (snout_rx* h, snout_pkt* out, uint64_t cap, uint64_t* n_out);
int  snout_rx_collect_view(snout_rx* h, const snout_pkt** recs, uint64_t* n_out);
/* Device copy of the records of the segment collected last (same lifetime as the collect_view
 * pointer): lets a multi-GPU gather take them GPU -> GPU (RCCL) without a trip through host memory. */
int  snout_rx_last_records_dev(snout_rx* h, const snout_pkt** recs_dev, uint64_t* n_out);

/* Page-locked host memory for `out`: records are then DMA'd straight into it (no staging copy).
 * Any other host pointer works too, through an internal pinned staging buffer. */
void* snout_host_alloc(size_t bytes);
void  snout_host_free(void* p);

/* Copy a soft intermediate of the LAST processed segment to host floats (tests only). */
int  snout_rx_soft   (snout_rx* h, uint32_t stage, uint32_t channel_slot,
                      float* out, uint64_t cap, uint64_t* n_out);
int  snout_rx_profile(snout_rx* h, snout_rx_prof* out);
/* Durations (ms, HIP events on the kernels' stream) of the dominant kernel of the last <= 64
 * segments, oldest first.  Read after the fact so a pipelined run is not perturbed. */
int  snout_rx_profile_history(snout_rx* h, float* ms, uint32_t cap, uint32_t* n_out);

/* Multi-GPU gather (no reference counterpart: the reference has one radio): copy the
 * device copy of the records of the segment collected last, from record `skip` on, into an exchange buffer
 * on the device, in the wire format = the first `width` bytes (a multiple of 16, 32..160) of every 160-byte
 * record, on `hip_stream`.  Records with sample_index < own_from (found in a segment's pre-roll: another
 * segment reports them) get sample_index = 2^62 and are dropped by the gather's sort.  longest_dev (may be
 * NULL): a device uint64 raised (atomic max) to the largest snout_pkt.len packed, so that a consumer that
 * never sees the records on the host can still refuse a record its wire format would truncate.
 * *n_packed = records written (<= dst_cap; SNOUT_EOVERFLOW if the segment had more behind `skip`). */
int  snout_rx_pack_last_records(snout_rx* h, void* dst_dev, uint64_t dst_cap, uint32_t width, uint64_t own_from,
                                uint64_t skip, void* longest_dev, void* hip_stream, uint64_t* n_packed);

/* Rank 0 of the gather: sort the gathered wire records (`blocks` rank blocks of `cap` record slots of `width` bytes,
 * block b valid up to min(counts_dev[b * counts_stride], cap)) by (proto, channel, sample_index) and drop what the overlaps
 * of neighbouring capture segments found twice -- same (proto, channel), sample_index at most `tol` apart and, for
 * tol > 0 (802.15.4: every receiver locks at its own phase), equal length and bytes; records disowned by
 * snout_rx_pack_last_records (sample_index = 2^62) are dropped.  The kept records go to out_dev (room for blocks * cap),
 * their number to *n_keep_dev; everything is enqueued on hip_stream, nothing waits.  work_dev: scratch of
 * snout_records_dedup_workspace(blocks, cap) bytes.  (The host statement of the same rule: snout_amd/dist.py::dedup_records.) */
size_t snout_records_dedup_workspace(uint32_t blocks, uint64_t cap);
int    snout_records_dedup(const void* rows_dev, uint32_t width, uint32_t blocks, uint64_t cap, const int64_t* counts_dev,
                           uint32_t counts_stride, uint32_t tol, void* out_dev, uint64_t* n_keep_dev, void* work_dev,
                           size_t work_bytes, void* hip_stream);

/* Host-side formatters for the two consumer contracts. */
/* btle_rx stdout grammar (snout/core/message.py:214-215,226-236). Returns bytes written
 * (excluding NUL) or negative error. */
int  snout_btle_format_line(const snout_pkt* p, double fs_hz, double t0_epoch, uint32_t pkt_number,
                            uint32_t access_addr, char* dst, size_t cap);
/* RFtap header + MPDU, as rftap_encap(2,195,'') emits with meta{qual=lqi/255}
 * (top_block.py:53, epy_block_0.py:20-24). Returns datagram length or negative error. */
int  snout_rftap_encap(const snout_pkt* p, uint8_t* dst, size_t cap);

/* Channel plans (a10). */
double   snout_zigbee_center_hz(uint32_t channel);   /* 1e6*(2400+5*(ch-10)), top_block.py:56,94-96 */
/* The 802.15.4 lane shape a handle with cfg.zb_core = cfg.zb_warmup = 0 uses: the clock recovery (clock_recovery_mm_ff,
 * top_block.py:69) runs in lanes of `core` samples that start `warmup` samples early.  Since ABI 3 one shape for every call
 * (`channel_samples` is ignored); results are a function of the shape, so a checker has to run the same one. */
void     snout_zigbee_lane_shape(uint64_t n_channels, uint32_t* core, uint32_t* warmup);
double   snout_btle_center_hz(uint32_t channel);     /* 37->2402, 38->2426, 39->2480, data channels */
int32_t  snout_btle_rf_to_channel(uint32_t rf_index);/* RF k (2402+2k MHz) -> BLE channel index    */

const char* snout_strerror(int code);
const char* snout_last_error(void);  /* thread-local detail string of the last failure */
uint32_t    snout_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SNOUT_RX_H */
