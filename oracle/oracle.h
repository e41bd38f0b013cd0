/*
 * oracle.h — declarations of the CPU oracle (TEST INFRASTRUCTURE ONLY; see oracle_btle.c).
 * The record layout is the ABI one from include/snout_rx.h.
 */
#ifndef SNOUT_ORACLE_H
#define SNOUT_ORACLE_H
#include <stdint.h>
#include "../include/snout_rx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BTLE (oracle_btle.c) ---- */
void     oracle_btle_whiten_seq(uint32_t channel, uint8_t* out, int nbytes);
uint32_t oracle_btle_crc24(const uint8_t* data, int n, uint32_t init);
uint32_t oracle_btle_crc24_table(const uint8_t* data, int n, uint32_t init);
void     oracle_btle_crc_bytes(uint32_t r, uint8_t out[3]);
uint64_t oracle_btle_bits(const float* iq, uint64_t n_samples, uint8_t* bits);
uint64_t oracle_btle_all_hits(const uint8_t* bits, uint64_t nb, uint32_t access_addr,
                              uint64_t* hits, uint64_t cap);
int      oracle_btle_segment(const float* iq, uint64_t n_samples, uint64_t first_sample_index,
                             uint32_t channel, uint32_t access_addr, uint32_t crc_init,
                             snout_pkt* out, uint64_t cap, uint64_t* n_out,
                             uint64_t* hits_out, uint64_t hits_cap, uint64_t* n_hits_out);

/* ---- Zigbee / IEEE 802.15.4 (oracle_zigbee.c) ---- */
const uint32_t* oracle_zb_chip_map(void);
const float*    oracle_zb_mmse_taps(void);
float    oracle_fast_atan2f(float y, float x);
void     oracle_zb_discrim(const float* iq, uint64_t n, float* d);
uint16_t oracle_crc16_154(const uint8_t* d, int n);
void     oracle_zb_iir_tables(double w[64], double* d64);
double*  oracle_zb_iir_carry(const float* d, uint64_t n, uint32_t core, uint32_t warmup, uint64_t n_lanes);
int      oracle_zigbee_segment(const float* iq, uint64_t n, uint64_t first_index, uint32_t channel,
                               uint32_t threshold, uint32_t core, uint32_t warmup,
                               snout_pkt* out, uint64_t cap, uint64_t* n_out);
int      oracle_zigbee_lane_soft(const float* iq, uint64_t n, uint32_t core, uint32_t warmup,
                                 uint32_t lane, uint32_t threshold, float* z, float* chips,
                                 uint64_t cap, uint64_t* n_chips);

/* ---- polyphase channelizer + wideband receivers (oracle_pfb.c) ---- */
uint64_t oracle_pfb_nout(uint64_t n, uint32_t M);
const float* oracle_pfb_proto(uint32_t M);
int      oracle_pfb(const float* iq, uint64_t n, uint32_t M, float* y, uint64_t y_stride);
/* the same sums in the term order of the experimental matrix-pipe FIR (snout_amd/csrc/pfb_mfma.hip); differs from
 * oracle_pfb only for non-finite samples and results that are zero */
int oracle_pfb_block_order(const float* iq, uint64_t n, uint32_t M, float* y, uint64_t y_stride);
/* M = 40: 1 = the Cooley-Tukey FFT with twiddles that rounds 1-3 specified (what the A/B partners pfb.hip / pfb_mfma.hip compute),
 * 0 = the shipped prime-factor form; process-wide, set around a call by the tests of those two kernels only */
void oracle_pfb_legacy_fft(int on);
uint32_t oracle_btle_bin_channel(uint32_t bin);
uint32_t oracle_zigbee_bin_channel(uint32_t bin);
int      oracle_wideband_segment(const float* iq, uint64_t n, uint64_t first_index, uint32_t proto,
                                 uint32_t access_addr, uint32_t crc_init, uint32_t threshold,
                                 uint32_t core, uint32_t warmup, snout_pkt* out, uint64_t cap,
                                 uint64_t* n_out);

/* ---- threads (bench.py's cpu_baseline legs) ---- */
int      oracle_set_threads(int n);      /* OpenMP threads for oracle_pfb / the parallel receivers; returns the setting */
int      oracle_hw_threads(void);
int      oracle_narrowband_parallel(const float* iq, uint64_t n, uint32_t proto, uint32_t channel,
                                    uint32_t access_addr, uint32_t crc_init, uint32_t threshold, uint32_t core,
                                    uint32_t warmup, uint64_t seg, uint64_t overlap, snout_pkt* out, uint64_t cap,
                                    uint64_t* n_out);
int      oracle_wideband_parallel(const float* iq, uint64_t n, uint32_t proto, uint32_t access_addr,
                                  uint32_t crc_init, uint32_t threshold, uint32_t core, uint32_t warmup,
                                  uint64_t seg, snout_pkt* out, uint64_t cap, uint64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif
