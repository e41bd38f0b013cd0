// snout_rx.cpp — the C ABI of libsnout_rx.so (include/snout_rx.h): handle management, argument
// checking, H->D staging for the host-pointer entry point, profiling read-back.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <new>

namespace snout {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int DevBuf::ensure(size_t bytes)
{
    if (bytes <= cap && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        set_last_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        p = nullptr;
        return SNOUT_ENOMEM;
    }
    cap = want;
    return 0;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

}  // namespace snout

using namespace snout;

struct snout_rx {
    snout_rx_cfg cfg;
    int device = 0;
    BtleCtx btle;
    ZbCtx zb;
    PfbCtx pfb;
    bool wide = false;
    DevBuf d_iq;              // staging for snout_rx_process (host input)
    uint64_t last_n = 0;      // input samples of the last segment
    uint64_t last_nch = 0;    // channel samples per slot of the last segment
    uint64_t last_pkts = 0;
    bool have_prof = false;
};

extern "C" {

uint32_t snout_abi_version(void) { return SNOUT_ABI_VERSION; }

const char* snout_last_error(void) { return g_err; }

const char* snout_strerror(int code)
{
    switch (code) {
        case SNOUT_OK: return "ok";
        case SNOUT_EINVAL: return "invalid argument";
        case SNOUT_ENODEV: return "no usable HIP device (libsnout_rx has no CPU fallback)";
        case SNOUT_ENOMEM: return "out of memory";
        case SNOUT_EHIP: return "HIP runtime error";
        case SNOUT_EOVERFLOW: return "capacity exceeded";
        case SNOUT_ERANGE: return "segment too long";
        default: return "unknown error";
    }
}

int snout_rx_create(const snout_rx_cfg* cfg, snout_rx** out)
{
    if (!cfg || !out) return SNOUT_EINVAL;
    *out = nullptr;
    if (cfg->abi_version != SNOUT_ABI_VERSION) {
        set_last_error("abi_version %u != %u", cfg->abi_version, SNOUT_ABI_VERSION);
        return SNOUT_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_last_error("no HIP device visible");
        return SNOUT_ENODEV;
    }
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) return SNOUT_ENODEV; }
    if (dev >= ndev) { set_last_error("device %d of %d", dev, ndev); return SNOUT_EINVAL; }
    SNOUT_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    SNOUT_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error("device %d is %s; libsnout_rx is built for gfx950 only", dev, prop.gcnArchName);
        return SNOUT_ENODEV;
    }
    snout_rx* h = new (std::nothrow) snout_rx();
    if (!h) return SNOUT_ENOMEM;
    h->cfg = *cfg;
    h->device = dev;
    snout_rx_cfg& c = h->cfg;
    if (c.access_addr == 0) c.access_addr = 0x8E89BED6u;
    if (c.crc_init == 0) c.crc_init = 0x555555u;
    if (c.chip_threshold == 0) c.chip_threshold = 10;
    if (c.taps_per_branch == 0) c.taps_per_branch = 16;
    if (c.zb_core == 0) c.zb_core = 16384;
    if (c.zb_warmup == 0) c.zb_warmup = 2048;
    if (c.n_channels == 0) c.n_channels = 1;
    int rc = SNOUT_EINVAL;
    if (c.proto == SNOUT_PROTO_BTLE && c.n_channels == 1) {
        if (c.channel > 39) { set_last_error("BTLE channel %u", c.channel); goto fail; }
        uint16_t ch = (uint16_t)c.channel;
        rc = h->btle.init(1, &ch, c.access_addr, c.crc_init, c.max_hits);
        if (rc) goto fail;
    } else if (c.proto == SNOUT_PROTO_ZIGBEE && c.n_channels == 1) {
        if (c.channel < 11 || c.channel > 26) { set_last_error("Zigbee channel %u", c.channel); goto fail; }
        if (c.zb_core < 1024 || c.zb_core > (1u << 24) || c.zb_warmup > (1u << 20)) {
            set_last_error("zb_core %u / zb_warmup %u out of range", c.zb_core, c.zb_warmup);
            goto fail;
        }
        uint16_t ch = (uint16_t)c.channel;
        rc = h->zb.init(1, &ch, c.chip_threshold, c.zb_core, c.zb_warmup);
        if (rc) goto fail;
    } else if ((c.proto == SNOUT_PROTO_BTLE && c.n_channels == 40) ||
               (c.proto == SNOUT_PROTO_ZIGBEE && c.n_channels == 16)) {
        // wideband: M-branch channelizer, every bin is a channel slot
        h->wide = true;
        const uint32_t M = c.n_channels;
        uint16_t chs[40];
        for (uint32_t b = 0; b < M; b++)
            chs[b] = c.proto == SNOUT_PROTO_BTLE ? (uint16_t)snout_btle_rf_to_channel((b + 20u) % 40u)
                                                 : (uint16_t)(11u + (b + 8u) % 16u);
        if (c.taps_per_branch != 16) { set_last_error("taps_per_branch must be 16"); goto fail; }
        rc = h->pfb.init(M);
        if (rc) goto fail;
        rc = c.proto == SNOUT_PROTO_BTLE ? h->btle.init(M, chs, c.access_addr, c.crc_init, c.max_hits)
                                         : h->zb.init(M, chs, c.chip_threshold, c.zb_core, c.zb_warmup);
        if (rc) goto fail;
    } else {
        set_last_error("configuration proto=%u n_channels=%u not supported (BTLE: 1 or 40, "
                       "Zigbee: 1 or 16)", c.proto, c.n_channels);
        goto fail;
    }
    *out = h;
    return SNOUT_OK;
fail:
    h->btle.destroy();
    h->zb.destroy();
    h->pfb.destroy();
    delete h;
    return rc;
}

void snout_rx_destroy(snout_rx* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->btle.destroy();
    h->zb.destroy();
    h->pfb.destroy();
    h->d_iq.release();
    delete h;
}

int snout_rx_process_dev(snout_rx* h, const float* iq_dev, uint64_t n_samples,
                         uint64_t first_sample_index, void* hip_stream, snout_pkt* out, uint64_t cap,
                         uint64_t* n_out)
{
    if (!h || !n_out || (!out && cap) || (!iq_dev && n_samples)) return SNOUT_EINVAL;
    *n_out = 0;
    h->have_prof = false;
    if (n_samples >= 0xFFFF0000ull) { set_last_error("segment of %llu samples", (unsigned long long)n_samples); return SNOUT_ERANGE; }
    SNOUT_HIP(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (n_samples < 5) {
        if (h->wide) { h->pfb.n_out = 0; h->last_nch = 0; }
        return SNOUT_OK;
    }
    // wideband input: channelize into [M][n_ch] channel IQ, then run the per-channel path on it
    const float* ch_iq = iq_dev;
    uint64_t n_ch = n_samples, ch_stride = n_samples;
    if (h->wide) {
        n_ch = h->pfb.n_out_for(n_samples);
        h->last_n = n_samples;
        h->last_nch = n_ch;
        if (n_ch < 5) {     // too short for any demodulator; still channelize (soft tap)
            h->have_prof = false;
            if (int rc = h->pfb.run(iq_dev, n_samples, st)) return rc;
            SNOUT_HIP(hipStreamSynchronize(st));
            return SNOUT_OK;
        }
    }
    if (h->cfg.proto == SNOUT_PROTO_BTLE) {
        BtleCtx& b = h->btle;
        int rc = 0;
        for (int attempt = 0; attempt < 12; attempt++) {
            if ((rc = b.reserve(n_ch))) return rc;
            if ((rc = b.begin(st))) return rc;
            if (h->wide) {
                if ((rc = h->pfb.run(iq_dev, n_samples, st))) return rc;
                ch_iq = h->pfb.d_y.as<float>();
                ch_stride = h->pfb.y_stride;
            }
            if ((rc = b.launch_demod_corr(ch_iq, n_ch, ch_stride, st))) return rc;
            rc = b.finish(n_ch, first_sample_index, st, out, cap, n_out);
            if (rc != SNOUT_EOVERFLOW || !(b.overflow_chunk || b.overflow_cand)) break;
            // more hits than provisioned: grow and run the segment again (results never truncated)
            if (b.overflow_chunk) b.hit_cap = std::min<uint32_t>(b.hit_cap * 4u, kChunkSamples);
            if (b.overflow_cand) b.max_cand_grown = b.max_cand * 4u;
        }
        h->last_n = n_samples;
        h->last_nch = n_ch;
        h->last_pkts = *n_out;
        h->have_prof = true;
        return rc;
    }
    if (h->cfg.proto == SNOUT_PROTO_ZIGBEE) {
        ZbCtx& z = h->zb;
        int rc = 0;
        for (int attempt = 0; attempt < 8; attempt++) {
            if ((rc = z.reserve(n_ch))) return rc;
            if (h->wide) {
                if ((rc = h->pfb.run(iq_dev, n_samples, st))) return rc;
                ch_iq = h->pfb.d_y.as<float>();
                ch_stride = h->pfb.y_stride;
            }
            rc = z.run(ch_iq, n_ch, ch_stride, first_sample_index, st, out, cap, n_out);
            if (rc != SNOUT_EOVERFLOW || !z.overflow) break;
            z.pkts_per_lane *= 4;       // a lane held more frames than provisioned: run again
        }
        h->last_n = n_samples;
        h->last_nch = n_ch;
        h->last_pkts = *n_out;
        h->have_prof = true;
        return rc;
    }
    return SNOUT_EINVAL;
}

int snout_rx_process(snout_rx* h, const float* iq_host, uint64_t n_samples,
                     uint64_t first_sample_index, snout_pkt* out, uint64_t cap, uint64_t* n_out)
{
    if (!h || !n_out || (!iq_host && n_samples)) return SNOUT_EINVAL;
    *n_out = 0;
    if (n_samples < 5) return SNOUT_OK;
    SNOUT_HIP(hipSetDevice(h->device));
    if (int rc = h->d_iq.ensure(n_samples * 8u)) return rc;
    SNOUT_HIP(hipMemcpy(h->d_iq.p, iq_host, n_samples * 8u, hipMemcpyHostToDevice));
    return snout_rx_process_dev(h, h->d_iq.as<float>(), n_samples, first_sample_index, nullptr, out,
                                cap, n_out);
}

void* snout_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        set_last_error("hipHostMalloc(%zu) failed", bytes);
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void snout_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int snout_rx_profile(snout_rx* h, snout_rx_prof* out)
{
    if (!h || !out) return SNOUT_EINVAL;
    memset(out, 0, sizeof(*out));
    if (!h->have_prof) { set_last_error("no processed segment to profile"); return SNOUT_EINVAL; }
    out->bytes_algorithmic = 8ull * h->last_n + 160ull * h->last_pkts;
    if (h->wide) {
        // the channelizer is the dominant kernel of the wideband paths
        BtleCtx& b = h->btle;
        ZbCtx& z = h->zb;
        const bool bt = h->cfg.proto == SNOUT_PROTO_BTLE;
        SNOUT_HIP(hipEventElapsedTime(&out->ms_total, h->pfb.ev_k0, bt ? b.ev_t1 : z.ev_t1));
        SNOUT_HIP(hipEventElapsedTime(&out->ms_dominant, h->pfb.ev_k0, h->pfb.ev_k1));
        out->dominant_launches = 1;
        out->n_hits = bt ? b.last_n_cand : z.total_lanes;
        snprintf(out->dominant_name, sizeof(out->dominant_name), "pfb_channelize<%u>", h->pfb.M);
        return SNOUT_OK;
    }
    if (h->cfg.proto == SNOUT_PROTO_ZIGBEE) {
        ZbCtx& z = h->zb;
        SNOUT_HIP(hipEventElapsedTime(&out->ms_total, z.ev_t0, z.ev_t1));
        SNOUT_HIP(hipEventElapsedTime(&out->ms_dominant, z.ev_k0, z.ev_k1));
        out->dominant_launches = 2;
        out->n_hits = z.total_lanes;
        snprintf(out->dominant_name, sizeof(out->dominant_name), "zb_discrim+zb_lanes");
        return SNOUT_OK;
    }
    BtleCtx& b = h->btle;
    SNOUT_HIP(hipEventElapsedTime(&out->ms_total, b.ev_t0, b.ev_t1));
    SNOUT_HIP(hipEventElapsedTime(&out->ms_dominant, b.ev_k0, b.ev_k1));
    out->dominant_launches = 1;
    out->n_hits = b.last_n_cand;
    snprintf(out->dominant_name, sizeof(out->dominant_name), "btle_demod_corr");
    return SNOUT_OK;
}

int snout_rx_soft(snout_rx* h, uint32_t stage, uint32_t channel_slot, float* out, uint64_t cap,
                  uint64_t* n_out)
{
    if (!h || !n_out) return SNOUT_EINVAL;
    *n_out = 0;
    SNOUT_HIP(hipSetDevice(h->device));
    if (stage == SNOUT_STAGE_CHAN_IQ && h->wide) {
        if (channel_slot >= h->pfb.M) return SNOUT_EINVAL;
        const uint64_t nf = 2ull * h->pfb.n_out;
        const uint64_t m = nf < cap ? nf : cap;
        SNOUT_HIP(hipMemcpy(out, h->pfb.d_y.as<float>() + 2ull * channel_slot * h->pfb.y_stride, m * 4u,
                            hipMemcpyDeviceToHost));
        *n_out = nf;
        return nf > cap ? SNOUT_EOVERFLOW : SNOUT_OK;
    }
    if (stage == SNOUT_STAGE_BTLE_BITS && h->cfg.proto == SNOUT_PROTO_BTLE) {
        BtleCtx& b = h->btle;
        if (channel_slot >= b.n_slots || h->last_nch < 5) return SNOUT_EINVAL;
        const uint64_t nb = h->last_nch - 4;
        const uint64_t words = (uint64_t)b.n_chunks * kChunkIters * 4u;
        std::vector<uint64_t> pl(words);
        SNOUT_HIP(hipMemcpy(pl.data(), b.d_planes.as<uint64_t>() + channel_slot * b.plane_stride,
                            words * 8u, hipMemcpyDeviceToHost));
        *n_out = nb;
        const uint64_t m = nb < cap ? nb : cap;
        for (uint64_t n = 0; n < m; n++) {
            const uint64_t g = n >> 8, l = (n & 255u) >> 2, j = n & 3u;
            out[n] = (float)((pl[g * 4u + j] >> l) & 1ull);
        }
        return nb > cap ? SNOUT_EOVERFLOW : SNOUT_OK;
    }
    if (h->cfg.proto == SNOUT_PROTO_ZIGBEE && stage >= SNOUT_STAGE_ZB_DISCRIM &&
        stage <= SNOUT_STAGE_ZB_CHIPS) {
        if (h->last_nch < 9) return SNOUT_EINVAL;
        return h->zb.soft(stage, channel_slot, h->last_nch, out, cap, n_out);
    }
    set_last_error("stage %u not available for this configuration", stage);
    return SNOUT_EINVAL;
}

}  // extern "C"
