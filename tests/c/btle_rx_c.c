/*
 * A plain-C consumer of include/snout_rx.h: what a maintainer's cgo / JNI / N-API / C glue would do.
 * Reads a capture (cf32, or the int8 IQ a HackRF writes and upstream btle_rx reads, or int16),
 * runs the BTLE receive path through the C ABI only (no Python, no torch) and prints the
 * btle_rx-format lines, like the child process the reference spawns (snout/util/btle.py:53,63-69).
 * Built by tests/test_c_abi_gpu.py with gcc + -lsnout_rx.
 *
 *   btle_rx_c <capture> <channel> <t0_epoch> [cf32|sc8|sc16]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "snout_rx.h"

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s capture channel t0 [cf32|sc8|sc16]\n", argv[0]); return 2; }
    uint32_t fmt = SNOUT_FMT_CF32, sample_bytes = 8;
    if (argc > 4 && !strcmp(argv[4], "sc8")) { fmt = SNOUT_FMT_SC8; sample_bytes = 2; }
    if (argc > 4 && !strcmp(argv[4], "sc16")) { fmt = SNOUT_FMT_SC16; sample_bytes = 4; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    const uint64_t n = (uint64_t)bytes / sample_bytes;
    void* iq = malloc((size_t)bytes);
    if (fread(iq, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);

    snout_rx_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = snout_abi_version();
    cfg.proto = SNOUT_PROTO_BTLE;
    cfg.n_channels = 1;
    cfg.channel = (uint32_t)atoi(argv[2]);
    cfg.device = -1;
    cfg.sample_format = fmt;
    snout_rx* h = NULL;
    int rc = snout_rx_create(&cfg, &h);
    if (rc) { fprintf(stderr, "create: %s: %s\n", snout_strerror(rc), snout_last_error()); return 1; }

    uint64_t cap = 4096, n_out = 0;
    snout_pkt* out = (snout_pkt*)snout_host_alloc(cap * sizeof(snout_pkt));   /* pinned: records are DMA'd into it */
    if (!out) { fprintf(stderr, "alloc: %s\n", snout_last_error()); return 1; }
    rc = snout_rx_process(h, iq, n, 0, out, cap, &n_out);
    if (rc) { fprintf(stderr, "process: %s: %s\n", snout_strerror(rc), snout_last_error()); return 1; }

    char line[512];
    for (uint64_t i = 0; i < n_out; i++) {
        const int k = snout_btle_format_line(&out[i], 4e6, atof(argv[3]), (uint32_t)i, 0x8E89BED6u, line, sizeof(line));
        if (k < 0) { fprintf(stderr, "format: %s\n", snout_strerror(k)); return 1; }
        fwrite(line, 1, (size_t)k, stdout);
    }
    snout_rx_prof prof;
    if (snout_rx_profile(h, &prof) == 0)
        fprintf(stderr, "%llu packets, %s %.3f ms\n", (unsigned long long)n_out, prof.dominant_name, prof.ms_dominant);
    snout_host_free(out);
    snout_rx_destroy(h);
    free(iq);
    return 0;
}
