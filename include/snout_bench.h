/* snout_bench.h — measurement aids exported by libsnout_rx.so beside the receive-path ABI.
 *
 * NOT part of the drop-in boundary (include/snout_rx.h): nothing the reference's call sites would bind.
 * bench.py uses it to state the channelizer's rate against the read-only streaming rate of the same buffer,
 * measured in the same process on the same GPU. */
#ifndef SNOUT_BENCH_H
#define SNOUT_BENCH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid (bench.py, SURVEY.md §8d "measured-copy-peak"): read-only streaming rate, in GB/s,
 * of `bytes` bytes of device memory at `dev` (a fully coalesced 16-byte-per-lane grid-stride read
 * kernel, `reps` timed launches after one warm-up, HIP events on `hip_stream`).  Not part of the
 * receive path; no reference counterpart. */
int  snout_bench_hbm_read_gbps(const void* dev, uint64_t bytes, uint32_t reps, void* hip_stream,
                         float* gbps_best, float* gbps_mean);

#ifdef __cplusplus
}
#endif
#endif
