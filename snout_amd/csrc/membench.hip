// membench.hip — read-only streaming rate of a device buffer, measured on the caller's GPU in the
// caller's process.  bench.py reports the dominant kernel's algorithmic GB/s both against the 8 TB/s
// HBM3E spec peak and against this measured ceiling (SURVEY.md §8d "report both spec-peak and
// measured-copy-peak fractions").  Not part of the receive path.
#include "common.h"

namespace snout {

using f4 = __attribute__((ext_vector_type(4))) float;

// fully coalesced grid-stride read: every wave instruction fetches 1 KiB, UNROLL loads in flight
template <int UNROLL>
__global__ __launch_bounds__(256) void hbm_read(const f4* __restrict__ p, uint64_t n4, float* __restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * 256u * UNROLL + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256u * UNROLL;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (; i + 256u * (UNROLL - 1) < n4; i += stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = p[i + 256u * k];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) acc += v[k];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;      // never true: keeps the loads
}

}  // namespace snout

using namespace snout;

extern "C" int snout_bench_hbm_read_gbps(const void* dev, uint64_t bytes, uint32_t reps, void* hip_stream,
                                   float* gbps_best, float* gbps_mean)
{
    if (!dev || bytes < (1u << 20) || !gbps_best) return SNOUT_EINVAL;
    hipStream_t st = (hipStream_t)hip_stream;
    const uint32_t grid = 8192;
    float* out = nullptr;
    SNOUT_HIP(hipMalloc((void**)&out, grid * sizeof(float)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SNOUT_OK;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = SNOUT_EHIP;
    const uint64_t n4 = bytes / 16u;
    float best = 0.0f, sum = 0.0f;
    if (!reps) reps = 1;
    for (uint32_t r = 0; r < reps + 1u && rc == SNOUT_OK; r++) {       // first launch is a warm-up
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL((hbm_read<8>), dim3(grid), dim3(256), 0, st, (const f4*)dev, n4, out);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { rc = SNOUT_EHIP; break; }
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r == 0 || ms <= 0.0f) continue;
        const float g = (float)((double)(n4 * 16u) / (ms * 1e-3) / 1e9);
        if (g > best) best = g;
        sum += g;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(out);
    *gbps_best = best;
    if (gbps_mean) *gbps_mean = sum / (float)reps;
    return rc;
}
