// records.hip — packing of decoded records for the multi-GPU gather: one launch per
// collected segment copies the device copy of its records into the exchange buffer in the wire format
// (the first `width` bytes of each 160-byte snout_pkt: BTLE records use 24 + <= 42 bytes), marking the
// records a segment found in its pre-roll as disowned.  Replaces five torch operations per segment.
#include "common.h"

namespace snout {

__global__ __launch_bounds__(256) void pack_records(const uint4* __restrict__ src, uint64_t n, uint4* __restrict__ dst,
                                                    uint32_t w16, uint64_t own_from)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;       // one 16-byte piece
    if (i >= n * w16) return;
    const uint64_t rec = i / w16;
    const uint32_t piece = (uint32_t)(i % w16);
    uint4 v = src[rec * (sizeof(snout_pkt) / 16u) + piece];
    if (piece == 0u && own_from) {
        const uint64_t si = (uint64_t)v.x | ((uint64_t)v.y << 32);
        if (si < own_from) { v.x = 0u; v.y = 1u << 30; }                 // sample_index = 2^62
    }
    dst[i] = v;
}

int launch_pack_records(const snout_pkt* src, uint64_t n, void* dst, uint32_t width, uint64_t own_from, hipStream_t st)
{
    const uint32_t w16 = width / 16u;
    const uint64_t pieces = n * w16;
    hipLaunchKernelGGL(pack_records, dim3((uint32_t)((pieces + 255u) / 256u)), dim3(256), 0, st,
                       reinterpret_cast<const uint4*>(src), n, reinterpret_cast<uint4*>(dst), w16, own_from);
    SNOUT_HIP(hipGetLastError());
    return 0;
}

}  // namespace snout
