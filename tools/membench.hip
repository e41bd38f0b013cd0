// Dev tool: read-only streaming ceilings on this device, for the access shapes btle_demod_corr could use.
//   hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f4 = __attribute__((ext_vector_type(4))) float;

// A: fully coalesced: each wave instruction reads 1 KiB contiguous; UNROLL independent loads in flight
template <int UNROLL>
__global__ __launch_bounds__(256) void rd_coalesced(const f4* __restrict__ p, size_t n4, float* out)
{
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    f4 acc = {0, 0, 0, 0};
    for (; i + 256 * (UNROLL - 1) < n4; i += stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = p[i + 256 * k];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) acc += v[k];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

// B: lane stride 32 B (two instructions cover a contiguous 2 KiB), wave-contiguous chunks like K1
__global__ __launch_bounds__(256) void rd_stride32(const f4* __restrict__ p, size_t n4, float* out, int iters)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t lane = threadIdx.x & 63;
    const f4* base = p + wave * (size_t)iters * 128;       // 2 KiB per iteration
    f4 acc = {0, 0, 0, 0};
    if ((wave + 1) * (size_t)iters * 128 <= n4)
        for (int r = 0; r < iters; r++) {
            f4 a = base[(size_t)r * 128 + lane * 2], b = base[(size_t)r * 128 + lane * 2 + 1];
            acc += a + b;
        }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

// C: wave-contiguous chunks, coalesced instructions (lane stride 16 B), 4 loads per iteration
__global__ __launch_bounds__(256) void rd_chunk_coal(const f4* __restrict__ p, size_t n4, float* out, int iters)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t lane = threadIdx.x & 63;
    const f4* base = p + wave * (size_t)iters * 256;       // 4 KiB per iteration
    f4 acc = {0, 0, 0, 0};
    if ((wave + 1) * (size_t)iters * 256 <= n4)
        for (int r = 0; r < iters; r++) {
            f4 a = base[(size_t)r * 256 + lane], b = base[(size_t)r * 256 + 64 + lane];
            f4 c = base[(size_t)r * 256 + 128 + lane], d = base[(size_t)r * 256 + 192 + lane];
            acc += a + b + c + d;
        }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

template <class F> float timeit(F f, int reps = 10)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    float best = 1e9;
    for (int i = 0; i < reps; i++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}

int main()
{
    const size_t bytes = 8ull << 30;
    f4* p; float* out;
    hipMalloc(&p, bytes); hipMalloc(&out, 1 << 20);
    hipMemset(p, 1, bytes);
    const size_t n4 = bytes / 16;
    for (int grid : {2048, 4096, 8192, 16384}) {
        float t4 = timeit([&] { rd_coalesced<4><<<grid, 256>>>(p, n4, out); });
        float t8 = timeit([&] { rd_coalesced<8><<<grid, 256>>>(p, n4, out); });
        printf("coalesced grid-stride grid=%5d: unroll4 %.3f ms %.0f GB/s | unroll8 %.3f ms %.0f GB/s\n", grid,
               t4, bytes / t4 / 1e6, t8, bytes / t8 / 1e6);
    }
    for (int iters : {16, 64, 256}) {
        int gridB = (int)(n4 / ((size_t)iters * 128) / 4), gridC = (int)(n4 / ((size_t)iters * 256) / 4);
        float tb = timeit([&] { rd_stride32<<<gridB, 256>>>(p, n4, out, iters); });
        float tc = timeit([&] { rd_chunk_coal<<<gridC, 256>>>(p, n4, out, iters); });
        printf("per-wave chunks iters=%3d: stride32 %.3f ms %.0f GB/s | coalesced %.3f ms %.0f GB/s\n", iters,
               tb, bytes / tb / 1e6, tc, bytes / tc / 1e6);
    }
    return 0;
}
