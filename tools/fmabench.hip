// Dev tool: f32 VALU issue ceilings on this device -- does v_pk_fma_f32 buy anything over v_fma_f32
// on gfx950, and what does an FMA stream sustain at the channelizer's occupancy (3-4 waves / SIMD)?
//   hipcc -O3 --offload-arch=gfx950 tools/fmabench.hip -o /tmp/fmabench && /tmp/fmabench
// Every kernel runs ITERS x 64 dependent-chain-free FMA instructions per wave on 16 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>

using f2 = __attribute__((ext_vector_type(2))) float;
constexpr int ITERS = 4096;

// scalar FMAs: 32 independent accumulators, 64 v_fma_f32 per loop body
__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b)
{
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 32; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < 32; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(a));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

// packed FMAs: 16 independent 2-wide accumulators, 32 v_pk_fma_f32 per loop body (= 64 scalar FMAs)
__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b)
{
    f2 acc[16];
    f2 av = {a, a}, bv = {b, b};
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = f2{(float)(threadIdx.x + i), (float)i};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(bv), "v"(av));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i].x + acc[i].y;
    if (s == 12345.678f) out[0] = s;
}

// scalar FMAs with one ds_read_b64 per 8 FMAs (roughly the FIR's mix: 23 reads per 256 FMAs is lighter)
__global__ __launch_bounds__(256) void k_fma_lds(float* out, float a, float b)
{
    __shared__ float2 buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = make_float2((float)i, 1.0f);
    __syncthreads();
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = (float)(threadIdx.x + i);
    int idx = threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const float2 w = buf[(idx + 41 * g) & 2047];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc[4 * g + i] = __builtin_fmaf(a, w.x, acc[4 * g + i]);
                acc[(4 * g + i + 16) & 31] = __builtin_fmaf(b, w.y, acc[(4 * g + i + 16) & 31]);
            }
        }
        idx += 7;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

template <class K> static void run(const char* name, K kern, int blocks_per_cu, double fma_per_thread)
{
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 1.0000001f, 0.9999999f);
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 1.0000001f, 0.9999999f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2.0 * fma_per_thread * 256.0 * grid;
    printf("%-12s %d WG/CU (%d waves/SIMD): %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, blocks_per_cu,
           best, flop / best / 1e9);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 3, 4, 8}) {
        run("v_fma_f32", k_fma, w, 64.0 * ITERS);
        run("v_pk_fma_f32", k_pkfma, w, 64.0 * ITERS);
        run("fma+ds_read", k_fma_lds, w, 64.0 * ITERS);
    }
    return 0;
}
