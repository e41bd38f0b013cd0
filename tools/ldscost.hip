// Dev tool: what does an LDS instruction cost a VALU-bound wave?  Each kernel runs the same 64-FMA block
// per iteration plus NOPS LDS instructions of one kind (independent addresses, conflict-free), at 1..4
// workgroups of 256 threads per CU.  cost = (t_mix - t_fma_only) / (LDS instructions per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/ldscost.hip -o build/ldscost && build/ldscost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2048;

template <int KIND, int NOPS>
__global__ __launch_bounds__(256) void k(float* out, float a, float b)
{
    __shared__ float4 buf[256 * 9];
    for (int i = threadIdx.x; i < 256 * 9; i += 256) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = (float)(threadIdx.x + i);
    const uint32_t base = (uint32_t)(uintptr_t)&buf[threadIdx.x];      // 16 B per lane, conflict-free
    const uint32_t base8 = (uint32_t)(uintptr_t)((float2*)buf + threadIdx.x);
    const uint32_t base4 = (uint32_t)(uintptr_t)((float*)buf + threadIdx.x);
    v4f r4 = {0, 0, 0, 0}; v2f r2 = {0, 0}; float r1 = 0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int g = 0; g < 8; g++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[4 * g + i]) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[(4 * g + i + 16) & 31]) : "v"(b), "v"(a));
            }
            if (g < NOPS) {
                if (KIND == 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r1) : "v"(base4), "n"(1024 * (g % 8)) : "memory");
                if (KIND == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2) : "v"(base8), "n"(2048 * (g % 8)) : "memory");
                if (KIND == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4) : "v"(base), "n"(4096 * (g % 8)) : "memory");
                if (KIND == 4) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(base4), "v"(acc[g]), "n"(1024 * (g % 8)) : "memory");
                if (KIND == 5) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(base8), "v"(r2), "n"(2048 * (g % 8)) : "memory");
                if (KIND == 6) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(base), "v"(r4), "n"(4096 * (g % 8)) : "memory");
                if (KIND == 7) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r1) : "v"(base4), "v"(acc[g]) : "memory");
                if (KIND == 8) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r4) : "v"(base8), "n"(2 * (g % 8)), "n"(2 * (g % 8) + 32) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = r1 + r2.x + r2.y + r4.x + r4.y + r4.z + r4.w;
#pragma unroll
    for (int i = 0; i < 32; i++) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

template <class K> static float run(K kern, int w)
{
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256 * w), dim3(256), 0, 0, out, 1.0000001f, 0.9999999f);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256 * w), dim3(256), 0, 0, out, 1.0000001f, 0.9999999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipFree(out);
    return best;
}

#define ROW(KIND, NAME)                                                                                   \
    {                                                                                                     \
        const float t4 = run(k<KIND, 4>, w), t8 = run(k<KIND, 8>, w);                                      \
        /* per SIMD: w waves x ITERS x NOPS instructions; clock ~2.0 GHz assumed for the cycle figure */   \
        printf("  %-14s 4 per 64 FMA: %.3f ms (+%.0f cyc/op)   8 per 64 FMA: %.3f ms (+%.0f cyc/op)\n", NAME, t4, \
               (t4 - t0) * 1e-3 * 2.0e9 / ((double)w * ITERS * 4), t8, (t8 - t0) * 1e-3 * 2.0e9 / ((double)w * ITERS * 8)); \
    }

int main()
{
    for (int w : {1, 2, 4}) {
        const float t0 = run(k<0, 0>, w);
        printf("%d waves/SIMD: 64 FMA block alone %.3f ms (%.2f cyc per FMA per SIMD at 2.0 GHz)\n", w, t0,
               t0 * 1e-3 * 2.0e9 / ((double)w * ITERS * 64));
        ROW(1, "ds_read_b32") ROW(2, "ds_read_b64") ROW(8, "ds_read2_b64") ROW(3, "ds_read_b128")
        ROW(4, "ds_write_b32") ROW(5, "ds_write_b64") ROW(6, "ds_write_b128") ROW(7, "ds_bpermute_b32")
    }
    return 0;
}
