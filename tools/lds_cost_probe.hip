// Dev tool: what an LDS read costs a SIMD that is busy with packed FMAs (the channelizer's FIR waves: 23 ds_read_b64 per 128
// v_pk_fma_f32).  One 256-thread workgroup = one wave per SIMD; 4 workgroups per CU = the channelizer's four waves per SIMD.
// Per loop body: 32 v_pk_fma_f32 on 8 independent accumulators + a number of conflict-free LDS reads whose results feed the FMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_cost_probe.hip -o build/lds_cost_probe && build/lds_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>

using f2 = __attribute__((ext_vector_type(2))) float;
using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int ITERS = 8192;

// MODE 0: no LDS; 1: 6 x ds_read_b64; 2: 3 x ds_read_b128 (the same bytes); 3: 6 x ds_read_b128; 4: 12 x ds_read_b64;
// 5: 6 x ds_read_b64 at a 320-byte lane stride... (lanes consecutive: the FIR's pattern is lane-consecutive 8 bytes)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a)
{
    __shared__ f2 buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = f2{(float)i, 1.0f};
    __syncthreads();
    f2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = f2{(float)(threadIdx.x + i), (float)i};
    f2 av = {a, a};
    const unsigned base8 = (unsigned)(size_t)&buf[0] + 8u * threadIdx.x;           // 8 bytes per lane, consecutive
    const unsigned base16 = (unsigned)(size_t)&buf[0] + 16u * threadIdx.x;         // 16 bytes per lane, consecutive
    constexpr int N64 = MODE == 1 ? 6 : MODE == 4 ? 12 : 0;
    constexpr int N128 = MODE == 2 ? 3 : MODE == 3 ? 6 : 0;
    f2 w[12];
    f4 q[6];
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = av;
#pragma unroll
    for (int i = 0; i < 6; i++) q[i] = f4{a, a, a, a};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < N64; i++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w[i]) : "v"(base8), "n"(i * 2048) : "memory");
#pragma unroll
        for (int i = 0; i < N128; i++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(base16), "n"(i * 4096) : "memory");
        // the FMAs of this iteration use the values read in the one before (the reads have a whole body to land)
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (N64)       asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(w[(i + j) % (N64 ? N64 : 1)]));
                else if (N128) { f2 h = (i & 1) ? f2{q[(i / 2 + j) % (N128 ? N128 : 1)].z, q[(i / 2 + j) % (N128 ? N128 : 1)].w} : f2{q[(i / 2 + j) % (N128 ? N128 : 1)].x, q[(i / 2 + j) % (N128 ? N128 : 1)].y};
                                 asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(h)); }
                else           asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(av));
            }
        }
        if (N64)  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]) :: "memory");
        if (N128) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]) :: "memory");
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <class K> static void run(const char* name, K kern, int wg_per_cu)
{
    float* out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 1.0000001f);
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 1.0000001f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: wg_per_cu waves, each ITERS bodies of 32 packed FMAs
    const double ns_per_body = best * 1e6 / ITERS;
    printf("%-28s %d waves/SIMD: %.3f ms  %.1f ns per body and wave  (%.2f ns per packed FMA and SIMD)\n", name, wg_per_cu, best,
           ns_per_body, ns_per_body / (32.0 * wg_per_cu));
    (void)hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run("32 pk_fma", k<0>, w);
        run("32 pk_fma + 6 ds_read_b64", k<1>, w);
        run("32 pk_fma + 3 ds_read_b128", k<2>, w);
        run("32 pk_fma + 6 ds_read_b128", k<3>, w);
        run("32 pk_fma + 12 ds_read_b64", k<4>, w);
    }
    return 0;
}
