// Dev probe: does ds_read_b128 work on a dword-aligned (not 16-byte-aligned) LDS address on gfx950, and what does it cost
// against four ds_read2_b32?   hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned_probe.hip -o build/scratch/ldsprobe && build/scratch/ldsprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void probe(const int* __restrict__ offs, float* __restrict__ out, int iters, unsigned long long* cyc)
{
    __shared__ float buf[256 * 41];
    for (int i = threadIdx.x; i < 256 * 41; i += 256) buf[i] = (float)i;
    __syncthreads();
    const int l = threadIdx.x;
    int off = offs[l];
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const float* p = &buf[l * 41 + off];
        float v[8];
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = p[k];
        } else if (MODE == 2) {
            // sample-major: the wave's 64 lanes side by side in every row: bank = lane whatever the lane's offset
            const float* q = &buf[(l >> 6) * 64 * 41 + off * 64 + (l & 63)];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = q[k * 64];
        } else {
            float4 a, b;
            const unsigned addr = (unsigned)(uintptr_t)p;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k] * (float)(k + 1);
        off = (off + 2 + (l & 1)) % 32;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + l] = acc;
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    const int blocks = 256 * 3, iters = 2000;
    std::vector<int> h(256);
    for (int i = 0; i < 256; i++) h[i] = (i * 7) % 31;
    int* d_off; float *o0, *o1; unsigned long long* d_c;
    hipMalloc(&d_off, 1024); hipMalloc(&o0, blocks * 1024); hipMalloc(&o1, blocks * 1024); hipMalloc(&d_c, 16);
    hipMemcpy(d_off, h.data(), 1024, hipMemcpyHostToDevice);
    unsigned long long c0 = 0, c1 = 0;
    hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, d_off, o0, iters, d_c); hipMemcpy(&c0, d_c, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d_off, o1, iters, d_c + 1); hipMemcpy(&c1, d_c + 1, 8, hipMemcpyDeviceToHost);
    unsigned long long c2 = 0;
    hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, d_off, o1, iters, d_c); hipMemcpy(&c2, d_c, 8, hipMemcpyDeviceToHost);
    printf("sample-major rows (lane = bank): %.1f cycles per iteration\n", (double)c2 / iters);
    hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d_off, o1, iters, d_c + 1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("unaligned ds_read_b128 FAULTED\n"); return 1; }
    std::vector<float> a(blocks * 256), b(blocks * 256);
    hipMemcpy(a.data(), o0, blocks * 1024, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, blocks * 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
    printf("dword-aligned ds_read_b128: %s (%d of %zu differ); cycles per iteration: 8 x b32 %.1f, 2 x b128 %.1f\n", bad ? "WRONG" : "equal to ds_read_b32",
           bad, a.size(), (double)c0 / iters, (double)c1 / iters);
    return 0;
}
