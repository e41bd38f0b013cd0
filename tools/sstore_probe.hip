// Dev tool: do scalar stores (s_store_dwordx2 + s_dcache_wb) work on gfx950, and is a later global atomic OR on the same dword
// kept (the scalar data cache written back before it)?   hipcc -O3 --offload-arch=gfx950 tools/sstore_probe.hip -o build/sstore_probe
// Run under `timeout 30` (an unsupported opcode faults the queue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k(unsigned long long* out, int n)
{
    // one 64-bit value per wave from SGPRs: the ballot of (lane parity == wave parity)
    const int w = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (w >= n) return;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(((threadIdx.x ^ w) & 1) == 0) ^ ((unsigned long long)w << 16);
    unsigned long long* p = out + w;       // wave-uniform: an SGPR pair
    asm volatile("s_store_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" :: "s"(m), "s"(p) : "memory");
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicOr(p, 0x8000800080008000ull);
}

int main()
{
    const int n = 1 << 16;
    unsigned long long* d;
    (void)hipMalloc(&d, n * 8);
    (void)hipMemset(d, 0, n * 8);
    hipLaunchKernelGGL(k, dim3(n / 4), dim3(256), 0, 0, d, n);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    unsigned long long* h = new unsigned long long[n];
    (void)hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < n; w++) {
        const unsigned long long par = (w & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
        const unsigned long long want = (par ^ ((unsigned long long)w << 16)) | 0x8000800080008000ull;
        if (h[w] != want) { if (bad < 5) printf("wave %d: %016llx want %016llx\n", w, h[w], want); bad++; }
    }
    printf("%d of %d wrong\n", bad, n);
    return bad != 0;
}
