// Dev tool: which compute units a CU-masked stream (hipExtStreamCreateWithCUMask) really runs on, per XCD.
//   hipcc --offload-arch=gfx950 -O2 -o build/cumask_probe tools/cumask_probe.hip && build/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <map>
__global__ void where(uint32_t* out)
{
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000ull) { }          // 20 us at 100 MHz
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
static void probe(const char* name, const std::vector<int>& bits)
{
    uint32_t mask[8] = {0};
    for (int b : bits) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int G = 4096;
    uint32_t* d; hipMalloc(&d, G * 8);
    hipLaunchKernelGGL(where, dim3(G), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<uint32_t> h(2 * G);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per;     // xcc -> {se, sh, cu}
    std::map<int, int> wgs;
    for (int i = 0; i < G; i++) { per[h[2 * i] & 15].insert((h[2 * i + 1] >> 8) & 0xff); wgs[h[2 * i] & 15]++; }
    printf("%-22s %3zu bits:", name, bits.size());
    int tot = 0;
    for (auto& kv : per) { printf(" x%d:%zu/%d", kv.first, kv.second.size(), wgs[kv.first]); tot += (int)kv.second.size(); }
    printf("  = %d CUs\n", tot);
    if (bits.size() <= 8) { for (auto& kv : per) { printf("    x%d:", kv.first); for (int c : kv.second) printf(" se%d.sh%d.cu%d", (c >> 5) & 7, (c >> 4) & 1, c & 15); printf("\n"); } }
    hipFree(d); hipStreamDestroy(st);
}
static std::vector<int> range(int a, int b, int step = 1) { std::vector<int> v; for (int i = a; i < b; i += step) v.push_back(i); return v; }
int main()
{
    probe("[0,256)", range(0, 256));
    probe("[0,128)", range(0, 128));
    probe("[128,256)", range(128, 256));
    probe("[0,144)", range(0, 144));
    probe("[144,256)", range(144, 256));
    probe("[0,160)", range(0, 160));
    probe("[0,32)", range(0, 32));
    probe("[0,8)", range(0, 8));
    probe("[8,16)", range(8, 16));
    probe("[32,40)", range(32, 40));
    probe("{0}", {0});
    probe("{1}", {1});
    probe("{8}", {8});
    probe("{0,64,128,192}", {0, 64, 128, 192});
    probe("even", range(0, 256, 2));
    return 0;
}
