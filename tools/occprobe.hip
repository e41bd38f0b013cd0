// Residency probe (diagnostic, not part of the library): how many workgroups of a given shape does a
// gfx950 CU actually hold at once?  Every workgroup stamps s_memrealtime at entry and exit, spins for a
// fixed time in between and records which CU it ran on (HW_REG_XCC_ID, HW_REG_HW_ID); the host counts the
// maximum number of overlapping intervals per CU.  Swept over LDS bytes per workgroup, waves per
// workgroup and register budget (128 VGPRs = 4 wave slots per SIMD); a second table repeats the
// channelizer's own shape (5 waves, 128 VGPRs, grid = 2 or 3 per CU, with and without a barrier).
//   hipcc --offload-arch=gfx950 -O2 -o tools/occprobe.bin tools/occprobe.hip && tools/occprobe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

extern __shared__ char dyn_lds[];

template <bool BIG, bool BAR>
__global__ void probe(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (BAR) __syncthreads();
    if (BIG) asm volatile("v_mov_b32 v127, 0" ::: "v127");       // 128 VGPRs allocated
    if (ticks == 0x12345ull) dyn_lds[threadIdx.x] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t0;
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // XCC_ID
        out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_ID
    }
}

// the same with STATIC LDS and, optionally, ~96 SGPRs: the channelizer's kernel descriptor
template <int LDS, bool MANY_SGPR>
__global__ __launch_bounds__(320) void probe_static(unsigned long long* out, unsigned long long ticks)
{
    __shared__ char lds[LDS];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    asm volatile("v_mov_b32 v113, 0" ::: "v113");
    if (MANY_SGPR) asm volatile("s_mov_b32 s93, 0" ::: "s93");
    if (ticks == 0x12345ull) lds[threadIdx.x] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t0;
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
        out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    }
    if (ticks == 0x12346ull) out[5] = lds[threadIdx.x ^ 1];
}

template <int LDS>
__global__ void probe_static_nolb(unsigned long long* out, unsigned long long ticks)
{
    __shared__ char lds[LDS];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    asm volatile("v_mov_b32 v113, 0" ::: "v113");
    if (ticks == 0x12345ull) lds[threadIdx.x] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = t0; out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
    if (ticks == 0x12346ull) out[5] = lds[threadIdx.x ^ 1];
}
__global__ __launch_bounds__(320) void probe_dyn_lb(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    asm volatile("v_mov_b32 v113, 0" ::: "v113");
    if (ticks == 0x12345ull) dyn_lds[threadIdx.x] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = t0; out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
    if (ticks == 0x12346ull) out[5] = dyn_lds[threadIdx.x ^ 1];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static const unsigned long long kTicks = 20000;                  // 0.2 ms at 100 MHz

// HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
static inline uint64_t cu_key(const unsigned long long* rec) { return (rec[2] & 0xf) << 16 | (rec[3] & 0xff00); }

template <bool BAR>
static int shape_table(unsigned long long* d, std::vector<unsigned long long>& h, bool attr)
{
    if (attr) CK(hipFuncSetAttribute((const void*)probe<true, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int waves : {4, 5})
        for (int grid : {512, 768})
            for (int lds : {36512, 53920, 54944, 57504, 65536, 81920}) {
                if (!attr && lds > 65536) continue;
                hipLaunchKernelGGL((probe<true, BAR>), dim3(grid), dim3(waves * 64), lds, 0, d, kTicks);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), d, grid * 32, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, t1 = 0;
                for (int b = 0; b < grid; b++) { t0 = std::min(t0, h[b * 4]); t1 = std::max(t1, h[b * 4 + 1]); }
                std::map<uint64_t, int> first;
                int n_first = 0;
                for (int b = 0; b < grid; b++) {
                    const uint64_t key = cu_key(&h[b * 4]);
                    first[key] += 0;
                    if (h[b * 4] - t0 < kTicks / 4) { first[key]++; n_first++; }
                }
                std::map<int, int> hist;
                for (auto& kv : first) hist[kv.second]++;
                printf("waves %d barrier %d grid %4d lds %6d: rounds %.1f, started in the first quarter round %4d; CUs by that count:",
                       waves, (int)BAR, grid, lds, (double)(t1 - t0) / kTicks, n_first);
                for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
                printf("\n");
            }
    return 0;
}

template <int LDS, bool MANY_SGPR>
static int static_row(unsigned long long* d, std::vector<unsigned long long>& h, unsigned long long ticks)
{
    const int grid = 512;
    if (ticks & 1) CK(hipFuncSetAttribute((const void*)probe_static<LDS, MANY_SGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - LDS));
    hipLaunchKernelGGL((probe_static<LDS, MANY_SGPR>), dim3(grid), dim3(320), (ticks & 2) ? 64 : 0, 0, d, ticks);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, grid * 32, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < grid; b++) { t0 = std::min(t0, h[b * 4]); t1 = std::max(t1, h[b * 4 + 1]); }
    printf("static lds %6d, many sgprs %d, spin %llu ticks, grid 512 x 320 threads: rounds %.1f\n", LDS, (int)MANY_SGPR, ticks, (double)(t1 - t0) / ticks);
    return 0;
}

template <bool BIG>
static int sweep(unsigned long long* d, std::vector<unsigned long long>& h, int grid)
{
    CK(hipFuncSetAttribute((const void*)probe<BIG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int lds_kb[] = {0, 16, 20, 26, 27, 32, 36, 40, 45, 50, 53, 54, 58, 64, 80, 81, 100, 160};
    for (int waves : {1, 4, 5, 8})
        for (int kb : lds_kb) {
            const size_t lds = (size_t)kb * 1024;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((probe<BIG, false>), dim3(grid), dim3(waves * 64), lds, 0, d, kTicks);
            CK(hipGetLastError());
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d, grid * 32, hipMemcpyDeviceToHost));
            std::map<uint64_t, std::vector<std::pair<uint64_t, int>>> ev;
            for (int b = 0; b < grid; b++) {
                const uint64_t key = cu_key(&h[b * 4]);
                ev[key].push_back({h[b * 4 + 0], +1});
                ev[key].push_back({h[b * 4 + 1], -1});
            }
            int mx = 0, mn = 1 << 30;
            for (auto& kv : ev) {
                auto& v = kv.second;
                std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first != b.first ? a.first < b.first : a.second < b.second; });
                int cur = 0, best = 0;
                for (auto& p : v) { cur += p.second; best = std::max(best, cur); }
                mx = std::max(mx, best); mn = std::min(mn, best);
            }
            printf("%5d %6d %9zu | %6zu %8d %8d %8.1f\n", waves, BIG ? 128 : 0, lds, ev.size(), mx, mn, ms / 0.2);
        }
    return 0;
}

static int rounds_of(unsigned long long* d, std::vector<unsigned long long>& h, const char* what, int lds)
{
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 512 * 32, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < 512; b++) { t0 = std::min(t0, h[b * 4]); t1 = std::max(t1, h[b * 4 + 1]); }
    printf("%s lds %6d grid 512 x 320: rounds %.1f\n", what, lds, (double)(t1 - t0) / kTicks);
    return 0;
}

int main(int argc, char** argv)
{
    const int grid = 256 * 8;
    unsigned long long* d;
    CK(hipMalloc(&d, grid * 32));
    std::vector<unsigned long long> h(grid * 4);
    hipLaunchKernelGGL((probe_static_nolb<53920>), dim3(512), dim3(320), 0, 0, d, kTicks); if (rounds_of(d, h, "static, no launch bounds", 53920)) return 1;
    hipLaunchKernelGGL((probe_static_nolb<54944>), dim3(512), dim3(320), 0, 0, d, kTicks); if (rounds_of(d, h, "static, no launch bounds", 54944)) return 1;
    hipLaunchKernelGGL((probe_static_nolb<81920>), dim3(512), dim3(320), 0, 0, d, kTicks); if (rounds_of(d, h, "static, no launch bounds", 81920)) return 1;
    for (int lds : {53920, 54944, 65536}) { hipLaunchKernelGGL(probe_dyn_lb, dim3(512), dim3(320), lds, 0, d, kTicks); if (rounds_of(d, h, "dynamic, launch bounds 320", lds)) return 1; }
    if (shape_table<true>(d, h, false)) return 1;
    printf("-- attribute set\n");
    if (shape_table<true>(d, h, true)) return 1;
    for (unsigned long long ticks : {20000ull, 20002ull, 20001ull, 20003ull})      // bit 0: attribute set, bit 1: 64 B dynamic too
        if (static_row<36512, false>(d, h, ticks) || static_row<53920, false>(d, h, ticks) || static_row<54944, false>(d, h, ticks) ||
            static_row<57504, false>(d, h, ticks) || static_row<36512, true>(d, h, ticks) || static_row<53920, true>(d, h, ticks) ||
            static_row<54944, true>(d, h, ticks) || static_row<57504, true>(d, h, ticks) || static_row<81920, true>(d, h, ticks)) return 1;
    if (argc > 1) {
        printf("%5s %6s %9s | %6s %8s %8s %8s\n", "waves", "vgpr", "lds_B", "CUs", "max/CU", "min/CU", "rounds");
        if (sweep<false>(d, h, grid) || sweep<true>(d, h, grid)) return 1;
    }
    return 0;
}
