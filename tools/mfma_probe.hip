// Dev tool: is v_mfma_f32_16x16x4_f32 usable for the channelizer's FIR under the bit-exact contract?
//   1. numerics: D = A B + C against fmaf chains in ascending / descending k and a pairwise tree
//   2. zeros in A leave the accumulator untouched; denormal inputs / outputs
//   3. throughput: MFMA-only waves, VALU-only waves, and both kinds on the same SIMDs
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o build/mfma_probe && build/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

// one wave: K = 4 * steps; A[16][K], B[K][16] row-major in global memory
__global__ void k_mfma(const float* A, const float* B, float* D, int steps, int K)
{
    const int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; s++) {
        const float a = A[(l % 16) * K + 4 * s + l / 16];      // A[row = l%16][k = l/16]
        const float b = B[(4 * s + l / 16) * 16 + l % 16];     // B[k = l/16][col = l%16]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) D[(4 * (l / 16) + i) * 16 + l % 16] = acc[i];   // D[4*(l/16)+i][l%16]
}

constexpr int ITERS = 2048;
__global__ __launch_bounds__(256) void k_mix(float* out, int mode)
{
    // mode 0: every wave MFMA; 1: every wave VALU; 2: even waves MFMA, odd waves VALU
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1) == 0);
    float s = 0;
    if (do_mfma) {
        v4f acc[4];
        for (int i = 0; i < 4; i++) acc[i] = v4f{(float)threadIdx.x, 1, 2, 3};
        float a = 1.0001f, b = 0.9999f;
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        float acc[32];
        for (int i = 0; i < 32; i++) acc[i] = (float)(threadIdx.x + i);
        float a = 1.0000001f, b = 0.9999999f;
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        for (int i = 0; i < 32; i++) s += acc[i];
    }
    if (s == 12345.678f) out[0] = s;
}

static float ms_of(int mode, int wg_per_cu)
{
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mix, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, mode);
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mix, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipFree(out);
    return best;
}

int main()
{
    const int steps = 10, K = 4 * steps;
    std::vector<float> A(16 * K), B(K * 16), D(256);
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    srand(1);
    long asc = 0, desc = 0, tree = 0, total = 0, zero_ok = 0, zero_tot = 0;
    for (int trial = 0; trial < 200; trial++) {
        for (auto& v : A) v = (float)rand() / RAND_MAX - 0.5f;
        for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * (trial % 3 == 0 ? 1e-3f : 1.0f);
        if (trial % 2) for (int r = 0; r < 16; r++) for (int k = 0; k < K; k++) if ((k + r) % 3) A[r * K + k] = 0.0f;   // banded-like zeros
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps, K);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            float c1 = 0, c2 = 0, c3 = 0, cz = 0;
            for (int k = 0; k < K; k++) c1 = fmaf(A[i * K + k], B[k * 16 + j], c1);
            for (int s = 0; s < steps; s++) for (int k = 3; k >= 0; k--) c2 = fmaf(A[i * K + 4 * s + k], B[(4 * s + k) * 16 + j], c2);
            for (int s = 0; s < steps; s++) {
                float p0 = A[i * K + 4 * s] * B[(4 * s) * 16 + j], p1 = A[i * K + 4 * s + 1] * B[(4 * s + 1) * 16 + j];
                float p2 = A[i * K + 4 * s + 2] * B[(4 * s + 2) * 16 + j], p3 = A[i * K + 4 * s + 3] * B[(4 * s + 3) * 16 + j];
                c3 = c3 + ((p0 + p1) + (p2 + p3));
            }
            for (int k = 0; k < K; k++) if (A[i * K + k] != 0.0f) cz = fmaf(A[i * K + k], B[k * 16 + j], cz);   // zeros skipped
            const float d = D[i * 16 + j];
            total++; asc += d == c1; desc += d == c2; tree += d == c3;
            if (trial % 2) { zero_tot++; zero_ok += d == cz; }
        }
    }
    printf("mfma_f32_16x16x4f32 == fmaf chain ascending k: %ld/%ld, descending within a step: %ld, pairwise tree: %ld; zeros skipped: %ld/%ld\n",
           asc, total, desc, tree, zero_ok, zero_tot);
    // denormals
    for (auto& v : A) v = 0; for (auto& v : B) v = 0;
    A[0] = 1e-39f; B[0] = 1.0f;          // denormal input a
    A[1 * K + 0] = 1e-20f; B[1] = 1e-20f; // product denormal (1e-40) at D[1][1]? uses B[0*16+1]
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps, K);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    printf("denormal input 1e-39 * 1 -> %g (fmaf: %g); 1e-20 * 1e-20 -> %g (fmaf: %g)\n", D[0], fmaf(1e-39f, 1.0f, 0.0f), D[1 * 16 + 1], fmaf(1e-20f, 1e-20f, 0.0f));
    // throughput
    for (int w : {1, 2, 4}) {
        const float m0 = ms_of(0, w), m1 = ms_of(1, w), m2 = ms_of(2, w);
        const double fl_m = 2048.0 * 8 * ITERS * 4 * 256.0 * w, fl_v = 128.0 * 32 * ITERS * 4 * 256.0 * w;
        printf("%d WG/CU: MFMA-only %.3f ms (%.1f TF)  VALU-only %.3f ms (%.1f TF)  half/half %.3f ms (expected if serial %.3f, if parallel %.3f)\n",
               w, m0, fl_m / m0 / 1e9, m1, fl_v / m1 / 1e9, m2, (m0 + m1) / 2, (m0 > m1 ? m0 : m1) / 2);
    }
    return 0;
}
