// Dev probe: cost of fetching one of 129 eight-float rows per lane (lane-random row) from LDS on gfx950, 12 waves per CU:
// two float4 arrays (2 ds_read_b128), eight float planes (8 ds_read_b32), four float2 planes (4 ds_read_b64), and two float4
// arrays replicated four times (lane & 3 picks the copy).   hipcc --offload-arch=gfx950 -O3 tools/lds_taps_probe.hip -o build/scratch/tapsprobe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* __restrict__ out, int iters, unsigned long long* cyc)
{
    __shared__ float4 A[4][132], B[4][132];
    __shared__ float P[8][132];
    __shared__ float2 Q[4][132];
    for (int i = threadIdx.x; i < 129; i += 256)
        for (int c = 0; c < 4; c++) {
            A[c][i] = make_float4(i, i + 1, i + 2, i + 3); B[c][i] = make_float4(i + 4, i + 5, i + 6, i + 7);
            Q[c][i] = make_float2(i + 2 * c, i + 2 * c + 1);
        }
    for (int i = threadIdx.x; i < 129; i += 256) for (int k = 0; k < 8; k++) P[k][i] = i + k;
    __syncthreads();
    unsigned r = threadIdx.x * 2654435761u + blockIdx.x;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        r = r * 1664525u + 1013904223u;
        const int imu = (r >> 8) % 129u;
        float v[8];
        if (MODE == 0) { const float4 a = A[0][imu], b = B[0][imu]; v[0]=a.x;v[1]=a.y;v[2]=a.z;v[3]=a.w;v[4]=b.x;v[5]=b.y;v[6]=b.z;v[7]=b.w; }
        else if (MODE == 1) { for (int k = 0; k < 8; k++) v[k] = P[k][imu]; }
        else if (MODE == 2) { for (int k = 0; k < 4; k++) { const float2 q = Q[k][imu]; v[2*k] = q.x; v[2*k+1] = q.y; } }
        else { const int c = threadIdx.x & 3; const float4 a = A[c][imu], b = B[c][imu]; v[0]=a.x;v[1]=a.y;v[2]=a.z;v[3]=a.w;v[4]=b.x;v[5]=b.y;v[6]=b.z;v[7]=b.w; }
#pragma unroll
        for (int k = 0; k < 8; k++) acc = __builtin_fmaf(v[k], (float)(k + 1), acc);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    const int blocks = 256 * 3, iters = 4000;
    float* o; unsigned long long* d_c; hipMalloc(&o, blocks * 1024); hipMalloc(&d_c, 64);
    unsigned long long c[4];
    hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, o, iters, d_c);
    hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, o, iters, d_c + 1);
    hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, o, iters, d_c + 2);
    hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, o, iters, d_c + 3);
    hipDeviceSynchronize();
    hipMemcpy(c, d_c, 32, hipMemcpyDeviceToHost);
    printf("cycles per lane-random 8-float row, 12 waves per CU: 2 x b128 %.1f | 8 x b32 planes %.1f | 4 x b64 planes %.1f | 2 x b128, four copies %.1f\n",
           (double)c[0] / iters, (double)c[1] / iters, (double)c[2] / iters, (double)c[3] / iters);
    return 0;
}
