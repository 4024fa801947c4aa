// Calibration: sustained fp32 / fp64 MFMA rate of this chip under its own DVFS (no memory traffic).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k64(double* out, int iters, double a, double b) {
    f64x4 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 4; ++e) acc[j][e] = (double)threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    double s = 0;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 4; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        const int grid = 256 * blocks_per_cu, iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 2;
            printf("f32 32x32x2: %d blocks/CU: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, flops / ms / 1e9);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        const int grid = 512, iters = 20000;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k64, dim3(grid), dim3(256), 0, 0, (double*)out, iters, 1.0, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 4 * iters * 4 * 2.0 * 16 * 16 * 4;
        printf("f64 16x16x4: %.3f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
    }
    return 0;
}
