// fp64 matrix-core ceiling on this chip: v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64, 1-8 independent accumulators,
// 1-4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f64.hip -o scripts/ubench/mfma_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(double* out, int iters, double a, double b) {
    f64x4 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 4; ++e) acc[j][e] = (double)threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    double s = 0;
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 4; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k4(double* out, int iters, double a, double b) {
    double acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = (double)threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j], 0, 0, 0);
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void run(K kern, const char* name, int nacc, double flops_per_instr, double* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        const int grid = 256 * bpc, iters = 8000;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double flops = (double)grid * 4 * iters * nacc * flops_per_instr;
        printf("%-22s acc=%d  %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", name, nacc, bpc, best, flops / best / 1e9);
    }
}

int main() {
    double* out;
    (void)hipMalloc(&out, 1024 * 256 * 8);
    run(k16<1>, "f64 16x16x4", 1, 2.0 * 16 * 16 * 4, out);
    run(k16<2>, "f64 16x16x4", 2, 2.0 * 16 * 16 * 4, out);
    run(k16<4>, "f64 16x16x4", 4, 2.0 * 16 * 16 * 4, out);
    run(k16<8>, "f64 16x16x4", 8, 2.0 * 16 * 16 * 4, out);
    run(k4<4>, "f64 4x4x4 (4 blocks)", 4, 2.0 * 4 * 4 * 4 * 4, out);
    run(k4<8>, "f64 4x4x4 (4 blocks)", 8, 2.0 * 4 * 4 * 4 * 4, out);
    return 0;
}
