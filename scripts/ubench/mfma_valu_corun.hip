// Do the matrix pipe and the vector ALU of a SIMD run concurrently when fed by different waves?
// Workgroup of 512 threads: waves 0-3 (one per SIMD) loop over v_mfma_f32_32x32x2_f32, waves 4-7 over
// v_pk_fma_f32; each role is also run alone.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_corun.hip -o scripts/ubench/mfma_valu_corun
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(float* out, int iters_m, int iters_v, int mode) {
    const int w = threadIdx.x >> 6;
    float s = 0.f;
    if (w < 4) {
        if (mode & 1) {
            f32x16 acc[4];
            for (int j = 0; j < 4; ++j)
                for (int e = 0; e < 16; ++e) acc[j][e] = (float)threadIdx.x;
            for (int i = 0; i < iters_m; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, 0.5f, acc[j], 0, 0, 0);
            for (int j = 0; j < 4; ++j)
                for (int e = 0; e < 16; ++e) s += acc[j][e];
        }
    } else if (mode & 2) {
        f32x2 a[16];
        for (int j = 0; j < 16; ++j) a[j] = f32x2{(float)threadIdx.x, 1.f + j};
        const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
        for (int i = 0; i < iters_v; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = __builtin_elementwise_fma(a[j], m, c);
        for (int j = 0; j < 16; ++j) s += a[j].x + a[j].y;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out;
    (void)hipMalloc(&out, 512 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters_m = 20000, iters_v = 40000;
    const char* names[4] = {"", "MFMA waves only", "VALU waves only", "both"};
    for (int mode = 1; mode <= 3; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters_m, iters_v, mode);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double fm = (mode & 1) ? 256.0 * 4 * iters_m * 4 * 2.0 * 32 * 32 * 2 : 0;
        const double fv = (mode & 2) ? 256.0 * 4 * 64 * (double)iters_v * 16 * 2 * 2 : 0;
        printf("%-18s %.3f ms   MFMA %.1f TFLOP/s   VALU %.1f TFLOP/s\n", names[mode], best, fm / best / 1e9, fv / best / 1e9);
    }
    return 0;
}
