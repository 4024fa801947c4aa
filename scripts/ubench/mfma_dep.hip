// Does a chain of dependent v_mfma_f32_32x32x2_f32 (same accumulator back to back) issue at full rate?
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_dep.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NACC accumulators, CHAIN dependent MFMAs on one accumulator before moving to the next
template <int NACC, int CHAIN>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int CHAIN>
void run(float* out, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = 256 * bpc, iters = 20000 / CHAIN;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<NACC, CHAIN>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double flops = (double)grid * 4 * iters * NACC * CHAIN * 2.0 * 32 * 32 * 2;
        printf("%-28s %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", name, bpc, best, flops / best / 1e9);
    }
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * 8);
    run<4, 1>(out, "4 acc round-robin");
    run<5, 1>(out, "5 acc round-robin");
    run<1, 4>(out, "1 acc, dependent chain");
    run<5, 4>(out, "5 acc, chains of 4");
    run<2, 1>(out, "2 acc round-robin");
    return 0;
}
