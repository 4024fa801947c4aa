// Calibration: sustained fp32 VALU FMA rate (plain v_fmac with an SGPR operand vs v_pk_fma_f32).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_fma(float* out, int iters, const float* __restrict__ wts) {
    float acc[32];
    for (int j = 0; j < 32; ++j) acc[j] = (float)threadIdx.x * 1e-3f + j;
    const float x = out[threadIdx.x];
    for (int i = 0; i < iters; ++i) {
        const float w0 = wts[i & 7];   // wave-uniform -> SGPR
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = fmaf(w0, x + (float)j, acc[j]);
    }
    float s = 0;
    for (int j = 0; j < 32; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk(float* out, int iters, const float* __restrict__ wts) {
    f32x2 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x2{(float)threadIdx.x * 1e-3f + j, 1.f};
    const float x = out[threadIdx.x];
    for (int i = 0; i < iters; ++i) {
        const f32x2 w = {wts[i & 7], wts[(i + 1) & 7]};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const f32x2 xv = {x + (float)j, x + (float)j};
            acc[j] = __builtin_elementwise_fma(w, xv, acc[j]);
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc[j].x + acc[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *out, *w;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&w, 64);
    hipMemset(out, 0, 4096 * 256 * 4);
    hipMemset(w, 0, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8, iters = 20000;
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, iters, w);
            else hipLaunchKernelGGL(k_pk, dim3(grid), dim3(256), 0, 0, out, iters, w);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)grid * 256 * iters * 32 * 2.0;
            printf("%s: %.3f ms  %.1f TFLOP/s\n", which ? "v_pk_fma_f32" : "v_fmac_f32 ", ms, flops / ms / 1e9);
        }
    return 0;
}
