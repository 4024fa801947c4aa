// v_pk_fma_f32 issue rate by operand form: all-VGPR vs one SGPR-pair operand, 1-8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_pk.hip -o scripts/ubench/valu_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: all VGPR   1: multiplier from SGPRs (wave-uniform)   2: plain v_fma_f32 all VGPR
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* __restrict__ wts) {
    f32x2 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x2{(float)threadIdx.x * 1e-3f + j, 1.f};
    f32x2 xv = {out[threadIdx.x], out[threadIdx.x + 256]};
    f32x2 wv = {out[threadIdx.x + 512] + 1.0001f, out[threadIdx.x + 768] + 0.9999f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1) {
            const f32x2 w = {wts[i & 7], wts[(i & 7) + 8]};
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = __builtin_elementwise_fma(w, xv, acc[j]);
        } else if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = __builtin_elementwise_fma(wv, xv, acc[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc[j].x = fmaf(wv.x, xv.x, acc[j].x);
                acc[j].y = fmaf(wv.y, xv.y, acc[j].y);
            }
        }
        asm volatile("" : "+v"(xv));
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc[j].x + acc[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *out, *w;
    (void)hipMalloc(&out, 8192 * 256 * 4);
    (void)hipMalloc(&w, 256);
    (void)hipMemset(out, 0, 8192 * 256 * 4);
    (void)hipMemset(w, 0, 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const char* names[3] = {"v_pk_fma_f32 all VGPR", "v_pk_fma_f32 SGPR multiplier", "v_fma_f32 all VGPR"};
    for (int mode = 0; mode < 3; ++mode)
        for (int wps = 1; wps <= 8; wps *= 2) {
            const int grid = 256 * wps, iters = 10000;
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, w);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, w);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, w);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double flops = (double)grid * 256 * iters * 32 * 2.0;
            printf("%-30s %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", names[mode], wps, best, flops / best / 1e9);
        }
    return 0;
}
