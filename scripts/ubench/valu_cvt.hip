// Issue rate of the conversion instructions the split-form epilogues are made of, against v_fma_f32 / v_pk_fma_f32.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_cvt.hip -o scripts/ubench/valu_cvt
#include <hip/hip_runtime.h>
#include <stdio.h>

// 16 independent chains of one instruction kind per iteration
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[16];
    for (int j = 0; j < 16; ++j) a[j] = out[threadIdx.x + 256 * j] + (float)j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[j]));
            if (MODE == 1) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[j]));
            if (MODE == 2) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[j]));
            if (MODE == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(a[j]));
            if (MODE == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[j]));
            if (MODE == 5) asm volatile("v_max_f32 %0, %0, %0" : "+v"(a[j]));
            if (MODE == 6) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[j]));
            if (MODE == 7) asm volatile("v_fma_mix_f32 %0, %0, %0, %0 op_sel_hi:[1,0,0]" : "+v"(a[j]));
            if (MODE == 8) asm volatile("v_fma_mixlo_f16 %0, %0, %0, %0" : "+v"(a[j]));
            if (MODE == 9) asm volatile("v_mov_b32 %0, %0" : "+v"(a[j]));
            if (MODE == 10) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a[j]));
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
__global__ __launch_bounds__(256) void k2(float* out, int iters) {     // 64-bit operands
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a[16];
    for (int j = 0; j < 16; ++j) a[j] = f32x2{out[threadIdx.x + 256 * j] + (float)j, 1.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[j]));
            if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a[j]));
            if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(a[j]));
            if (MODE == 3) asm volatile("v_pk_mov_b32 %0, %0, %0" : "+v"(a[j]));
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += a[j][0] + a[j][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int wps = 4, grid = 256 * wps, iters = 20000;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        launch(grid, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wave-instructions per SIMD: wps waves x iters x 16; cycles at a nominal 2.4 GHz
    const double inst = (double)wps * iters * 16, cyc = best * 1e-3 * 2.4e9;
    printf("%-28s %.3f ms  %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, best, cyc / inst);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 8192 * 256 * 4);
    (void)hipMemset(out, 0, 8192 * 256 * 4);
#define R1(M, NAME) run(NAME, [&](int g, int it) { hipLaunchKernelGGL(k<M>, dim3(g), dim3(256), 0, 0, out, it); })
#define R2(M, NAME) run(NAME, [&](int g, int it) { hipLaunchKernelGGL(k2<M>, dim3(g), dim3(256), 0, 0, out, it); })
    R1(0, "v_fma_f32"); R1(1, "v_cvt_f32_f16"); R1(2, "v_cvt_f16_f32"); R1(3, "v_cvt_pk_f16_f32"); R1(4, "v_cvt_f32_f16 sdwa");
    R1(5, "v_max_f32"); R1(6, "v_max3_f32"); R1(7, "v_fma_mix_f32"); R1(8, "v_fma_mixlo_f16"); R1(9, "v_mov_b32"); R1(10, "v_cndmask_b32");
    R2(0, "v_pk_fma_f32"); R2(1, "v_pk_mul_f32"); R2(2, "v_pk_add_f32"); R2(3, "v_pk_mov_b32");
    return 0;
}
