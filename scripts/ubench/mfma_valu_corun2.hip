// Which vector-ALU instructions run beside which MFMAs on one SIMD when they come from DIFFERENT waves?
// (mfma_valu_corun.hip answered "none" for v_mfma_f32_32x32x2_f32 beside v_pk_fma_f32 -- both of which use the fp32 FMA
//  datapath.  The kernels of this path issue fp16 MFMAs; their epilogues use packed fp32 arithmetic and conversions.)
// Workgroup of 512 threads: waves 0-3 (one per SIMD) loop over the MFMA, waves 4-7 over the vector instruction.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_corun2.hip -o scripts/ubench/mfma_valu_corun2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int MK, int VK>
__global__ __launch_bounds__(512) void k(float* out, int iters_m, int iters_v, int mode) {
    const int w = threadIdx.x >> 6;
    float s = 0.f;
    if (w < 4) {
        if (mode & 1) {
            if (MK == 0) {          // v_mfma_f32_32x32x2_f32
                f32x16 acc[4];
                for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = (float)threadIdx.x;
                for (int i = 0; i < iters_m; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, 0.5f, acc[j], 0, 0, 0);
                for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
            } else if (MK == 1) {   // v_mfma_f32_32x32x16_f16
                f32x16 acc[4];
                h8 a, b;
                for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); b[e] = (_Float16)(0.5f + e); }
                for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = (float)threadIdx.x;
                for (int i = 0; i < iters_m; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
                for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
            } else {                // v_mfma_f32_16x16x32_f16
                f32x4 acc[8];
                h8 a, b;
                for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); b[e] = (_Float16)(0.5f + e); }
                for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = (float)threadIdx.x;
                for (int i = 0; i < iters_m; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
                for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) s += acc[j][e];
            }
        }
    } else if (mode & 2) {
        if (VK == 0) {              // v_pk_fma_f32
            f32x2 a[16];
            for (int j = 0; j < 16; ++j) a[j] = f32x2{(float)threadIdx.x, 1.f + j};
            const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
            for (int i = 0; i < iters_v; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) a[j] = __builtin_elementwise_fma(a[j], m, c);
            for (int j = 0; j < 16; ++j) s += a[j].x + a[j].y;
        } else if (VK == 1) {       // v_fma_f32
            float a[16];
            for (int j = 0; j < 16; ++j) a[j] = (float)threadIdx.x + j;
            for (int i = 0; i < iters_v; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(1.0001f), "v"(1e-3f));
            for (int j = 0; j < 16; ++j) s += a[j];
        } else if (VK == 2) {       // conversions: v_cvt_pk_f16_f32 + v_cvt_f32_f16
            float a[16];
            for (int j = 0; j < 16; ++j) a[j] = (float)threadIdx.x + j;
            for (int i = 0; i < iters_v; ++i)
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    unsigned p;
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(a[j]), "v"(a[j + 1]));
                    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[j]) : "v"(p));
                }
            for (int j = 0; j < 16; ++j) s += a[j];
        } else {                    // v_max_f32
            float a[16];
            for (int j = 0; j < 16; ++j) a[j] = (float)threadIdx.x + j;
            for (int i = 0; i < iters_v; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[j]) : "v"(0.5f));
            for (int j = 0; j < 16; ++j) s += a[j];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MK, int VK>
static void run(float* out, const char* mn, const char* vn) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters_m = MK == 0 ? 10000 : 20000, iters_v = 40000;
    float t[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k<MK, VK>), dim3(256), dim3(512), 0, 0, out, iters_m, iters_v, mode);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        t[mode] = best;
    }
    printf("%-26s beside %-30s: MFMA alone %.3f ms, vector alone %.3f ms, together %.3f ms  (sum %.3f, max %.3f) -> overlap %.0f %%\n", mn, vn, t[1], t[2],
           t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2], 100.0 * (t[1] + t[2] - t[3]) / (t[1] < t[2] ? t[1] : t[2]));
}

int main() {
    float* out;
    (void)hipMalloc(&out, 512 * 512 * 4);
    run<0, 0>(out, "v_mfma_f32_32x32x2_f32", "v_pk_fma_f32");
    run<0, 1>(out, "v_mfma_f32_32x32x2_f32", "v_fma_f32");
    run<1, 0>(out, "v_mfma_f32_32x32x16_f16", "v_pk_fma_f32");
    run<1, 1>(out, "v_mfma_f32_32x32x16_f16", "v_fma_f32");
    run<1, 2>(out, "v_mfma_f32_32x32x16_f16", "v_cvt_pk_f16_f32 + v_cvt_f32_f16");
    run<1, 3>(out, "v_mfma_f32_32x32x16_f16", "v_max_f32");
    run<2, 0>(out, "v_mfma_f32_16x16x32_f16", "v_pk_fma_f32");
    run<2, 1>(out, "v_mfma_f32_16x16x32_f16", "v_fma_f32");
    run<2, 2>(out, "v_mfma_f32_16x16x32_f16", "v_cvt_pk_f16_f32 + v_cvt_f32_f16");
    return 0;
}
