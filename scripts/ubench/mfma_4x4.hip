// Layout + rate probe for v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products, K=1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    const float a = (float)(l + 1);          // A[block = l/4][i = l%4]
    const float b = (float)(100 * (l + 1));  // B[block = l/4][j = l%4]
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

__global__ __launch_bounds__(256) void rate(float* out, int iters, float a, float b) {
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{(float)threadIdx.x, 0, 0, 0};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 22);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // decode: value = a*b = (ia+1)*100*(ib+1) -> find (ia, ib)
    for (int l = 0; l < 12; ++l) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            int v = (int)(h[l * 4 + r] / 100.0f + 0.5f);
            int found = 0;
            for (int ia = 0; ia < 64 && !found; ++ia)
                for (int ib = 0; ib < 64; ++ib)
                    if ((ia + 1) * (ib + 1) == v && ia / 4 == ib / 4 && ia / 4 == l / 4) { printf("  reg%d=A[lane %d]*B[lane %d]", r, ia, ib); found = 1; break; }
            if (!found) printf("  reg%d=?(%g)", r, h[l * 4 + r]);
        }
        printf("\n");
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        const int grid = 512, iters = 20000;
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 4 * iters * 16 * 2.0 * 16 * 4 * 4;
        printf("f32 4x4x1_16b: %.3f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
    }
    return 0;
}
