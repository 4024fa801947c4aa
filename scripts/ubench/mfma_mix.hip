// Which ingredient of the dense-layer K loop costs matrix-pipe cycles?  Same instruction mix as
// gemm_glds_kernel (per k-slice: 20 MFMA 32x32x2 + 6 ds_read_b128; per K step: 4 slices, 1 barrier,
// 9 global_load_lds), ingredients switched on one at a time.  No real data dependence on memory contents.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_mix.hip -o scripts/ubench/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int READS, bool BARRIER, bool GLDS, int STAGE = 0>
__global__ __launch_bounds__(256) void k(float* out, const float* src, int steps) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 288 * 32];
    const long long c0 = __builtin_readcyclecounter();
    const long long r0 = wall_clock64();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 288 * 32; i += 256) lds[i] = 1e-3f * (i & 15);
    __syncthreads();
    f32x16 acc[5];
    for (int j = 0; j < 5; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int frow = lane & 31, fsw = frow & 7, fhalf = lane >> 5;
    const float* gsrc = src + (size_t)(blockIdx.x & 63) * 9216 + w * 64 * 4 + lane * 4;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(src + (size_t)(blockIdx.x & 63) * 9216), 0, 0x7fffffff, 0x00020000);
    const int voff = (w * 64 + lane) * 16;
    f32x4 fa = {1.f, 2.f, 3.f, 4.f}, fb[5];
    for (int j = 0; j < 5; ++j) fb[j] = fa;
    for (int kt = 0; kt < steps; ++kt) {
        const int buf = kt & 1;
        if (BARRIER) __syncthreads();
        if (GLDS) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + t * 1024),
                                                 (__attribute__((address_space(3))) void*)(lds + (buf ^ 1) * 9216 + (w + 4 * t) * 256),
                                                 16, 0, 0);
        }
        f32x4 stage[9];
        if (STAGE == 7) {            // 9 raw buffer loads -> LDS, 16 B per lane, 32-bit lane offset
#pragma unroll
            for (int t = 0; t < 9; ++t)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (buf ^ 1) * 9216 + (w + 4 * t) * 256),
                                                     16, voff, t * 4096, 0, 0);
        }
        if (STAGE == 10) {           // global_load_lds, uniform 64-bit base (SGPR pair) + 32-bit lane offset
            const char* ub = (const char*)(src + (size_t)(blockIdx.x & 63) * 9216) + (kt & 1) * 64;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + t * 4096 + (unsigned)voff),
                                                 (__attribute__((address_space(3))) void*)(lds + (buf ^ 1) * 9216 + (w + 4 * t) * 256),
                                                 16, 0, 0);
        }
        if (STAGE == 8 || STAGE == 9) {   // dword-wide global_load_lds: 36 (same bytes) or 9 (a quarter of the bytes)
#pragma unroll
            for (int t = 0; t < (STAGE == 8 ? 36 : 9); ++t)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + t * 256 - lane * 3),
                                                 (__attribute__((address_space(3))) void*)(lds + (buf ^ 1) * 9216 + (w + 4 * t) * 64),
                                                 4, 0, 0);
        }
        if (STAGE == 11) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, t * 4096, 0);
                stage[t] = __builtin_bit_cast(f32x4, v);
            }
        }
        if (STAGE == 3) {
#pragma unroll
            for (int t = 0; t < 9; ++t) stage[t] = *reinterpret_cast<const f32x4*>(gsrc + t * 1024 + kt * 0);
        }
        const float* As = lds + buf * 9216 + (w * 32 + frow) * 32;
        const float* Bs = lds + buf * 9216 + (128 + frow) * 32;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int sl = ((2 * kk + fhalf) ^ fsw) * 4;
            if (STAGE == 2) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (t % 4 == kk)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + t * 1024),
                                                         (__attribute__((address_space(3))) void*)(lds + (buf ^ 1) * 9216 + (w + 4 * t) * 256),
                                                         16, 0, 0);
            }
            if (STAGE == 4) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (t % 4 == kk) stage[t] = *reinterpret_cast<const f32x4*>(gsrc + t * 1024);
            }
            if (READS >= 1) fa = *reinterpret_cast<const f32x4*>(As + sl);
#pragma unroll
            for (int j = 0; j < 5; ++j)
                if (READS >= 1 + j + 1) fb[j] = *reinterpret_cast<const f32x4*>(Bs + j * 1024 + sl);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[j].w, acc[j], 0, 0, 0);
                if (STAGE == 5 || STAGE == 6) {
                    const int g = kk * 5 + j;                       // group of 4 MFMAs, 0..19
                    const int t = STAGE == 5 ? g : (g % 2 == 0 ? g / 2 : 99);
                    if (t < 9) {
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + t * 1024),
                                                         (__attribute__((address_space(3))) void*)(lds + (buf ^ 1) * 9216 + (w + 4 * t) * 256),
                                                         16, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (STAGE == 3 || STAGE == 4 || STAGE == 11) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
                *reinterpret_cast<f32x4*>(lds + (buf ^ 1) * 9216 + (w + 4 * t) * 256 + lane * 4) = stage[t];
        }
    }
    float s = 0;
    for (int j = 0; j < 5; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        long long* clk = reinterpret_cast<long long*>(out + 4096 * 256) + 4 * blockIdx.x;
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = r0;
        clk[2] = wall_clock64();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        clk[3] = ((long long)xcc << 32) | hw;
    }
}

template <int READS, bool BARRIER, bool GLDS, int STAGE = 0>
void run(float* out, const float* src, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = 256 * bpc, steps = 400;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k<READS, BARRIER, GLDS, STAGE>), dim3(grid), dim3(256), 0, 0, out, src, steps);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double flops = (double)grid * 4 * steps * 80 * 2.0 * 32 * 32 * 2;
        static long long clk[4 * 512];
        (void)hipMemcpy(clk, out + 4096 * 256, sizeof(long long) * 4 * grid, hipMemcpyDeviceToHost);
        long long t0 = clk[1], t1 = clk[2];
        double life_min = 1e18, life_max = 0, life_sum = 0, start_max = 0;
        for (int b = 0; b < grid; ++b) { if (clk[4 * b + 1] < t0) t0 = clk[4 * b + 1]; if (clk[4 * b + 2] > t1) t1 = clk[4 * b + 2]; }
        int cu_count[8][64] = {};
        for (int b = 0; b < grid; ++b) {
            double life = (clk[4 * b + 2] - clk[4 * b + 1]) * 0.01, st = (clk[4 * b + 1] - t0) * 0.01;
            life_sum += life; if (life < life_min) life_min = life; if (life > life_max) life_max = life; if (st > start_max) start_max = st;
            unsigned hw = (unsigned)clk[4 * b + 3], xcc = (unsigned)(clk[4 * b + 3] >> 32) & 15;
            unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cu_count[xcc & 7][(se * 2 + sh) * 16 + cu]++;
        }
        int hist[8] = {};
        for (int x = 0; x < 8; ++x) for (int c = 0; c < 64; ++c) hist[cu_count[x][c] > 7 ? 7 : cu_count[x][c]]++;
        printf("%-44s %d waves/SIMD: %.3f ms  %.1f TFLOP/s | block life us min/avg/max %.0f/%.0f/%.0f  latest start %.0f us  span %.0f us | CUs with 1/2/3/4 blocks: %d/%d/%d/%d\n",
               name, bpc, best, flops / best / 1e9, life_min, life_sum / grid, life_max, start_max, (t1 - t0) * 0.01, hist[1], hist[2], hist[3], hist[4]);
    }
}

int main() {
    float *out, *src;
    (void)hipMalloc(&out, 4096 * 256 * 8);
    (void)hipMalloc(&src, 64 * 9216 * 4 + 65536);
    (void)hipMemset(src, 0, 64 * 9216 * 4 + 65536);
    run<0, false, false>(out, src, "MFMA only");
    run<1, false, false>(out, src, "+ 1 ds_read_b128 / slice");
    run<3, false, false>(out, src, "+ 3 ds_read_b128 / slice");
    run<6, false, false>(out, src, "+ 6 ds_read_b128 / slice");
    run<0, true, false>(out, src, "MFMA + barrier / step");
    run<6, true, false>(out, src, "+ 6 reads + barrier");
    run<0, false, true>(out, src, "MFMA + 9 global_load_lds / step");
    run<0, true, true>(out, src, "MFMA + barrier + 9 global_load_lds");
    run<6, true, true>(out, src, "all (kernel mix)");
    run<6, true, false, 7>(out, src, "reads + barrier + 9 raw_buffer_load_lds x4");
    run<6, true, false, 11>(out, src, "reads + barrier + 9 buffer_load x4 -> VGPR -> ds_write");
    run<6, true, false, 10>(out, src, "reads + barrier + 9 glds x4 (saddr + voffset)");
    run<6, true, false, 8>(out, src, "reads + barrier + 36 global_load_lds x1");
    run<6, true, false, 9>(out, src, "reads + barrier + 9 global_load_lds x1");
    run<6, true, false, 5>(out, src, "reads + barrier + glds 1 per 4 MFMA");
    run<6, true, false, 6>(out, src, "reads + barrier + glds 1 per 8 MFMA");
    run<6, true, false, 2>(out, src, "reads + barrier + glds spread over slices");
    run<6, true, false, 3>(out, src, "reads + barrier + reg-staged (burst)");
    run<6, true, false, 4>(out, src, "reads + barrier + reg-staged (spread)");
    return 0;
}
