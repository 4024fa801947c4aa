// Experiment (not product code): fp32 dense layer emulated on the fp16 matrix cores with a hi/lo split,
//   x = hi + lo * 2^-11,  hi = fp16(x),  lo = fp16((x - hi) * 2^11)
//   a . w  ~=  sum hi_a hi_w  +  2^-11 * sum (hi_a lo_w + lo_a hi_w)          (lo_a lo_w ~ 2^-22 relative, dropped)
// three v_mfma_f32_32x32x16_f16 per fp32 product block, fp32 accumulation in two accumulator sets.  Operands are
// pre-split into the SAME byte geometry as fp32 rows: per row and per 32-wide K block, 32 hi halves then 32 lo
// halves (128 bytes), so the dense-layer kernel's LDS-DMA pipeline, swizzle and tile shape carry over unchanged.
// Reports accuracy against a float64 host product (small problem) and throughput (whole-round problem).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/gemm_f16x3.hip -o scripts/ubench/gemm_f16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void split_kernel(const float* __restrict__ x, _Float16* __restrict__ out, int64_t rows, int K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * K) return;
    const int64_t r = i / K;
    const int k = (int)(i - r * K);
    const float v = x[i];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
    _Float16* blk = out + (r * (K / 32) + k / 32) * 64;
    blk[k % 32] = hi;
    blk[32 + k % 32] = lo;
}

#ifndef WAVES
#define WAVES 4
#endif
constexpr int BM = 32 * WAVES, NSUB = 5, BN = 32 * NSUB, ROWS = BM + BN, BK = 32;
constexpr int CHUNKS = ROWS / 8, PER_WAVE = (CHUNKS + WAVES - 1) / WAVES, A_T = BM / 8 / WAVES;   // chunk i = w + WAVES t; t < A_T: A rows

__global__ __launch_bounds__(64 * WAVES, 8 / WAVES) void gemm_f16x3_kernel(const float* __restrict__ As, const float* __restrict__ Ws,
                                                           float* __restrict__ C, int64_t M, int N, int K, int tiles_n, long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[2 * ROWS * 32];
    const unsigned tile = blockIdx.x;
    const int64_t m0 = (int64_t)(tile / (unsigned)tiles_n) * BM;
    const int n0 = (int)(tile % (unsigned)tiles_n) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LDS row r keeps its logical 16-byte slot s at physical slot s ^ ((r >> 1) & 7): 16 consecutive rows then cover all
    // 64 banks (gfx950 serves a ds_read_b128 sixteen lanes at a time; the (r & 7) form of CDNA3 leaves a 2-way conflict)
    const int sub = lane >> 3, srccol = ((lane & 7) ^ (((w & 1) * 4 + (lane >> 4)) & 7)) * 4;
    const int a_rows = (int)((M - m0) < BM ? (M - m0) : BM) - 1;
    const int b_rows = ((N - n0) < BN ? (N - n0) : BN) - 1;
    auto uptr = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(uptr(As + m0 * K), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(uptr(Ws + (int64_t)n0 * K), 0, 0x7fffffff, 0x00020000);
    int voff[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int row = 8 * (w + WAVES * t) + sub;
        voff[t] = t < A_T ? (min(row, a_rows) * K + srccol) * 4 : (min(row - BM, b_rows) * K + srccol) * 4;
    }
#ifndef ABL
#define ABL 0
#endif
    u32x4 sink[PER_WAVE];
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) {
            if (w + WAVES * t >= CHUNKS) continue;        // (wave-uniform: the last pass is partial with 8 waves)
            float* dst = lds + buf * (ROWS * 32) + (w + WAVES * t) * 256;
#if defined(__HIP_DEVICE_COMPILE__)
#if ABL & 16      // the same loads into registers (consumed at the end of the K step): vector-memory issue and L2 traffic without the LDS writes
            if (kt > 0) { sink[t] = __builtin_amdgcn_raw_buffer_load_b128(t < A_T ? rs_a : rs_w, voff[t], kt * (BK * 4), 0); continue; }
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(t < A_T ? rs_a : rs_w, (__attribute__((address_space(3))) void*)dst, 16, voff[t],
                                                     kt * (BK * 4), 0, 0);
#endif
        }
    };
    f32x16 hh[NSUB], xx[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) hh[j][e] = xx[j][e] = 0.f;
#ifdef W2X2
    // 2 x 2 wave layout, wave tile 64 x 80 as 4 x 5 tiles of v_mfma_f32_16x16x32_f16: 18 fragment reads per K step
    // instead of 24 (each A fragment serves 5 N tiles, each B fragment 4 M tiles), same MFMA cycles (60 x 16)
    static_assert(WAVES == 4, "2 x 2 layout");
    f32x4 c1[4][5], c2[4][5];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) c1[a][b] = c2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wm2 = w >> 1, wn2 = w & 1, r16 = lane & 15, kg16 = lane >> 4;
#endif
    const int frow = lane & 31, fsw = (frow >> 1) & 7, fhalf = lane >> 5;
#ifdef DEPHASE      // timing experiment: the second workgroup of every CU starts half a tile late (its first tile is cut short)
    const int nk = (blockIdx.x >= 256 && blockIdx.x < 512) ? K / BK / 2 : K / BK;
#else
    const int nk = K / BK;
#endif
    const long long c0 = clock64(), w0 = wall_clock64();
    issue(0, 0);
    // ABL (ablation bitmask, timing only -- results are wrong): 1 = no operand loads inside the loop, 2 = fragment
    // reads do not depend on kt (the compiler hoists them), 4 = no barrier
#ifndef ABL
#define ABL 0
#endif
    for (int kt = 0; kt < nk; ++kt) {
        if (!(ABL & 4)) __syncthreads();
        if (!(ABL & 1) && kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const int kb = (ABL & 2) ? 0 : (kt & 1);
        const float* Asl = lds + kb * (ROWS * 32) + (w * 32 + frow) * 32;
        const float* Bsl = lds + kb * (ROWS * 32) + (BM + frow) * 32;
#ifdef W2X2
        {
            // a 128-byte row of this K block is [32 hi | 32 lo]; lane (row r16, k group kg16) takes k = 8 kg16 .. + 7:
            // logical slot kg16 (hi), 4 + kg16 (lo), at physical slot s ^ ((row >> 1) & 7)
            const float* A2 = lds + kb * (ROWS * 32) + (wm2 * 64 + r16) * 32;
            const float* B2 = lds + kb * (ROWS * 32) + (BM + wn2 * 80 + r16) * 32;
            const int sw = (r16 >> 1) & 7;                       // rows 16 a + r16: (row >> 1) & 7 = ((8 a) + (r16 >> 1)) & 7 = sw
            const int shi = ((kg16) ^ sw) * 4, slo = ((4 + kg16) ^ sw) * 4;
            h8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                ah[a] = *reinterpret_cast<const h8*>(A2 + a * 16 * 32 + shi);
                al[a] = *reinterpret_cast<const h8*>(A2 + a * 16 * 32 + slo);
            }
            bh[0] = *reinterpret_cast<const h8*>(B2 + shi);
            bl[0] = *reinterpret_cast<const h8*>(B2 + slo);
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                if (b + 1 < 5) {
                    bh[(b + 1) & 1] = *reinterpret_cast<const h8*>(B2 + (b + 1) * 16 * 32 + shi);
                    bl[(b + 1) & 1] = *reinterpret_cast<const h8*>(B2 + (b + 1) * 16 * 32 + slo);
                }
                __builtin_amdgcn_sched_barrier(0);
                // (no accumulator twice in a row: a 4-pass MFMA that reads the previous one's result stalls the pipe)
#pragma unroll
                for (int a = 0; a < 4; ++a) c1[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b & 1], c1[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a) c2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b & 1], c2[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a) c2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b & 1], c2[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
#endif
        // flattened (g, j) stages q = 5 g + j: stage q issues the B fragments of stage q + 2 (and the A fragments of the
        // next g at q = 3) before its three MFMAs, so every LDS read has ~200 cycles of MFMA work between issue and use
        h8 ahi[2], alo[2], bhi[3], blo[3];
#if ABL & 8
        for (int i = 0; i < 8; ++i) {     // fragments from registers: no LDS reads at all
            ahi[0][i] = ahi[1][i] = (_Float16)(float)lane; alo[0][i] = alo[1][i] = (_Float16)(float)(lane + i);
            for (int t = 0; t < 3; ++t) { bhi[t][i] = (_Float16)(float)(t + i); blo[t][i] = (_Float16)(float)(t * lane); }
        }
#define RD(dst, src) asm volatile("" : "+v"(dst))
#else
#define RD(dst, src) dst = src
#endif
        auto sl_hi = [&](int g) { return ((2 * g + fhalf) ^ fsw) * 4; };
        auto sl_lo = [&](int g) { return ((4 + 2 * g + fhalf) ^ fsw) * 4; };
        auto readB = [&](int q, int slot) {
            const int g = q / 5, j = q % 5;
            RD(bhi[slot], *reinterpret_cast<const h8*>(Bsl + j * 32 * 32 + sl_hi(g)));
            RD(blo[slot], *reinterpret_cast<const h8*>(Bsl + j * 32 * 32 + sl_lo(g)));
        };
        RD(ahi[0], *reinterpret_cast<const h8*>(Asl + sl_hi(0)));
        RD(alo[0], *reinterpret_cast<const h8*>(Asl + sl_lo(0)));
        readB(0, 0);
        readB(1, 1);
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const int g = q / 5, j = q % 5;
            if (q + 2 < 10) readB(q + 2, (q + 2) % 3);
            if (q == 3) {
                RD(ahi[1], *reinterpret_cast<const h8*>(Asl + sl_hi(1)));
                RD(alo[1], *reinterpret_cast<const h8*>(Asl + sl_lo(1)));
            }
            __builtin_amdgcn_sched_barrier(0);
            hh[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[g], bhi[q % 3], hh[j], 0, 0, 0);
            xx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[g], blo[q % 3], xx[j], 0, 0, 0);
            xx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[g], bhi[q % 3], xx[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#if ABL & 16
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) asm volatile("" ::"v"(sink[t]));
#endif
    }
    if (clk && tid == 0 && blockIdx.x < 8192) {      // effective shader clock of the K loop: s_memtime ticks per 100 MHz wall tick
        clk[2 * blockIdx.x] = (long long)(clock64() - c0);
        clk[2 * blockIdx.x + 1] = (long long)(wall_clock64() - w0);
    }
#ifdef W2X2
    // 16x16 C layout: col = lane & 15, rows 4 (lane >> 4) + i
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const int col = n0 + wn2 * 80 + b * 16 + r16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = m0 + wm2 * 64 + a * 16 + 4 * kg16 + i;
                if (row < M && col < N) C[row * N + col] = c1[a][b][i] + c2[a][b][i] * (1.0f / 2048.0f);
            }
        }
    return;
#endif
    const int colb = lane & 31, rowb = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        const int col = n0 + j * 32 + colb;
        if (col >= N) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = m0 + w * 32 + rowb + (e & 3) + 8 * (e >> 2);
            if (row < M) C[row * N + col] = hh[j][e] + xx[j][e] * (1.0f / 2048.0f);
        }
    }
}

static void run(int64_t M, int N, int K, bool check) {
    std::vector<float> ha((size_t)M * K), hw((size_t)N * K);
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 33) & 0xffffff) / 8388608.0f - 1.0f; };
    for (auto& v : ha) v = rnd() * 3.0f;            // activations of a few units
    for (auto& v : hw) v = rnd() * 0.05f;           // weights ~ 1/sqrt(K)
    float *a, *wt, *c;
    _Float16 *as, *ws;
    (void)hipMalloc(&a, ha.size() * 4);
    (void)hipMalloc(&wt, hw.size() * 4);
    (void)hipMalloc(&as, ha.size() * 4);
    (void)hipMalloc(&ws, hw.size() * 4);
    (void)hipMalloc(&c, (size_t)M * N * 4);
    long long* clk;
    (void)hipMalloc(&clk, 16384 * 8);
    (void)hipMemset(clk, 0, 16384 * 8);
    (void)hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(wt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((ha.size() + 255) / 256)), dim3(256), 0, 0, a, as, M, K);
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((hw.size() + 255) / 256)), dim3(256), 0, 0, wt, ws, (int64_t)N, K);
    const int tiles_n = (N + BN - 1) / BN;
    const unsigned grid = (unsigned)(((M + BM - 1) / BM) * tiles_n);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    const int reps = check ? 1 : 40;
    const size_t filler = getenv("FILLER_MB") ? (size_t)atoi(getenv("FILLER_MB")) << 20 : 0;    // see gemm_f16x3_w64.hip
    void* fbuf = nullptr;
    if (filler) (void)hipMalloc(&fbuf, filler);
    for (int rep = 0; rep < reps; ++rep) {
        if (filler) (void)hipMemsetAsync(fbuf, rep, filler, 0);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(gemm_f16x3_kernel, dim3(grid), dim3(64 * WAVES), 0, 0, (const float*)as, (const float*)ws, c, M, N, K, tiles_n, clk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        if (rep >= reps / 2) sum += ms;
    }
    {
        std::vector<long long> hclk(16384);
        (void)hipMemcpy(hclk.data(), clk, 16384 * 8, hipMemcpyDeviceToHost);
        double cs = 0, wsum = 0;
        const int nt = grid < 8192 ? (int)grid : 8192;
        for (int i = 0; i < nt; ++i) { cs += (double)hclk[2 * i]; wsum += (double)hclk[2 * i + 1]; }
        if (!check) printf("  K loop, mean over %d tiles: %.0f shader cycles, %.2f us -> %.2f GHz effective; %.0f cycles per K step\n",
                           nt, cs / nt, wsum / nt / 100.0, cs / wsum / 10.0, cs / nt / (K / 32));
    }
    const float mean = sum / (reps - reps / 2);
    printf("M=%lld N=%d K=%d: %.3f ms  %.1f fp32-equivalent TFLOP/s | mean of last half %.3f ms %.1f (%s)\n", (long long)M, N, K, best,
           2.0 * M * N * K / best / 1e9, mean, 2.0 * M * N * K / mean / 1e9, hipGetErrorString(hipGetLastError()));
    if (fbuf) (void)hipFree(fbuf);
    if (check) {
        std::vector<float> hc((size_t)M * N);
        (void)hipMemcpy(hc.data(), c, hc.size() * 4, hipMemcpyDeviceToHost);
        double e_split = 0, e_f32 = 0, scale = 0;
        for (int64_t i = 0; i < M; i += 7)
            for (int j = 0; j < N; j += 3) {
                double ref = 0;
                float f32 = 0.f;
                for (int k = 0; k < K; ++k) {
                    ref += (double)ha[i * K + k] * (double)hw[(size_t)j * K + k];
                    f32 = fmaf(ha[i * K + k], hw[(size_t)j * K + k], f32);
                }
                e_split = fmax(e_split, fabs((double)hc[i * N + j] - ref));
                e_f32 = fmax(e_f32, fabs((double)f32 - ref));
                scale = fmax(scale, fabs(ref));
            }
        printf("  max |err| vs float64: f16x3 split %.3e, plain fp32 fmaf chain %.3e   (max |value| %.2f)\n", e_split, e_f32, scale);
    }
    (void)hipFree(a); (void)hipFree(wt); (void)hipFree(as); (void)hipFree(ws); (void)hipFree(c);
}

int main() {
    run(300, 170, 1440, true);
    run(1000, 800, 800, true);
    run(65536, 1440, 1440, false);
    run(65536, 800, 800, false);
    return 0;
}
