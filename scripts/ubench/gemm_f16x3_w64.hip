// Experiment (not product code): the fp16x3 dense-layer K loop with ONE wave per SIMD and the whole 512-register file.
//   workgroup tile 256 x 160 x 32, 4 waves x (64 rows x 160 columns): two 32-row blocks share every W fragment, so a
//   wave issues 28 fragment reads + 13 LDS-DMA loads per 60 MFMAs (0.68 operand instructions per MFMA) against
//   24 + 9 per 30 (1.1) of the 128 x 160 kernel with 32 x 160 per wave (gemm_f16x3.hip).
//   Three LDS operand buffers (3 x 52 KB): tile kt + 2 is in flight while tile kt is multiplied; ONE barrier per K step,
//   placed two stages before the step's end so that the first fragments of tile kt + 1 are read under the last MFMAs of
//   tile kt (with one wave per SIMD nothing else hides an LDS round trip); counted vmcnt, raw s_barrier.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/gemm_f16x3_w64.hip -o scripts/ubench/gemm_f16x3_w64
//   -DPATTERN=n   where the 13 LDS-DMA loads of a K step sit among its MFMAs (dma_piece)
//   -DTOPBAR      barrier at the top of the K step (no cross-step fragment prefetch)
//   -DPERSIST     256 workgroups walk the tiles; a tile's first operands are requested before the previous tile's epilogue
//   -DABL=mask    1 = no operand loads in the loop, 8 = fragments from registers (timing only)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef ABL
#define ABL 0
#endif

// lo_mask: timing experiment -- clear that many low mantissa bits of every lo half (16: lo = 0): how much of the K loop's
// time is the energy of multiplying full-entropy operands (the chip clocks to its power budget)?
__global__ void split_kernel(const float* __restrict__ x, _Float16* __restrict__ out, int64_t rows, int K, int lo_mask = 0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * K) return;
    const int64_t r = i / K;
    const int k = (int)(i - r * K);
    const float v = x[i];
    const _Float16 hi = (_Float16)v;
    _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
    if (lo_mask) {
        unsigned short b = __builtin_bit_cast(unsigned short, lo);
        b = lo_mask >= 16 ? 0 : (unsigned short)(b & ~((1u << lo_mask) - 1u));
        lo = __builtin_bit_cast(_Float16, b);
    }
    _Float16* blk = out + (r * (K / 32) + k / 32) * 64;
    blk[k % 32] = hi;
    blk[32 + k % 32] = lo;
}

constexpr int BM = 256, NSUB = 5, BN = 32 * NSUB, ROWS = BM + BN, BK = 32, NBUF = 3;
constexpr int CHUNKS = ROWS / 8, PER_WAVE = CHUNKS / 4, A_T = BM / 8 / 4;   // chunk i = w + 4 t; t < A_T: A rows
constexpr int BUF_FLOATS = ROWS * 32;
static_assert(CHUNKS % 4 == 0, "chunks divide over the waves");

// which LDS-DMA piece (0..12, -1 = none) of tile kt + 2 is issued after MFMA pair `slot` (0..2) of stage q; all before stage 8
#ifndef PATTERN
#define PATTERN 0
#endif
constexpr int dma_piece(int q, int slot) {
#if PATTERN == 0       // two per stage, stages 0..6
    return (slot < 2 && q < 6) ? 2 * q + slot : (q == 6 && slot == 0) ? 12 : -1;
#elif PATTERN == 1     // three per stage, stages 0..4
    return q < 4 ? 3 * q + slot : (q == 4 && slot == 0) ? 12 : -1;
#else                  // evenly over the 24 positions of stages 0..7
    const int p = 3 * q + slot;
    for (int i = 0; i < 13; ++i)
        if (i * 24 / 13 == p) return i;
    return -1;
#endif
}

// The two accumulator sets of a 64 x 160 wave tile are 320 registers: more than the 256 accumulation registers, and hipcc
// puts the accumulator of EVERY builtin MFMA of a kernel into one class (3,594 spilled registers when tried).  The MFMAs are
// therefore inline asm: 16 accumulators live in AGPRs ("+a"), 4 in VGPRs ("+v").  An accumulate chain needs no wait states;
// the compiler waits for the ds_reads that feed an asm statement's inputs like for any other consumer.
__device__ __forceinline__ void mfma_acc(f32x16& c, const h8& a, const h8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vgp(f32x16& c, const h8& a, const h8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#ifdef ASM_ACC
#define MFMA(c, a, b, in_vgpr) do { if (in_vgpr) mfma_vgp(c, a, b); else mfma_acc(c, a, b); } while (0)
#else    // 16 accumulators through the builtin (the compiler keeps them in the 256 AGPRs), 4 through asm in VGPRs
#define MFMA(c, a, b, in_vgpr) do { if (in_vgpr) mfma_vgp(c, a, b); else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); } while (0)
#endif

#ifndef NVG
#define NVG 4      // accumulators (of 20) kept in VGPRs
#endif
struct Frags {
    h8 ahi[2][2], alo[2][2];     // [gk][row block]
    h8 bhi[5], blo[5];           // ring slot = column block
};

__global__ __launch_bounds__(256, 1) void gemm_w64_kernel(const float* __restrict__ As, const float* __restrict__ Ws, float* __restrict__ C,
                                                          int64_t M, int N, int K, int tiles_n, long long* clk, unsigned ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[NBUF * BUF_FLOATS];      // 159,744 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane >> 3, srccol = ((lane & 7) ^ (((w & 1) * 4 + (lane >> 4)) & 7)) * 4;
    // buffer descriptors as four SGPR dwords: base, base high (stride 0), num_records, flags
    auto make_rsrc = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((uint32_t)v);
        r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        r[2] = 0x7fffffffu;
        r[3] = 0x00020000u;
        return r;
    };
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds) + (unsigned)w * 1024u;
    // per-tile state (PERSIST: a workgroup walks the tiles blockIdx.x, + gridDim.x, ...)
    int64_t m0;
    int n0;
    u32x4 rs_a, rs_w;
    int voff[PER_WAVE];
    auto setup = [&](unsigned tile) {
        m0 = (int64_t)(tile / (unsigned)tiles_n) * BM;
        n0 = (int)(tile % (unsigned)tiles_n) * BN;
        const int a_rows = (int)((M - m0) < BM ? (M - m0) : BM) - 1;
        const int b_rows = ((N - n0) < BN ? (N - n0) : BN) - 1;
        rs_a = make_rsrc(As + m0 * K);
        rs_w = make_rsrc(Ws + (int64_t)n0 * K);
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) {
            const int row = 8 * (w + 4 * t) + sub;
            voff[t] = t < A_T ? (min(row, a_rows) * K + srccol) * 4 : (min(row - BM, b_rows) * K + srccol) * 4;
        }
    };
    unsigned tile = blockIdx.x;
    setup(tile);
    // `nrec` = num_records of the descriptor for this tile: 0 for a tile index past the end -- every lane is then out of
    // range and the load is dropped (the scalar offset is not part of the range check, so it cannot carry the condition),
    // but it still counts in vmcnt: the loop needs no tail variants.
    auto dma = [&](int t, unsigned nrec, int bufoff, int kofs) {
        const unsigned dst = lds_base + (unsigned)(bufoff * 4 + t * 4096);
        u32x4 rs = t < A_T ? rs_a : rs_w;
        rs[2] = nrec;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff[t]), "s"(rs), "s"(kofs) : "memory");
    };
    f32x16 hh[2][NSUB], xx[2][NSUB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < NSUB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) hh[b][j][e] = xx[b][j][e] = 0.f;
    };
    zero_acc();

    // fragment addresses: row (lane & 31) of a 32-row block; logical 16-byte slot x | fhalf (x = 2 gk for the hi halves,
    // 4 + 2 gk for the lo halves) sits at physical slot (x | fhalf) ^ fsw = x ^ (fhalf ^ fsw)
    const int frow = lane & 31, fsw = (frow >> 1) & 7, fhalf = lane >> 5, fc = fhalf ^ fsw;
    const int a_lane = (w * 64 + frow) * 32, b_lane = (BM + frow) * 32;
    auto slot = [&](int x) { return (x ^ fc) * 4; };
    const int nk = K / BK;
    Frags f;
#if ABL & 8
    for (int i = 0; i < 8; ++i) {
        for (int g = 0; g < 2; ++g)
            for (int b = 0; b < 2; ++b) { f.ahi[g][b][i] = (_Float16)(float)(lane + b); f.alo[g][b][i] = (_Float16)(float)(lane + i + g); }
        for (int t = 0; t < 5; ++t) { f.bhi[t][i] = (_Float16)(float)(t + i); f.blo[t][i] = (_Float16)(float)(t * lane); }
    }
#define RD(dst, src) asm volatile("" : "+v"(dst))
#else
#define RD(dst, src) dst = src
#endif
    auto read_a = [&](int gk, int bufoff) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            RD(f.ahi[gk][b], *reinterpret_cast<const h8*>(lds + bufoff + a_lane + b * 32 * 32 + slot(2 * gk)));
            RD(f.alo[gk][b], *reinterpret_cast<const h8*>(lds + bufoff + a_lane + b * 32 * 32 + slot(4 + 2 * gk)));
        }
    };
    const long long c0 = clock64(), w0 = wall_clock64();
    // W fragments of column block j always live in ring slot j (5 slots: 10 stages per K step, so no ring phase to unroll)
    auto rb = [&](int q, int bufoff) {
        const int gk = q / NSUB, j = q % NSUB;
        RD(f.bhi[j], *reinterpret_cast<const h8*>(lds + bufoff + b_lane + j * 32 * 32 + slot(2 * gk)));
        RD(f.blo[j], *reinterpret_cast<const h8*>(lds + bufoff + b_lane + j * 32 * 32 + slot(4 + 2 * gk)));
    };
    // prologue: tiles 0 and 1 in flight, tile 0 landed, first fragments read
    constexpr unsigned NREC = 0x7fffffffu;
    auto issue_first_two = [&]() {
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) dma(t, NREC, 0, 0);
        const unsigned nrec = nk > 1 ? NREC : 0u;
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) dma(t, nrec, BUF_FLOATS, BK * 4);
    };
    issue_first_two();
#ifdef PERSIST
  for (;;) {
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();
#ifndef TOPBAR
    read_a(0, 0);
    rb(0, 0);
    rb(1, 0);
#endif
    // one K step on the tile in buffer BUFC; the loop is unrolled by three so that every LDS address is a lane constant +
    // an immediate (a rotating run-time offset also made hipcc lose the 16-byte alignment of the fragment reads)
    auto step = [&](auto bufc, int kt) {
        constexpr int cur = decltype(bufc)::value * BUF_FLOATS, nxt = ((decltype(bufc)::value + 1) % 3) * BUF_FLOATS,
                      fil = ((decltype(bufc)::value + 2) % 3) * BUF_FLOATS;
        const int kofs = (kt + 2) * (BK * 4);
        const unsigned nrec = kt + 2 < nk ? NREC : 0u;
#pragma unroll
        for (int q = 0; q < 2 * NSUB; ++q) {
            const int gk = q / NSUB, j = q % NSUB, s = j;
#ifdef TOPBAR
            if (q == 0) {
                if (kt > 0) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
                read_a(0, cur);
                rb(0, cur);
                rb(1, cur);
            }
            if (q + 2 < 2 * NSUB) rb(q + 2, cur);
#else
            if (q == 8) {
                // tile kt + 1 has landed (this wave's share: all but the 13 youngest loads) and is visible to all waves
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (q + 2 < 2 * NSUB) rb(q + 2, cur);
            else rb(q + 2 - 2 * NSUB, nxt);               // stages 0 / 1 of the next step
            if (q == 8) read_a(0, nxt);
#endif
            if (q == 2) read_a(1, cur);
            __builtin_amdgcn_sched_barrier(0);
            const bool vg = j >= 3;      // row block 1 of column blocks 3, 4: accumulators in VGPRs
            const bool vg0 = NVG >= 6 ? j >= 4 : false, vx0 = NVG >= 5 ? j >= 4 : false;
            MFMA(hh[0][j], f.ahi[gk][0], f.bhi[s], vg0);
            MFMA(hh[1][j], f.ahi[gk][1], f.bhi[s], vg);
            if (!(ABL & 1) && dma_piece(q, 0) >= 0) { __builtin_amdgcn_sched_barrier(0); dma(dma_piece(q, 0), nrec, fil, kofs); __builtin_amdgcn_sched_barrier(0); }
            MFMA(xx[0][j], f.ahi[gk][0], f.blo[s], vx0);
            MFMA(xx[1][j], f.ahi[gk][1], f.blo[s], vg);
            if (!(ABL & 1) && dma_piece(q, 1) >= 0) { __builtin_amdgcn_sched_barrier(0); dma(dma_piece(q, 1), nrec, fil, kofs); __builtin_amdgcn_sched_barrier(0); }
            MFMA(xx[0][j], f.alo[gk][0], f.bhi[s], vx0);
            MFMA(xx[1][j], f.alo[gk][1], f.bhi[s], vg);
            if (!(ABL & 1) && dma_piece(q, 2) >= 0) { __builtin_amdgcn_sched_barrier(0); dma(dma_piece(q, 2), nrec, fil, kofs); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int kt = 0;
    for (; kt + 3 <= nk; kt += 3) {
        step(std::integral_constant<int, 0>(), kt);
        step(std::integral_constant<int, 1>(), kt + 1);
        step(std::integral_constant<int, 2>(), kt + 2);
    }
    if (kt < nk) step(std::integral_constant<int, 0>(), kt);
    if (kt + 1 < nk) step(std::integral_constant<int, 1>(), kt + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // last MFMA results -> compiler-generated readers
    if (clk && tid == 0 && blockIdx.x < 8192) {
        clk[2 * blockIdx.x] = (long long)(clock64() - c0);
        clk[2 * blockIdx.x + 1] = (long long)(wall_clock64() - w0);
    }
    const int64_t em0 = m0;
    const int en0 = n0;
#ifdef PERSIST
    // the next tile's first two K tiles are requested BEFORE this tile's epilogue: with one workgroup per CU nothing else hides
    // the first operand round trip of a tile
    const unsigned next = tile + gridDim.x;
    const bool more = next < ntiles;
    __builtin_amdgcn_s_barrier();              // every wave is done with the operand buffers
    if (more) {
        setup(next);
        issue_first_two();
    }
#endif
    const int colb = lane & 31, rowb = 4 * (lane >> 5);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            const int col = en0 + j * 32 + colb;
            if (col >= N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = em0 + w * 64 + b * 32 + rowb + (e & 3) + 8 * (e >> 2);
                if (row < M) C[row * N + col] = hh[b][j][e] + xx[b][j][e] * (1.0f / 2048.0f);
            }
        }
#ifdef PERSIST
    if (!more) break;
    tile = next;
    zero_acc();
  }
#endif
}

static int g_lo_mask = 0;
static void run(int64_t M, int N, int K, bool check) {
    std::vector<float> ha((size_t)M * K), hw((size_t)N * K);
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 33) & 0xffffff) / 8388608.0f - 1.0f; };
    for (auto& v : ha) v = rnd() * 3.0f;
    for (auto& v : hw) v = rnd() * 0.05f;
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
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((ha.size() + 255) / 256)), dim3(256), 0, 0, a, as, M, K, g_lo_mask);
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((hw.size() + 255) / 256)), dim3(256), 0, 0, wt, ws, (int64_t)N, K, g_lo_mask);
    const int tiles_n = (N + BN - 1) / BN;
    const unsigned grid = (unsigned)(((M + BM - 1) / BM) * tiles_n);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    const int reps = check ? 1 : 40;
    // FILLER_MB=n: a memset of n MB between the launches (a low-power phase, as the other kernels of the real step are):
    // the chip clocks to its power budget over milliseconds, so back-to-back launches see the lowest clock
    const size_t filler = getenv("FILLER_MB") ? (size_t)atoi(getenv("FILLER_MB")) << 20 : 0;
    void* fbuf = nullptr;
    if (filler) (void)hipMalloc(&fbuf, filler);
    for (int rep = 0; rep < reps; ++rep) {
        if (filler) (void)hipMemsetAsync(fbuf, rep, filler, 0);
        (void)hipEventRecord(e0);
#ifdef PERSIST
        const unsigned launch_grid = grid < 256 ? grid : 256;
#else
        const unsigned launch_grid = grid;
#endif
        hipLaunchKernelGGL(gemm_w64_kernel, dim3(launch_grid), dim3(256), 0, 0, (const float*)as, (const float*)ws, c, M, N, K, tiles_n, clk, grid);
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
    printf("M=%lld N=%d K=%d: best %.3f ms  %.1f TFLOP/s | mean of last half %.3f ms %.1f TFLOP/s fp32-equivalent (%s)\n", (long long)M, N, K, best,
           2.0 * M * N * K / best / 1e9, mean, 2.0 * M * N * K / mean / 1e9, hipGetErrorString(hipGetLastError()));
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
    if (fbuf) (void)hipFree(fbuf);
}

int main(int argc, char** argv) {
    if (argc > 1) {          // ./gemm_f16x3_w64 <lo_mask bits>: timing runs only
        g_lo_mask = atoi(argv[1]);
        printf("lo halves with the low %d mantissa bits cleared\n", g_lo_mask);
        run(1000, 800, 800, true);
        run(65536, 1440, 1440, false);
        run(65536, 800, 800, false);
        return 0;
    }
    run(300, 170, 1440, true);
    run(1000, 800, 800, true);
    run(777, 320, 32, true);       // one K step
    run(777, 320, 64, true);       // two
    run(777, 320, 96, true);
    run(777, 320, 128, true);
    run(777, 320, 160, true);
    run(777, 320, 192, true);
    run(65536, 1440, 1440, false);
    run(65536, 800, 800, false);
    run(65536, 1120, 1120, false);
    return 0;
}
