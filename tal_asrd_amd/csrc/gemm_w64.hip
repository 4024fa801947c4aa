// The fp16x3 dense layer with ONE wave per SIMD and the whole 512-register file: workgroup tile 256 x 160 x 32,
// 4 waves x (64 rows x 160 columns).  The TDS block's 1x1 Conv1d pair (tal/asr/models.py:312-318,330) on long inputs.
//
// Why: in the 128 x 160 kernel (gemm_f32.hip) a wave owns 32 x 160 and re-reads the whole W tile for 30 MFMAs -- 24 fragment
// reads + 9 LDS-DMA loads per K step, 1.1 operand instructions per MFMA.  Here two 32-row blocks share every W fragment:
// 28 reads + 13 loads per 60 MFMAs (0.68), 28 % fewer operand bytes through L2 -> LDS, and W is re-read by half as many
// tiles.  Measured stand-alone (scripts/ubench/gemm_f16x3_w64.hip, profiles/r3_ubench_gemm_w64*.txt): the K loop costs
// 2,240-2,290 shader cycles per K step against 1,920 of pure MFMA issue -- the issue side is done -- and sustained
// throughput is 370-383 TFLOP/s fp32-equivalent against 350-363 of the 128 x 160 kernel on the same box; what both are
// bound by is the clock the chip sustains under this load (1.4-1.6 GHz of 2.4), DESIGN.md section 5.
//
// Structure:
//   * three LDS operand buffers (3 x 52 KB): tile kt + 2 is in flight while tile kt is multiplied; ONE barrier per K step,
//     placed two stages before the step's end, so that the first fragments of tile kt + 1 are read under the last MFMAs
//     of tile kt (one wave per SIMD: nothing else hides an LDS round trip); counted vmcnt, raw s_barrier;
//   * LDS-DMA in inline asm: hipcc would make every ds_read wait for ALL LDS-DMA loads in flight (it cannot tell the
//     buffers apart), which serialises the pipeline; M0 is written in the same statement; completion is counted by hand;
//   * a tile index past the end of K becomes a descriptor with num_records = 0 (the load is dropped by the range check
//     but still counts in vmcnt), so the loop has no tail variants -- hipcc answers multi-exit loops with thousands of
//     spilled registers here;
//   * 20 accumulators of 16 registers: 15 through the MFMA builtin (hipcc keeps every builtin MFMA's accumulator in the 256
//     AGPRs) and 5 through inline asm in VGPRs -- an accumulate chain needs no wait states, and the compiler waits for the
//     ds_reads that feed an asm statement like for any other consumer;
//   * same operand geometry, swizzle, arithmetic order per accumulator and epilogues as the 128 x 160 kernel: whole tiles
//     produce bit-identical results.
// The tiles of the last partial scheduling round (1 workgroup per CU: rounds of 256 tiles) are cut along K into slices
// dispatched behind the whole tiles; gemm_splitk_fixup_kernel adds them in a fixed order (as in gemm_f32.hip).
#include <type_traits>

#include "gemm_common.h"

namespace tal {

namespace {

typedef unsigned u32x4w __attribute__((ext_vector_type(4)));

constexpr int W_BM = 256, W_NSUB = 5, W_BN = 32 * W_NSUB, W_ROWS = W_BM + W_BN, W_NBUF = 3;
constexpr int W_CHUNKS = W_ROWS / 8, W_PER_WAVE = W_CHUNKS / 4, W_AT = W_BM / 8 / 4;   // chunk i = w + 4 t; t < W_AT: A rows
constexpr int W_BUF_FLOATS = W_ROWS * 32;
static_assert(W_CHUNKS % 4 == 0, "chunks divide over the waves");

// which LDS-DMA piece (0..12, -1 = none) of tile kt + 2 is issued after MFMA pair `slot` (0..2) of stage q: two per stage,
// stages 0..6 -- all of them before the barrier of stage 8 (evenly spread or three per stage measured the same or worse)
constexpr int dma_piece(int q, int slot) { return (slot < 2 && q < 6) ? 2 * q + slot : (q == 6 && slot == 0) ? 12 : -1; }

__device__ __forceinline__ void mfma_vgpr(f32x16& c, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#define W64_MFMA(c, a, b, in_vgpr) do { if (in_vgpr) mfma_vgpr(c, a, b); else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); } while (0)

struct Frags {
    f16x8 ahi[2][2], alo[2][2];     // [gk][row block]
    f16x8 bhi[5], blo[5];           // ring slot = column block
};

}  // namespace

template <int MODE, bool SPLITK, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_w64_kernel(const GemmArgs g) {
    constexpr int NSUB = W_NSUB, BM = W_BM, BN = W_BN, PER_WAVE = W_PER_WAVE, A_T = W_AT, BUF_FLOATS = W_BUF_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[W_NBUF * W_BUF_FLOATS];      // 159,744 B: one workgroup per CU

    if (g.stagger_ticks > 0 && blockIdx.x < (unsigned)g.stagger_blocks && (blockIdx.x & 1u)) {
        // (every tile of a round runs the same K loop, so all 256 workgroups reach their epilogue -- a 42 MB write burst with the
        //  matrix pipes idle -- together; a late start of every other first-round workgroup puts the two halves of the chip out
        //  of phase for the whole launch: later workgroups start as CUs free up)
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)g.stagger_ticks) __builtin_amdgcn_s_sleep(64);
    }
    // XCD-remapped tile index; SPLITK launches carry the whole-round tiles and, behind them, the K slices of the rest
    const bool is_slice = SPLITK && blockIdx.x >= (unsigned)g.tile_base;
    const unsigned sbid = blockIdx.x - (unsigned)g.tile_base;
    const int slice = is_slice ? (int)(sbid / (unsigned)g.tail_tiles) : 0;
    const unsigned logical = is_slice ? (unsigned)g.tile_base + sbid % (unsigned)g.tail_tiles
                                      : (SPLITK ? logical_tile_of((unsigned)g.tile_base, blockIdx.x) : logical_tile());
    const unsigned tile_m = g.tiles_n == 1 ? logical : __umulhi(logical, g.tiles_n_magic);
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = (int)(logical - tile_m * (unsigned)g.tiles_n) * BN;
    const int64_t M = g.M;
    const int N = g.N, K = g.K;
    const float* A = g.A;
    const float* W = g.W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = wave_id();

    // operand loads: as in gemm_glds_kernel -- chunk = 8 rows x 128 B, slot s of row r holds the row's logical 16-byte column
    // s ^ ((r >> 1) & 7) (swizzle on the SOURCE address, linear LDS destination)
    const int sub = lane >> 3, srccol = ((lane & 7) ^ (((w & 1) * 4 + (lane >> 4)) & 7)) * 4;
    const int a_rows = (int)((M - m0) < BM ? (M - m0) : BM) - 1;
    const int b_rows = ((N - n0) < BN ? (N - n0) : BN) - 1;
    auto make_rsrc = [](const float* p) {      // buffer descriptor as four SGPR dwords: base, base high (stride 0), num_records, flags
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        u32x4w r;
        r[0] = __builtin_amdgcn_readfirstlane((uint32_t)v);
        r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        r[2] = 0x7fffffffu;
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4w rs_a = make_rsrc(A + m0 * g.lda), rs_w = make_rsrc(W + (int64_t)n0 * g.ldw);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds) + (unsigned)w * 1024u;
    int voff[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int row = 8 * (w + 4 * t) + sub;
        voff[t] = t < A_T ? (int)((min(row, a_rows) * g.lda + srccol) * 4) : (int)((min(row - BM, b_rows) * g.ldw + srccol) * 4);
    }
    auto dma = [&](int t, unsigned nrec, int bufoff, int kofs) {
        const unsigned dst = lds_base + (unsigned)(bufoff * 4 + t * 4096);
        u32x4w rs = t < A_T ? rs_a : rs_w;
        rs[2] = nrec;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff[t]), "s"(rs), "s"(kofs) : "memory");
    };

    // accumulators: hh = sum hi*hi (EPI: started from the bias), xx = the cross terms hi*lo + lo*hi (scaled by 2^11)
    f32x16 hh[2][NSUB], xx[2][NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        const float b0 = (EPI && g.bias && !is_slice) ? g.bias[n0 + j * 32 + (lane & 31)] : 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                hh[b][j][e] = b0;
                xx[b][j][e] = 0.f;
            }
    }

    // fragment addresses: row (lane & 31) of a 32-row block; logical 16-byte slot x | fhalf (x = 2 gk: hi halves, 4 + 2 gk: lo
    // halves) sits at physical slot (x | fhalf) ^ fsw = x ^ (fhalf ^ fsw)
    const int frow = lane & 31, fsw = (frow >> 1) & 7, fhalf = lane >> 5, fc = fhalf ^ fsw;
    const int a_lane = (w * 64 + frow) * 32, b_lane = (BM + frow) * 32;
    auto slot = [&](int x) { return (x ^ fc) * 4; };
    Frags f;
    auto read_a = [&](int gk, int bufoff) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f.ahi[gk][b] = *reinterpret_cast<const f16x8*>(lds + bufoff + a_lane + b * 32 * 32 + slot(2 * gk));
            f.alo[gk][b] = *reinterpret_cast<const f16x8*>(lds + bufoff + a_lane + b * 32 * 32 + slot(4 + 2 * gk));
        }
    };
    auto rb = [&](int q, int bufoff) {          // W fragments of stage q = 5 gk + j into ring slot j
        const int gk = q / NSUB, j = q % NSUB;
        f.bhi[j] = *reinterpret_cast<const f16x8*>(lds + bufoff + b_lane + j * 32 * 32 + slot(2 * gk));
        f.blo[j] = *reinterpret_cast<const f16x8*>(lds + bufoff + b_lane + j * 32 * 32 + slot(4 + 2 * gk));
    };

    const int nk_all = K / BK;
    const int kt0 = is_slice ? (int)((int64_t)slice * nk_all / g.split) : 0;
    const int kt1 = is_slice ? (int)((int64_t)(slice + 1) * nk_all / g.split) : nk_all;
    constexpr unsigned NREC = 0x7fffffffu;
    // prologue: tiles kt0 and kt0 + 1 in flight, tile kt0 landed, first fragments read
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) dma(t, NREC, 0, kt0 * (BK * 4));
    {
        const unsigned nrec = kt0 + 1 < kt1 ? NREC : 0u;
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) dma(t, nrec, BUF_FLOATS, (kt0 + 1) * (BK * 4));
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();
    read_a(0, 0);
    rb(0, 0);
    rb(1, 0);
    // one K step on the tile in buffer BUFC; the loop is unrolled by three so that every LDS address is a lane constant + an
    // immediate (a rotating run-time offset also makes hipcc lose the 16-byte alignment of the fragment reads)
    auto step = [&](auto bufc, int kt) {
        constexpr int cur = decltype(bufc)::value * BUF_FLOATS, nxt = ((decltype(bufc)::value + 1) % 3) * BUF_FLOATS,
                      fil = ((decltype(bufc)::value + 2) % 3) * BUF_FLOATS;
        const int kofs = (kt + 2) * (BK * 4);
        const unsigned nrec = kt + 2 < kt1 ? NREC : 0u;
#pragma unroll
        for (int q = 0; q < 2 * NSUB; ++q) {
            const int gk = q / NSUB, j = q % NSUB;
            if (q == 8) {
                // tile kt + 1 has landed (this wave's share: all but the 13 youngest loads) and is visible to all waves
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (q + 2 < 2 * NSUB) rb(q + 2, cur);
            else rb(q + 2 - 2 * NSUB, nxt);               // stages 0 / 1 of the next step
            if (q == 8) read_a(0, nxt);
            if (q == 2) read_a(1, cur);
            __builtin_amdgcn_sched_barrier(0);
            // 5 of the 20 accumulators live in VGPRs (row block 1 of column blocks 3, 4; the cross terms of row block 0, column block 4)
            const bool vg = j >= 3, vx0 = j >= 4;
            W64_MFMA(hh[0][j], f.ahi[gk][0], f.bhi[j], false);
            W64_MFMA(hh[1][j], f.ahi[gk][1], f.bhi[j], vg);
            if (dma_piece(q, 0) >= 0) { __builtin_amdgcn_sched_barrier(0); dma(dma_piece(q, 0), nrec, fil, kofs); __builtin_amdgcn_sched_barrier(0); }
            W64_MFMA(xx[0][j], f.ahi[gk][0], f.blo[j], vx0);
            W64_MFMA(xx[1][j], f.ahi[gk][1], f.blo[j], vg);
            if (dma_piece(q, 1) >= 0) { __builtin_amdgcn_sched_barrier(0); dma(dma_piece(q, 1), nrec, fil, kofs); __builtin_amdgcn_sched_barrier(0); }
            W64_MFMA(xx[0][j], f.alo[gk][0], f.bhi[j], vx0);
            W64_MFMA(xx[1][j], f.alo[gk][1], f.bhi[j], vg);
            if (dma_piece(q, 2) >= 0) { __builtin_amdgcn_sched_barrier(0); dma(dma_piece(q, 2), nrec, fil, kofs); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int kt = kt0;
    for (; kt + 3 <= kt1; kt += 3) {
        step(std::integral_constant<int, 0>(), kt);
        step(std::integral_constant<int, 1>(), kt + 1);
        step(std::integral_constant<int, 2>(), kt + 2);
    }
    if (kt < kt1) step(std::integral_constant<int, 0>(), kt);
    if (kt + 1 < kt1) step(std::integral_constant<int, 1>(), kt + 1);
    // every LDS-DMA has landed (dropped ones included) and the last MFMA results are readable; then all waves are done with
    // the operand buffers before they become the store stage
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 s11 = {1.0f / 2048.0f, 1.0f / 2048.0f};
    // (a lambda called twice with a compile-time row block: the generic epilogue is too large for hipcc to unroll a loop around
    //  it, and a run-time index would send the accumulators to scratch memory for the whole kernel)
    auto finish = [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        f32x16 acc[NSUB];
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 x2 = {xx[b][j][e], xx[b][j][e + 1]}, a2 = {hh[b][j][e], hh[b][j][e + 1]};
                const f32x2 o2 = __builtin_elementwise_fma(x2, s11, a2);
                acc[j][e] = o2[0];
                acc[j][e + 1] = o2[1];
            }
        // the shared epilogues address rows as m0 + 32 w + ...: this wave's row block b starts at m0 + 64 w + 32 b
        const int64_t m0e = m0 + 32 * w + 32 * b;
        if (b == 1) __builtin_amdgcn_wave_barrier();        // (the wave's stage slice is reused: LDS operations of a wave are in order)
        if (is_slice) {
            GemmArgs gp = g;
            gp.M = m0 + BM;
            gp.N = n0 + BN;
            gp.ldy = BN;
            gp.out_split = 0;            // partial sums stay fp32; the fix-up kernel writes the split form
            float* tile_ws = g.splitk_ws + ((size_t)(logical - g.tile_base) * g.split + slice) * (BM * BN);
            gemm_epilogue<0, NSUB>(gp, acc, lds, tile_ws - (m0 * BN + n0), nullptr, nullptr, m0e, n0, lane, w, w, 0);
        } else {
            if constexpr (EPI != 0) gemm_epilogue_split<MODE, NSUB>(g, acc, lds, g.Y, g.res, m0e, n0, lane, w);
            else gemm_epilogue<MODE, NSUB>(g, acc, lds, g.Y, g.bias, g.res, m0e, n0, lane, w, w, 0);
        }
    };
    finish(std::integral_constant<int, 0>());
    finish(std::integral_constant<int, 1>());
}

template <int MODE, bool SPLITK>
static void launch_w64_mode(const GemmArgs& g, dim3 grid, hipStream_t s) {
    if constexpr (MODE == 1 || MODE == 2) {
        // the TDS block's layers: split-form output under the range guard (MODE 2: split-form residual too)
        const bool y_ok = g.ldy % 4 == 0 && g.ldy < (1 << 21) && (reinterpret_cast<uintptr_t>(g.Y) & 15) == 0 && g.N % W_BN == 0;
        const bool r_ok = MODE != 2 || (g.res_split && g.ldres % 4 == 0 && g.ldres < (1 << 21) && (reinterpret_cast<uintptr_t>(g.res) & 15) == 0);
        if (g.out_split && g.range_flag && y_ok && r_ok) {
            hipLaunchKernelGGL((gemm_w64_kernel<MODE, SPLITK, 1>), grid, dim3(256), 0, s, g);
            return;
        }
    }
    hipLaunchKernelGGL((gemm_w64_kernel<MODE, SPLITK, 0>), grid, dim3(256), 0, s, g);
}

void launch_gemm_w64(const GemmArgs& g0, int mode, bool splitk, dim3 grid, hipStream_t s) {
    GemmArgs g = g0;
    const int pct = opt(OPT_GEMM_W64_STAGGER);
    if (pct > 0 && grid.x >= 2u * (unsigned)device_cus()) {          // (at least two rounds: one round has nothing to be out of phase with)
        const double tile_us = 1.39 * (g.K / 32) + 12.0;              // scripts/bench_gemm_f16x3_fit.py: K step + fixed cost per round
        g.stagger_ticks = (int)(tile_us * pct);                       // us * pct / 100 * 100 ticks per us
        g.stagger_blocks = device_cus();
    }
    if (splitk) {
        switch (mode) {
            case 0: launch_w64_mode<0, true>(g, grid, s); break;
            case 1: launch_w64_mode<1, true>(g, grid, s); break;
            case 2: launch_w64_mode<2, true>(g, grid, s); break;
            default: launch_w64_mode<3, true>(g, grid, s); break;
        }
        return;
    }
    switch (mode) {
        case 0: launch_w64_mode<0, false>(g, grid, s); break;
        case 1: launch_w64_mode<1, false>(g, grid, s); break;
        case 2: launch_w64_mode<2, false>(g, grid, s); break;
        default: launch_w64_mode<3, false>(g, grid, s); break;
    }
}

}  // namespace tal
