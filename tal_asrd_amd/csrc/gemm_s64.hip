// The fp16x3 dense layer for SHORT inputs (a few hundred to a few thousand rows): workgroup tile 64 rows x 80 columns x 32 k,
// 4 waves x (16 rows x 80 columns) of v_mfma_f32_16x16x32_f16.  The TDS block's 1x1 Conv1d pair (tal/asr/models.py:312-318,
// 330) on clips of seconds to minutes.
//
// Why: a 30-second clip has 1,501 / 751 / 376 rows in the three stages.  On 128 x 160 tiles that is 60 / 42 / 27 tiles for
// 256 CUs, so launch_gemm cut every tile along K into 4-6 slices and ran a fix-up kernel behind each layer (13-16 us +
// 7-9 us per layer, 22 layers: 60 % of the clip, profiles/r2_clip_30s_kernel_sequence.txt).  Here the same layer is
// 240 / 168 / 108 whole tiles: no slices, no scratch traffic, no second launch, and the summation order does not depend on
// the launch geometry.  A tile streams (64 + 80) x K x 4 bytes from L2 -- 461 KB at K = 800, 3.4 us at the 64 B/clk a CU
// gets from its L2 -- beside 15 MFMAs of 16 cycles per K step (240 cycles against 288 for the operands): the kernel is
// bound by the L2 -> LDS path by design, which is the price of 240 tiles instead of 60.
//
// Structure:
//   * operands go L2 -> LDS by LDS-DMA (inline asm, counted vmcnt: see gemm_w64.hip for why not the builtin), FOUR buffers
//     of 18 KB: tiles kt + 2 .. kt + 4 are in flight while tile kt is multiplied (a tile's MFMAs are 0.1 us, an L2 round
//     trip is several times that);
//   * the fragments of tile kt + 1 are read from LDS into a second register set while tile kt is multiplied from the
//     first (12 ds_read_b128 per wave and K step): with one wave per SIMD nothing else hides an LDS round trip; these reads
//     and the LDS-DMA loads of tile kt + 4 are issued one behind each of the step's 15 MFMAs (issued in front of them they
//     cost 1.5-2.5 us per layer: 24.1 -> 21.0 us per layer pair at K = 800, 35.4 -> 30.8 at 1440);
//   * ONE barrier per K step; same operand geometry and source-side XOR swizzle as gemm_glds_kernel;
//   * MFMA operand A = 16 W rows (output columns), B = 16 X rows: a lane ends up with 4 consecutive output columns of one
//     row -- the epilogue (bias, ReLU / ReZero residual, hi / lo split, range guard) runs on registers and stores 8 or 16
//     bytes per lane and column block, no LDS staging.
// Arithmetic per output: hi*hi in one fp32 accumulator (started from the bias), hi*lo + lo*hi in a second one, combined as
// acc + 2^-11 accx like the long-input kernels; K order inside an MFMA differs from the 32x32x16 form, so results agree
// with those kernels to fp32 rounding, not bitwise.
#include <type_traits>

#include "gemm_common.h"

namespace tal {

namespace {

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));

// NWV waves per workgroup, 16 rows each: 4 (64 x 80 tiles) or 2 (32 x 80 tiles: launches with so few tiles that half of the CUs
// would stay idle -- a CU streams its operands at ~36 B/clk whatever it computes, so more CUs with 112 instead of 144 operand rows
// per K step each are faster, profiles/r4_gemm_s64_order_depth.txt)
constexpr int S_BN = 80, S_NJ = S_BN / 16, S_NBUF = 4;
template <int NWV>
struct SGeo {
    static constexpr int BM = 16 * NWV, ROWS = BM + S_BN;
    static constexpr int CHUNKS = ROWS / 8;                        // 18 / 14 wave-loads of 1 KB (8 rows x 128 B) per K step
    static constexpr int PER_WAVE = (CHUNKS + NWV - 1) / NWV;      // 5 / 7; chunk i = w + NWV t; i >= CHUNKS (4 waves: t = 4 of waves 2, 3) lands
                                                                   // in a dummy 2 KB behind the buffers (every wave issues the same number
                                                                   // of loads: one vmcnt schedule for all)
    static constexpr int AT = BM / 8 / NWV;                        // t < 2: X rows
    static constexpr int BUF_FLOATS = ROWS * 32;
};

struct SFrags {
    f16x8 xh, xl;                // 16 X rows of this wave
    f16x8 wh[S_NJ], wl[S_NJ];    // 5 blocks of 16 W rows
};

}  // namespace

template <int MODE, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void gemm_s64_kernel(const GemmArgs g) {
    static_assert(MODE == 1 || MODE == 2, "relu or residual layer");
    using GEO = SGeo<NWV>;
    constexpr int NJ = S_NJ, BM = GEO::BM, BN = S_BN, PER_WAVE = GEO::PER_WAVE, BUF = GEO::BUF_FLOATS, S_AT = GEO::AT, S_CHUNKS = GEO::CHUNKS;
    __shared__ __attribute__((aligned(16))) float lds[S_NBUF * GEO::BUF_FLOATS + 512];      // 75,776 B (4 waves): two workgroups per CU

    const unsigned logical = logical_tile();
    unsigned tile_m, tile_n;
    if (g.n_major) {
        tile_n = g.tiles_m == 1 ? logical : __umulhi(logical, g.tiles_m_magic);
        tile_m = logical - tile_n * (unsigned)g.tiles_m;
    } else {
        tile_m = g.tiles_n == 1 ? logical : __umulhi(logical, g.tiles_n_magic);
        tile_n = logical - tile_m * (unsigned)g.tiles_n;
    }
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = (int)tile_n * BN;
    const int64_t M = g.M;
    const int K = g.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = wave_id();

    // operand loads: chunk = 8 rows x 128 B; slot s of LDS row r holds the row's logical 16-byte column s ^ ((r >> 1) & 7)
    const int sub = lane >> 3, srccol = ((lane & 7) ^ (((w & 1) * 4 + (lane >> 4)) & 7)) * 4;
    const int a_rows = (int)((M - m0) < BM ? (M - m0) : BM) - 1;
    auto make_rsrc = [](const float* p, unsigned nrec) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        u32x4s r;
        r[0] = __builtin_amdgcn_readfirstlane((uint32_t)v);
        r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        r[2] = nrec;
        r[3] = 0x00020000u;
        return r;
    };
    constexpr unsigned NREC = 0x7fffffffu;
    const u32x4s rs_a = make_rsrc(g.A + m0 * g.lda, NREC), rs_w = make_rsrc(g.W + (int64_t)n0 * g.ldw, NREC);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds) + (unsigned)w * 1024u;
    int voff[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int row = 8 * (w + NWV * t) + sub;
        voff[t] = t < S_AT ? (int)((min(row, a_rows) * g.lda + srccol) * 4) : (int)((min(row - BM, BN - 1) * g.ldw + srccol) * 4);
    }
    const bool last_chunk = w + NWV * (PER_WAVE - 1) < S_CHUNKS;          // wave-uniform: this wave's last chunk exists
    auto dma = [&](int t, unsigned nrec, int bufoff, int kofs) {
        const unsigned dst = (t < PER_WAVE - 1 || last_chunk) ? lds_base + (unsigned)(bufoff * 4 + t * (NWV * 1024))
                                                              : lds_base + (unsigned)(S_NBUF * BUF * 4 - 2048);
        u32x4s rs = t < S_AT ? rs_a : rs_w;
        rs[2] = nrec;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff[t]), "s"(rs), "s"(kofs) : "memory");
    };

    const int nk = K / BK;
    // prologue: tiles 0..3 requested first of all
#pragma unroll
    for (int b = 0; b < S_NBUF; ++b) {
        const unsigned nrec = b < nk ? NREC : 0u;
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) dma(t, nrec, b * BUF, b * (BK * 4));
    }

    // a lane's outputs: row m0 + 16 w + (lane & 15), columns n0 + 16 j + 4 (lane >> 4) + {0..3}
    const int col = lane & 15, q4 = lane >> 4;
    f32x4 acc[NJ], accx[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        acc[j] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n0 + 16 * j + 4 * q4) : z;
        accx[j] = z;
    }
    // the residual of this lane's outputs is requested before the K loop (MODE 2): 20 registers against an exposed round trip
    const int64_t row = m0 + 16 * w + col;
    const bool row_ok = row < M;
    const int64_t rrow = row_ok ? row : M - 1;
    u32x2s rh[NJ], rl[NJ];
    f32x4 rf[NJ];
    if (MODE == 2) {
        if (g.res_split) {
            const char* rp = reinterpret_cast<const char*>(g.res + rrow * g.ldres);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = n0 + 16 * j + 4 * q4;
                rh[j] = *reinterpret_cast<const u32x2s*>(rp + (c >> 5) * 128 + (c & 31) * 2);
                rl[j] = *reinterpret_cast<const u32x2s*>(rp + (c >> 5) * 128 + (c & 31) * 2 + 64);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) rf[j] = *reinterpret_cast<const f32x4*>(g.res + rrow * g.ldres + n0 + 16 * j + 4 * q4);
        }
    }

    // fragment addresses: row (lane & 15) of a 16-row block, logical 16-byte slot q4 (hi halves) / 4 + q4 (lo halves)
    const int fsw = (col >> 1) & 7;
    const int x_lane = (16 * w + col) * 32, w_lane = (BM + col) * 32;
    const int sh = ((q4) ^ fsw) * 4, sl = ((4 + q4) ^ fsw) * 4;
    SFrags f[2];
    auto read_frags = [&](SFrags& d, int bufoff) {
        d.xh = *reinterpret_cast<const f16x8*>(lds + bufoff + x_lane + sh);
        d.xl = *reinterpret_cast<const f16x8*>(lds + bufoff + x_lane + sl);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            d.wh[j] = *reinterpret_cast<const f16x8*>(lds + bufoff + w_lane + j * 16 * 32 + sh);
            d.wl[j] = *reinterpret_cast<const f16x8*>(lds + bufoff + w_lane + j * 16 * 32 + sl);
        }
    };

    // The bias and residual registers are consumed HERE as far as hipcc can tell: its wait for them (it counts only the loads
    // it can see, so it is a vmcnt(0)) lands in front of the K loop instead of inside it -- one round trip for the first four
    // tiles, the bias and the residual together.
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        asm volatile("" : "+v"(acc[j]));
        if (MODE == 2) {
            if (g.res_split) asm volatile("" : "+v"(rh[j]), "+v"(rl[j]));
            else asm volatile("" : "+v"(rf[j]));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(f[0], 0);
    // step on tile kt (fragment set B & 1): tile kt + 1 (LDS buffer (B + 1) % 4) goes into the other set, tile kt + 4 is
    // requested into buffer B -- every wave has read tile kt out of it before it arrived at this step's barrier
    auto step = [&](auto bc, int kt) {
        constexpr int B = decltype(bc)::value, cur = B & 1, nxt = cur ^ 1;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER_WAVE) : "memory");
        __builtin_amdgcn_s_barrier();
        // One wave per SIMD: whatever is issued in front of the MFMAs is matrix time lost.  The 12 fragment reads of tile kt + 1 and
        // the 5 LDS-DMA loads of tile kt + 4 are therefore issued BETWEEN the 15 MFMAs of tile kt, one behind each (an MFMA
        // occupies the pipe for 16 cycles and the issue port for 4; with two waves per workgroup a wave has 7 loads per step).
        const unsigned nrec = kt + 4 < nk ? NREC : 0u;
        const int kofs = (kt + 4) * (BK * 4);
        constexpr int nbuf = ((B + 1) % S_NBUF) * BUF;
        auto read_item = [&](int m) {          // m-th fragment read of the next tile: xh, xl, wh[0], wl[0], wh[1], ...
            if (m == 0) f[nxt].xh = *reinterpret_cast<const f16x8*>(lds + nbuf + x_lane + sh);
            else if (m == 1) f[nxt].xl = *reinterpret_cast<const f16x8*>(lds + nbuf + x_lane + sl);
            else if (m & 1) f[nxt].wl[(m - 2) >> 1] = *reinterpret_cast<const f16x8*>(lds + nbuf + w_lane + ((m - 2) >> 1) * 16 * 32 + sl);
            else f[nxt].wh[(m - 2) >> 1] = *reinterpret_cast<const f16x8*>(lds + nbuf + w_lane + ((m - 2) >> 1) * 16 * 32 + sh);
        };
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[cur].wh[j], f[cur].xh, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (3 * j < 12) read_item(3 * j);
            __builtin_amdgcn_sched_barrier(0);
            accx[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[cur].wh[j], f[cur].xl, accx[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (3 * j + 1 < 12) read_item(3 * j + 1);
            __builtin_amdgcn_sched_barrier(0);
            accx[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[cur].wl[j], f[cur].xh, accx[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (3 * j + 2 < 12) read_item(3 * j + 2);
            dma(j, nrec, B * BUF, kofs);
            if (j + NJ < PER_WAVE) dma(j + NJ, nrec, B * BUF, kofs);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int kt = 0;
    for (; kt + 4 <= nk; kt += 4) {
        step(std::integral_constant<int, 0>(), kt);
        step(std::integral_constant<int, 1>(), kt + 1);
        step(std::integral_constant<int, 2>(), kt + 2);
        step(std::integral_constant<int, 3>(), kt + 3);
    }
    if (kt < nk) step(std::integral_constant<int, 0>(), kt);
    if (kt + 1 < nk) step(std::integral_constant<int, 1>(), kt + 1);
    if (kt + 2 < nk) step(std::integral_constant<int, 2>(), kt + 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (dropped loads included; the residual registers are older than every LDS-DMA)

    // ---- epilogue on registers ----
    const float alpha = g.alpha, s1 = 1.0f / 2048.0f;
    float amax = 0.f;
    char* yrow = reinterpret_cast<char*>(g.Y + rrow * g.ldy);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(accx[j][e], s1, acc[j][e]);
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(v[e], 0.f);
        } else {
            f32x4 r;
            if (g.res_split) {
                const f16x4 h4 = __builtin_bit_cast(f16x4, rh[j]), l4 = __builtin_bit_cast(f16x4, rl[j]);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf((float)l4[e], s1, (float)h4[e]);
            } else
                r = rf[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(alpha, v[e], r[e]);
        }
        const int c = n0 + 16 * j + 4 * q4;
        if (g.out_split) {
            amax = amax4(amax, v);
            f16x4 hi, lo;
            if (g.range_flag) {
                f16x2p h01, l01, h23, l23;
                split_f16x3_pair(v[0], v[1], h01, l01);
                split_f16x3_pair(v[2], v[3], h23, l23);
                hi = f16x4{h01[0], h01[1], h23[0], h23[1]};
                lo = f16x4{l01[0], l01[1], l23[0], l23[1]};
            } else {          // unguarded call: clamped halves
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 h, l;
                    split_f16x3(v[e], h, l);
                    hi[e] = h;
                    lo[e] = l;
                }
            }
            if (row_ok) {
                char* yp = yrow + (c >> 5) * 128 + (c & 31) * 2;
                *reinterpret_cast<f16x4*>(yp) = hi;
                *reinterpret_cast<f16x4*>(yp + 64) = lo;
            }
        } else if (row_ok)
            *reinterpret_cast<f32x4*>(yrow + c * 4) = v;
    }
    if (g.out_split && g.range_flag) note_range(row_ok ? amax : 0.f, g.range_flag);
}

// M, N, K, leading dimensions and pointers as launch_gemm checked them for the fp16x3 form; N % 80 == 0
bool gemm_s64_ok(const GemmArgs& g, int mode) {
    if (!(mode == 1 || mode == 2) || !g.f16x3 || g.N % S_BN != 0 || g.K % BK != 0) return false;
    if (((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W) | reinterpret_cast<uintptr_t>(g.Y)) & 15) != 0) return false;
    if (g.bias && (reinterpret_cast<uintptr_t>(g.bias) & 15) != 0) return false;
    if (g.ldy % 4 != 0 || g.lda >= (1 << 21) || g.ldw >= (1 << 21)) return false;
    if (mode == 2 && (!g.res || (reinterpret_cast<uintptr_t>(g.res) & 15) != 0 || g.ldres % 4 != 0)) return false;
    if ((g.out_split && g.N % 32 != 0) || (mode == 2 && g.res_split && g.ldres % 32 != 0)) return false;
    return true;
}

void launch_gemm_s64(GemmArgs g, int mode, hipStream_t s) {
    g.tiles_n = g.N / S_BN;
    g.tiles_n_magic = g.tiles_n > 1 ? (unsigned)((1ull << 32) / (unsigned)g.tiles_n) + 1u : 0u;
    // 32-row tiles (two waves per workgroup) while even those leave CUs idle: stage 3 of a 30-second clip is 108 tiles of 64 rows
    // on 256 CUs, 216 of 32.  opt gemm_s64_rows: 1 / 2 force 64 / 32 rows.
    const int rows_opt = opt(OPT_GEMM_S64_ROWS);
    const bool half = rows_opt == 2 || (rows_opt == 0 && cdiv(g.M, 32) * g.tiles_n <= (int64_t)device_cus());
    g.tiles_m = (int)cdiv(g.M, half ? 32 : 64);
    g.tiles_m_magic = g.tiles_m > 1 ? (unsigned)((1ull << 32) / (unsigned)g.tiles_m) + 1u : 0u;
    // An XCD (its own L2) takes a contiguous eighth of the tile order.  Fewer rows than columns: W (N x K) is the larger operand,
    // so the order keeps an XCD on a few W panels; otherwise on a few X panels (measured, profiles/r4_gemm_s64_order_depth.txt:
    // 30.5 -> 28.8 us per layer pair at 376 rows x 1440, 28.0 -> 26.6 at 751 x 1120; the other way round it loses as much).
    // opt gemm_s64_order: 1 / 2 force m-major / n-major.
    const int order = opt(OPT_GEMM_S64_ORDER);
    g.n_major = order == 1 ? 0 : order == 2 ? 1 : (g.M < (int64_t)g.N ? 1 : 0);
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
    if (half) {
        if (mode == 1) hipLaunchKernelGGL((gemm_s64_kernel<1, 2>), grid, dim3(128), 0, s, g);
        else hipLaunchKernelGGL((gemm_s64_kernel<2, 2>), grid, dim3(128), 0, s, g);
    } else {
        if (mode == 1) hipLaunchKernelGGL((gemm_s64_kernel<1, 4>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_s64_kernel<2, 4>), grid, dim3(256), 0, s, g);
    }
}

}  // namespace tal
