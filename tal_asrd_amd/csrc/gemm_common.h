// Shared pieces of the dense-layer kernels (gemm_f32.hip, gemm_w64.hip): the epilogues and the XCD-aware tile order.
#pragma once
#include "common.h"

namespace tal {

constexpr int BK = 32;

// ---------------------------------------------------------------------------------------------
// shared epilogue
// ---------------------------------------------------------------------------------------------
template <int MODE, int NSUB>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16 (&acc)[NSUB], float* lds, float* Y,
                                              const float* bias, const float* res, int64_t m0, int n0, int lane,
                                              int w, int wm, int wn) {
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    const int64_t M = g.M;
    const int N = g.N;
    const int colb = lane & 31;
    const int rowb = 4 * (lane >> 5);
    const float alpha = g.alpha;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float amax = 0.f;              // out_split: largest |value| this lane turned into halves
    constexpr int CW = 32 * NSUB;  // columns owned by one wave
    if (MODE == 4) {
        // Fused arg-max over this wave's CW columns (the [M, 6008] speaker logits are never written,
        // tal/baseline/reconcile.py:84).  For a fixed accumulator element e the 32 lanes of a half
        // wave hold 32 columns of one row; columns are visited in ascending order and a strict '>'
        // keeps the lowest index on ties, as torch.argmax does.
        float bcol[NSUB];
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            const int col = n0 + (wn * NSUB + j) * 32 + colb;
            bcol[j] = (bias && col < N) ? bias[col] : 0.f;
        }
        const int pcol = (n0 / CW) + wn;   // n0 is a multiple of BN = CW * WAVES_N
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                const int col = n0 + (wn * NSUB + j) * 32 + colb;
                const float v = acc[j][e] + bcol[j];
                if (col < N && v > best) {
                    best = v;
                    bi = col;
                }
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                const float ov = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ov > best || (ov == best && oi < bi)) {
                    best = ov;
                    bi = oi;
                }
            }
            const int64_t row = m0 + wm * 32 + rowb + (e & 3) + 8 * (e >> 2);
            if (colb == 0 && row < M) {
                g.part_val[row * g.part_ld + pcol] = best;
                g.part_idx[row * g.part_ld + pcol] = bi;
            }
        }
        return;
    }
    const bool vec_ok = (g.ldy % 4 == 0) && (MODE != 2 || g.ldres % 4 == 0) &&
                        ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) &&
                        (MODE != 2 || (reinterpret_cast<uintptr_t>(res) & 15) == 0);
    if (vec_ok) {
        // Each wave only touches its own LDS slice, so no barrier inside the epilogue.
        float* stage = lds + w * (16 * CW);
        const int n_base = n0 + wn * CW;
        const bool cols_full = n_base + CW <= N;  // wave-uniform
        // descriptors of this wave's 16 x CW windows (one per half) of Y / res, extent = the rows inside M (the
        // row offset of a half must not travel in the scalar offset: it is not part of the range check),
        // and of its bias slice
        const bool buf_ok = g.ldy < (1 << 21) && (MODE != 2 || g.ldres < (1 << 21));
        auto uptr = [](const float* p) {
            const uint64_t v = reinterpret_cast<uint64_t>(p);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
        };
        __amdgpu_buffer_rsrc_t rs_y[2], rs_res[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int64_t row0 = m0 + wm * 32 + half * 16;
            const int rows_valid = (int)(M - row0 < 0 ? 0 : (M - row0 > 16 ? 16 : M - row0));
            const int ext_y = rows_valid > 0 ? (int)(((rows_valid - 1) * g.ldy + CW) * 4) : 0;
            const int ext_r = rows_valid > 0 && MODE == 2 ? (int)(((rows_valid - 1) * g.ldres + CW) * 4) : 0;
            rs_y[half] = __builtin_amdgcn_make_buffer_rsrc(uptr(Y + row0 * g.ldy + n_base), 0, __builtin_amdgcn_readfirstlane(ext_y), 0x00020000);
            rs_res[half] = __builtin_amdgcn_make_buffer_rsrc(uptr(MODE == 2 ? res + row0 * g.ldres + n_base : Y), 0,
                                                             __builtin_amdgcn_readfirstlane(ext_r), 0x00020000);
        }
        __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(uptr(bias ? bias + n_base : Y), 0, CW * 4, 0x00020000);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int j = 0; j < NSUB; ++j)
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = half * 8 + e8;
                    const int r = rowb + (e8 & 3) + 8 * (e8 >> 2);  // 0..15 inside this half
                    stage[r * CW + j * 32 + colb] = acc[j][e];
                }
            if (cols_full && buf_ok) {
                // Interior fast path.  Output, residual and bias go through buffer descriptors (SGPRs) whose
                // extent is exactly this wave's valid rows: rows past M need no clamp and no predicate (loads
                // return 0, stores are dropped), addresses are 32-bit lane offsets + a scalar row offset, and
                // the buffer form is the cheap one to issue.  All loads are issued before any is consumed.
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                int r = lane / (CW / 4), c4 = lane - r * (CW / 4);
                int offy[2 * NSUB], offr[2 * NSUB], offb[2 * NSUB], ldsv[2 * NSUB];
                f32x4 rv[2 * NSUB], bv[2 * NSUB];
#pragma unroll
                for (int t = 0; t < 2 * NSUB; ++t) {
                    offy[t] = (r * (int)g.ldy + c4 * 4) * 4;
                    offr[t] = (r * (int)g.ldres + c4 * 4) * 4;
                    offb[t] = c4 * 16;
                    ldsv[t] = r * CW + c4 * 4;
                    c4 += 64 - CW / 4;          // lane + 64 (t + 1) = (r + 1) * 40 + c4 + 24
                    r += 1;
                    if (c4 >= CW / 4) {
                        c4 -= CW / 4;
                        r += 1;
                    }
                }
                typedef unsigned u32x2r __attribute__((ext_vector_type(2)));
                u32x2r rh[2 * NSUB], rl[2 * NSUB];      // split residual: 4 hi halves, 4 lo halves
#pragma unroll
                for (int t = 0; t < 2 * NSUB; ++t) {
                    if (MODE == 2 && g.res_split) {
                        // the window starts on a 32-column block (n_base % 160 == 0): column cw -> block cw / 32, slot cw % 32
                        const int cw = offb[t] >> 2;
                        const int so = offr[t] - offb[t] + (cw >> 5) * 128 + (cw & 31) * 2;
                        rh[t] = __builtin_amdgcn_raw_buffer_load_b64(rs_res[half], so, 0, 0);
                        rl[t] = __builtin_amdgcn_raw_buffer_load_b64(rs_res[half], so + 64, 0, 0);
                    } else if (MODE == 2)
                        rv[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res[half], offr[t], 0, 0));
                    bv[t] = bias ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, offb[t], 0, 0)) : zero4;
                }
#pragma unroll
                for (int t = 0; t < 2 * NSUB; ++t) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&stage[ldsv[t]]) + bv[t];
                    if (MODE == 1) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    if (MODE == 2 && g.res_split) {
                        const f16x4 h4 = __builtin_bit_cast(f16x4, rh[t]), l4 = __builtin_bit_cast(f16x4, rl[t]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) rv[t][q] = (float)h4[q] + (float)l4[q] * (1.0f / 2048.0f);
                    }
                    if (MODE == 2) v = rv[t] + alpha * v;
                    if (MODE == 3 && (g.scale_cols == 0 || n_base + (offb[t] >> 2) < g.scale_cols)) v = alpha * v;
                    if (g.out_split) {
                        // the next layer's A operand: hi / lo fp16 halves of these 4 columns, in the row's 128-byte
                        // K blocks [32 hi | 32 lo]  (column c of the wave's window -> block c / 32, slot c % 32)
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        f16x4 hi, lo;
                        amax = amax4(amax, v);
                        if (g.range_flag) {       // guarded call: packed conversions, no clamps (an out-of-range value raises the flag)
                            f16x2p h01, l01, h23, l23;
                            split_f16x3_pair(v.x, v.y, h01, l01);
                            split_f16x3_pair(v.z, v.w, h23, l23);
                            hi = {h01[0], h01[1], h23[0], h23[1]};
                            lo = {l01[0], l01[1], l23[0], l23[1]};
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                _Float16 h, l;
                                split_f16x3(v[q], h, l);
                                hi[q] = h;
                                lo[q] = l;
                            }
                        }
                        const int cw = offb[t] >> 2;                                  // column inside the window
                        const int so = offy[t] - offb[t] + (cw >> 5) * 128 + (cw & 31) * 2;   // row part + block + slot
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rs_y[half], so, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rs_y[half], so + 64, 0, 0);
                        continue;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y[half], offy[t], 0, 0);
                }
                if (g.out_split && half == 1) note_range(amax, g.range_flag);
                continue;
            }
#pragma unroll
            for (int t = 0; t < 2 * NSUB; ++t) {
                const int i = lane + 64 * t;
                const int r = i / (CW / 4);
                const int c = (i - r * (CW / 4)) * 4;
                const int64_t row = m0 + wm * 32 + half * 16 + r;
                const int col = n_base + c;
                if (row < M && col < N) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&stage[r * CW + c]);
                    if (col + 3 < N) {
                        if (bias) v += *reinterpret_cast<const f32x4*>(bias + col);
                        if (MODE == 1) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        if (MODE == 2) v = *reinterpret_cast<const f32x4*>(res + row * g.ldres + col) + alpha * v;
                        if (MODE == 3 && (g.scale_cols == 0 || col < g.scale_cols)) v = alpha * v;
                        *reinterpret_cast<f32x4*>(Y + row * g.ldy + col) = v;
                    } else {
                        for (int q = 0; q < 4 && col + q < N; ++q) {
                            float x = v[q] + (bias ? bias[col + q] : 0.f);
                            if (MODE == 1) x = fmaxf(x, 0.f);
                            if (MODE == 2) x = res[row * g.ldres + col + q] + alpha * x;
                            if (MODE == 3 && (g.scale_cols == 0 || col + q < g.scale_cols)) x = alpha * x;
                            Y[row * g.ldy + col + q] = x;
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            const int col = n0 + (wn * NSUB + j) * 32 + colb;
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + wm * 32 + rowb + (e & 3) + 8 * (e >> 2);
                if (row < M) {
                    float v = acc[j][e] + bv;
                    if (MODE == 1) v = fmaxf(v, 0.f);
                    if (MODE == 2) v = res[row * g.ldres + col] + alpha * v;
                    if (MODE == 3 && (g.scale_cols == 0 || col < g.scale_cols)) v = alpha * v;
                    Y[row * g.ldy + col] = v;
                }
            }
        }
    }
}

// The TDS block's two hot epilogues (EPI = 1): relu -> split form (MODE 1) and split-form residual + alpha * v -> split form
// (MODE 2), both under the range guard; the bias is already in the accumulators (gemm_glds_kernel starts them from it).
// On a gfx950 SIMD the vector ALU and the matrix pipe do not overlap, and all tiles of a launch reach their epilogue
// together (same K), so every vector instruction of a tile is exposed time.  The generic epilogue spends ~50 of them per
// 16-byte output piece on address arithmetic and run-time flags (per wave and tile ~1000, against 1350 MFMAs at K = 1440;
// 10-15 us of fixed cost per round of tiles, scripts/bench_gemm_f16x3_fit.py).  Here a store iteration is (rg, b) = (8-row
// group, 32-column block): lane -> row rg * 8 + lane / 8, columns b * 32 + (lane % 8) * 4, so every address is one lane
// constant + an immediate -- no vector arithmetic left but the conversions themselves (7-8 us per round for MODE 1, the
// same as a plain fp32 store).  Rows past M are outside the buffer descriptors (loads return 0, stores are dropped).
template <int MODE, int NSUB>
__device__ __forceinline__ void gemm_epilogue_split(const GemmArgs& g, const f32x16 (&acc)[NSUB], float* lds, float* Y,
                                                    const float* res, int64_t m0, int n0, int lane, int w) {
    static_assert(MODE == 1 || MODE == 2, "relu or residual");
    constexpr int CW = 32 * NSUB;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int64_t M = g.M;
    const int colb = lane & 31, rowb = 4 * (lane >> 5);
    float* stage = lds + w * (16 * CW);
    auto uptr = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    const int lr = lane >> 3, lp = lane & 7;
    const int ldy4 = (int)g.ldy * 4, ldr4 = (int)g.ldres * 4;
    // lane offsets of the two 8-row groups: row part + this lane's 8-byte slot inside a 64-byte hi (lo) run
    const int vy0 = lr * ldy4 + lp * 8, vy1 = vy0 + 8 * ldy4;
    const int vr0 = lr * ldr4 + lp * 8, vr1 = vr0 + 8 * ldr4;
    const float* sp = stage + lr * CW + lp * 4;
    const float alpha = g.alpha;
    const f32x2 al2 = {alpha, alpha}, s11 = {1.0f / 2048.0f, 1.0f / 2048.0f};
    float amax = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int64_t row0 = m0 + w * 32 + half * 16;
        const int rows_valid = (int)(M - row0 < 0 ? 0 : (M - row0 > 16 ? 16 : M - row0));
        const int ext_y = rows_valid > 0 ? (rows_valid - 1) * ldy4 + CW * 4 : 0;
        const int ext_r = rows_valid > 0 && MODE == 2 ? (rows_valid - 1) * ldr4 + CW * 4 : 0;
        __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(uptr(Y + row0 * g.ldy + n0), 0, __builtin_amdgcn_readfirstlane(ext_y), 0x00020000);
        __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(uptr(MODE == 2 ? res + row0 * g.ldres + n0 : Y), 0,
                                                                      __builtin_amdgcn_readfirstlane(ext_r), 0x00020000);
        u32x2 rh[2 * NSUB], rl[2 * NSUB];
        if (MODE == 2) {
#pragma unroll
            for (int t = 0; t < 2 * NSUB; ++t) {
                const int rg = t / NSUB, b = t % NSUB;
                rh[t] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, (rg ? vr1 : vr0) + b * 128, 0, 0);
                rl[t] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, (rg ? vr1 : vr0) + b * 128 + 64, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int r = rowb + (e8 & 3) + 8 * (e8 >> 2);
                stage[r * CW + j * 32 + colb] = acc[j][half * 8 + e8];
            }
        f32x4 sv[2 * NSUB];
#pragma unroll
        for (int t = 0; t < 2 * NSUB; ++t) sv[t] = *reinterpret_cast<const f32x4*>(sp + (t / NSUB) * 8 * CW + (t % NSUB) * 32);
#pragma unroll
        for (int t = 0; t < 2 * NSUB; ++t) {
            const int rg = t / NSUB, b = t % NSUB;
            f32x2 va = {sv[t][0], sv[t][1]}, vb = {sv[t][2], sv[t][3]};
            if (MODE == 1) {
                va = __builtin_elementwise_max(va, f32x2{0.f, 0.f});
                vb = __builtin_elementwise_max(vb, f32x2{0.f, 0.f});
            } else {
                const f16x4 h4 = __builtin_bit_cast(f16x4, rh[t]), l4 = __builtin_bit_cast(f16x4, rl[t]);
                const f32x2 ha = {(float)h4[0], (float)h4[1]}, hb = {(float)h4[2], (float)h4[3]};
                const f32x2 la = {(float)l4[0], (float)l4[1]}, lb = {(float)l4[2], (float)l4[3]};
                va = al2 * va + (la * s11 + ha);
                vb = al2 * vb + (lb * s11 + hb);
            }
            amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(va[0]), __builtin_fabsf(va[1])));
            amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(vb[0]), __builtin_fabsf(vb[1])));
            f16x2p h01, l01, h23, l23;
            split_f16x3_pair(va[0], va[1], h01, l01);
            split_f16x3_pair(vb[0], vb[1], h23, l23);
            const f16x4 hi = {h01[0], h01[1], h23[0], h23[1]}, lo = {l01[0], l01[1], l23[0], l23[1]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rs_y, (rg ? vy1 : vy0) + b * 128, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rs_y, (rg ? vy1 : vy0) + b * 128 + 64, 0, 0);
        }
    }
    note_range(amax, g.range_flag);
}

// XCD-aware bijective remap: XCD x (= blockIdx % 8) walks a contiguous range of logical tiles,
// N-tiles of one M-tile first, so the A panel is re-read from that XCD's L2.
__device__ __forceinline__ unsigned logical_tile_of(unsigned nb, unsigned bid) {
    const unsigned xcd = bid & 7u, q = nb >> 3, r = nb & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ unsigned logical_tile() { return logical_tile_of(gridDim.x, blockIdx.x); }

}  // namespace tal
