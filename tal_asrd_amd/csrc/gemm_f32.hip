// fp32 dense layer on the CDNA4 matrix cores: Y = epilogue(X . W^T + b), optionally batched.
//
// Covers nn.Linear and the 1x1 Conv1d pair of TDSBlock (tal/asr/models.py:312-318,
// 86.9 % of the encoder's MACs), the encoder projections (:100,131), the SD / speaker
// heads (:421-422,143-146), the decoder's projections / FFN (:493-499) and -- through the
// batched form with per-(batch, head) strides -- the QK^T and PV contractions of
// nn.MultiheadAttention (:514-518).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bit-for-bit an fmaf chain),
// so results stay inside the 1e-3 fp32 logit tolerance of BASELINE.json without any
// reduced-precision trick.  Roofline: 157.3 TFLOP/s (fp32 matrix peak of MI355X; a pure-MFMA
// loop reaches 155 on this chip, scripts/ubench/mfma_peak.hip).
//
// Kernels, all 4 waves x (32 rows x 32*NSUB columns) of 32x32 accumulators:
//   gemm_glds_kernel    128(M) x 160(N) x 32(K), the hot one (M > 512, K % 32 == 0).  Operand
//       tiles go HBM/L2 -> LDS directly (LDS-DMA `buffer_load_dwordx4 ... lds`: SGPR descriptor,
//       32-bit lane offset, no VGPR staging, no ds_write), double-buffered, ONE barrier per K step;
//       the LDS image is lane-linear, so bank conflicts are removed by an XOR swizzle applied to the
//       per-lane SOURCE offset and again on the fragment read.  160 divides every TDS width
//       (800/1120/1440): no N-tail waste.  The tiles of the last partial scheduling round can be cut
//       along K (SPLITK launches + gemm_splitk_fixup_kernel) when the caller provides scratch.
//   gemm_splitk4_kernel 32 x 32 tile, 4 waves split K inside the workgroup: the decoder's short,
//       latency-bound problems (M <= 512 rows).
//   gemm_nt_f32_kernel<.,4,5>  128 x 160, register-staged, single LDS buffer: K tails
//       (K % 32 != 0; only small test models);  <.,1,1>  32 x 128: small M with mode 4 / odd shapes.
// Epilogue (shared): the wave's tile is staged through LDS 16 rows at a time and written as whole
// rows with 16-byte buffer stores whose descriptor ends at the last valid row; bias / ReLU /
// ReZero-residual / scale are fused; all bias and residual loads of a half tile are issued before
// any of them is consumed.
// Cost model (measured): the matrix pipe and the vector ALU of a SIMD do not
// overlap, so every VALU / vector-memory instruction here is matrix time lost -- hence SGPR
// descriptors, 32-bit offsets and scalar address arithmetic wherever possible.
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

// ---------------------------------------------------------------------------------------------
// hot kernel: direct-to-LDS operand loads, double buffer, one barrier per K step
// ---------------------------------------------------------------------------------------------
// (the fp16x3 form carries two accumulator sets: it is held to 256 registers = 2 workgroups per CU explicitly)
template <int MODE, int NSUB, bool SPLITK, int STAGE, bool F16X3 = false, int EPI = 0>
__global__ __launch_bounds__(256, F16X3 ? 2 : 1) void gemm_glds_kernel(const GemmArgs g) {
    // Set-up and epilogue are short VALU / memory sequences; the co-resident workgroup is usually deep in
    // its MFMA loop, and at equal priority every one of these instructions queues behind a 64-cycle MFMA.
    __builtin_amdgcn_s_setprio(3);
    constexpr int BM = 128, BN = 32 * NSUB;
    constexpr int ROWS = BM + BN;                 // 288 operand rows of 32 floats (128 B) per K step
    constexpr int CHUNKS = ROWS / 8;              // 36 wave-loads of 1 KB (8 rows) each
    constexpr int PER_WAVE = CHUNKS / 4;          // 9: t < 4 -> A rows, t >= 4 -> W rows
    __shared__ __attribute__((aligned(16))) float lds[2 * ROWS * 32];  // 73,728 B -> 2 workgroups / CU

    // XCD-remapped tile index.  SPLITK launches carry the whole-round tiles (blocks < tile_base) AND, behind
    // them in dispatch order, the K slices of the last partial round (block -> (tile, K slice)), so the
    // slices fill slots as the whole tiles drain instead of waiting for a kernel boundary.
    const bool is_slice = SPLITK && blockIdx.x >= (unsigned)g.tile_base;
    const unsigned sbid = blockIdx.x - (unsigned)g.tile_base;
    const int slice = is_slice ? (int)(sbid / (unsigned)g.tail_tiles) : 0;
    const unsigned logical = is_slice ? (unsigned)g.tile_base + sbid % (unsigned)g.tail_tiles
                                      : (SPLITK ? logical_tile_of((unsigned)g.tile_base, blockIdx.x) : logical_tile());
    // (tile row by a multiply-high with the launch's magic number: a run-time division is ~40 vector instructions)
    const unsigned tile_m = g.tiles_n == 1 ? logical : __umulhi(logical, g.tiles_n_magic);
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = (int)(logical - tile_m * (unsigned)g.tiles_n) * BN;
    const int64_t M = g.M;
    const int N = g.N, K = g.K;
    const int z1 = EPI ? 0 : (int)blockIdx.y / g.nb2, z2 = EPI ? 0 : (int)blockIdx.y % g.nb2;    // (EPI: never batched)
    const float* A = g.A + z1 * g.a_s1 + z2 * g.a_s2;
    const float* W = g.W + z1 * g.w_s1 + z2 * g.w_s2;
    float* Y = g.Y + z1 * g.y_s1 + z2 * g.y_s2;
    const float* bias = g.bias ? g.bias + z2 * g.bias_s2 : nullptr;
    const float* res = g.res ? g.res + z1 * g.r_s1 + z2 * g.r_s2 : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = wave_id();
    const int wm = w, wn = 0;

    // Wave w loads chunks i = w + 4t.  A chunk is 8 rows x 128 B; lane l lands at LDS byte
    // i*1024 + l*16 = row r = 8i + l/8, 16-byte slot l%8.  Slot s of row r holds the row's logical
    // 16-byte column s ^ ((r >> 1) & 7): the swizzle is applied to the global source address, the LDS
    // destination stays linear as the instruction requires.  gfx950 serves a ds_read_b128 sixteen
    // lanes (256 bytes, 64 banks) at a time: with (r >> 1) & 7 sixteen consecutive rows cover all 64
    // banks (SQ_LDS_BANK_CONFLICT = 0); the (r & 7) form, enough for 8-lane phases, measured 50 %
    // conflict cycles.  (r >> 1) & 7 = (4 (i & 1) + (l >> 4)) & 7, and i & 1 = w & 1: a per-lane constant.
    const int sub = lane >> 3;                    // row inside the chunk
    const int srccol = ((lane & 7) ^ (((w & 1) * 4 + (lane >> 4)) & 7)) * 4;    // logical float column this lane fetches
    const int a_rows = (int)((M - m0) < BM ? (M - m0) : BM) - 1;
    const int b_rows = ((N - n0) < BN ? (N - n0) : BN) - 1;
    const float* src[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int row = 8 * (w + 4 * t) + sub;    // 0..287
        if (t < 4)
            src[t] = A + (m0 + min(row, a_rows)) * g.lda + srccol;
        else
            src[t] = W + (int64_t)(n0 + min(row - BM, b_rows)) * g.ldw + srccol;
    }
    // STAGE 2 (default): the LDS-DMA loads use the buffer form -- descriptor in SGPRs, one 32-bit lane
    // offset, the K offset in an SGPR -- instead of 64-bit lane addresses.  On gfx950 every vector-memory
    // instruction costs the SIMD matrix-pipe issue cycles and the global form (plus the 64-bit VALU add
    // per load that advances its address) costs the most: K-loop step 4.77 -> 4.43 us of an ideal 4.27
    // (scripts/ubench/mfma_mix.hip isolates the ingredients; scripts/bench_gemm_fit.py the real kernel).
    // STAGE 0 keeps the global form for leading dimensions whose lane offsets do not fit 32 bits.
    // (64-bit multiplies run on the VALU even for uniform values; the descriptors must sit in SGPRs)
    auto uniform_ptr = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(A + m0 * g.lda), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(W + (int64_t)n0 * g.ldw), 0, 0x7fffffff, 0x00020000);
    int voff[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int row = 8 * (w + 4 * t) + sub;
        voff[t] = t < 4 ? (int)((min(row, a_rows) * g.lda + srccol) * 4) : (int)((min(row - BM, b_rows) * g.ldw + srccol) * 4);
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) {
            float* dst = lds + buf * (ROWS * 32) + (w + 4 * t) * 256;
            if (STAGE == 2) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass drops the whole kernel stub when it meets this builtin)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(t < 4 ? rsrc_a : rsrc_w, (__attribute__((address_space(3))) void*)dst, 16,
                                                         voff[t], kt * (BK * 4), 0, 0);
#endif
            } else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + kt * BK),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x16 acc[NSUB];
    f32x16 accx[F16X3 ? NSUB : 1];     // fp16x3: the cross terms hi*lo + lo*hi (scaled by 2^11)
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        // EPI: the accumulators start from the bias (all 16 elements of acc[j] belong to column n0 + 32 j + lane % 32);
        // a K slice starts from zero, its bias is added by the fix-up kernel
        const float b0 = (EPI && bias && !is_slice) ? bias[n0 + j * 32 + (lane & 31)] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = b0;
    }
#pragma unroll
    for (int j = 0; j < (F16X3 ? NSUB : 1); ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accx[j][e] = 0.f;

    // fragment read: row (l & 31) of a 32-row block, logical 16-byte column 2*kk + (l >> 5)
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    const int fhalf = lane >> 5;
    const int nk_all = K / BK;
    const int kt0 = is_slice ? (int)((int64_t)slice * nk_all / g.split) : 0;
    const int nk = is_slice ? (int)((int64_t)(slice + 1) * nk_all / g.split) : nk_all;
    issue(kt0, kt0 & 1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_assume(kt0 < nk);       // (K >= 32: without this the accumulators are initialised twice, 160 moves)
    for (int kt = kt0; kt < nk; ++kt) {
        // tile kt has landed (vmcnt(0) is part of the barrier while LDS-DMA is in flight) and every
        // wave is done reading the other buffer (it finished step kt-1 before arriving here)
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const float* As = lds + (kt & 1) * (ROWS * 32) + (wm * 32 + frow) * 32;
        const float* Bs = lds + (kt & 1) * (ROWS * 32) + (BM + frow) * 32;
        if (F16X3) {
            // operands are hi/lo fp16 splits: a 128-byte row of this K block is [32 hi | 32 lo]; one MFMA takes 16 k,
            // lanes < 32 the first 8 and lanes >= 32 the next 8: 16-byte slot 2g + half (hi), 4 + 2g + half (lo).
            // a.w = sum hi*hi + 2^-11 sum (hi*lo + lo*hi), fp32 accumulation; lo*lo (2^-22 relative) is dropped.
            // Flattened (gk, j) stages q = NSUB gk + j: stage q issues the B fragments of stage q + 2 (and the A
            // fragments of the next gk) before its three MFMAs -- an MFMA is only 32 cycles here, and with two
            // waves per SIMD every LDS read needs ~200 cycles of issued matrix work between issue and use.
            f16x8 ahi[2], alo[2], bhi[3], blo[3];
            auto slh = [&](int gk) { return ((2 * gk + fhalf) ^ fsw) * 4; };
            auto sll = [&](int gk) { return ((4 + 2 * gk + fhalf) ^ fsw) * 4; };
            auto read_b = [&](int q, int slot) {
                const int gk = q / NSUB, j = q % NSUB;
                bhi[slot] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * 32 + slh(gk));
                blo[slot] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * 32 + sll(gk));
            };
            ahi[0] = *reinterpret_cast<const f16x8*>(As + slh(0));
            alo[0] = *reinterpret_cast<const f16x8*>(As + sll(0));
            read_b(0, 0);
            read_b(1, 1);
#pragma unroll
            for (int q = 0; q < 2 * NSUB; ++q) {
                const int gk = q / NSUB, j = q % NSUB;
                if (q + 2 < 2 * NSUB) read_b(q + 2, (q + 2) % 3);
                if (q == NSUB - 2) {
                    ahi[1] = *reinterpret_cast<const f16x8*>(As + slh(1));
                    alo[1] = *reinterpret_cast<const f16x8*>(As + sll(1));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[gk], bhi[q % 3], acc[j], 0, 0, 0);
                accx[F16X3 ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[gk], blo[q % 3], accx[F16X3 ? j : 0], 0, 0, 0);
                accx[F16X3 ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[gk], bhi[q % 3], accx[F16X3 ? j : 0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
        f32x4 fa[2], fb[2][NSUB];
        fa[0] = *reinterpret_cast<const f32x4*>(As + ((fhalf) ^ fsw) * 4);
#pragma unroll
        for (int j = 0; j < NSUB; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * 32 + ((fhalf) ^ fsw) * 4);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 8) {
                const int sl = ((2 * (kk + 1) + fhalf) ^ fsw) * 4;
                fa[nxt] = *reinterpret_cast<const f32x4*>(As + sl);
#pragma unroll
                for (int j = 0; j < NSUB; ++j) fb[nxt][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * 32 + sl);
            }
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur][j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur][j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur][j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur][j].w, acc[j], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // all waves done with the operand buffers before they become the store stage
    __builtin_amdgcn_s_setprio(3);
    if (F16X3) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 s11 = {1.0f / 2048.0f, 1.0f / 2048.0f};
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 x2 = {accx[F16X3 ? j : 0][e], accx[F16X3 ? j : 0][e + 1]}, a2 = {acc[j][e], acc[j][e + 1]};
                const f32x2 o2 = __builtin_elementwise_fma(x2, s11, a2);
                acc[j][e] = o2[0];
                acc[j][e + 1] = o2[1];
            }
    }
    if (is_slice) {
        // raw accumulators of this K slice -> scratch tile [(tile, slice)][128][BN]; the fix-up kernel
        // adds the slices in a fixed order and applies bias / activation / residual
        GemmArgs gp = g;
        gp.M = m0 + BM;
        gp.N = n0 + BN;
        gp.ldy = BN;
        gp.out_split = 0;            // partial sums stay fp32; the fix-up kernel writes the split form
        float* tile_ws = g.splitk_ws + ((size_t)(logical - g.tile_base) * g.split + slice) * (BM * BN);
        gemm_epilogue<0, NSUB>(gp, acc, lds, tile_ws - (m0 * BN + n0), nullptr, nullptr, m0, n0, lane, w, wm, wn);
    } else {
        if constexpr (EPI != 0) gemm_epilogue_split<MODE, NSUB>(g, acc, lds, Y, res, m0, n0, lane, w);
        else gemm_epilogue<MODE, NSUB>(g, acc, lds, Y, bias, res, m0, n0, lane, w, wm, wn);
    }
}

// ---------------------------------------------------------------------------------------------
// register-staged kernels (K tails, and the small tile)
// ---------------------------------------------------------------------------------------------
template <int MODE, int WAVES_M, int NSUB, int BKT>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const GemmArgs g) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int BM = 32 * WAVES_M;
    constexpr int BN = 32 * NSUB * WAVES_N;
    constexpr int LDS_LD = BKT + 4;            // padded pitch: conflict-free ds_read_b128 of 16 rows x 4 floats
    constexpr int C4 = BKT / 4;                // 16-byte columns per K slab
    constexpr int RPP = 256 / C4;              // rows covered by one pass of the 256 threads
    constexpr int A_LOADS = BM / RPP;
    constexpr int B_LOADS = BN / RPP;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_LD];
    float* As = lds;
    float* Bs = lds + BM * LDS_LD;

    const unsigned logical = logical_tile();
    const int64_t m0 = (int64_t)(logical / (unsigned)g.tiles_n) * BM;
    const int n0 = (int)(logical % (unsigned)g.tiles_n) * BN;
    const int64_t M = g.M;
    const int N = g.N, K = g.K;

    // batch z = z1 * nb2 + z2 with independent strides (e.g. z1 = batch item, z2 = attention head)
    const int z1 = (int)blockIdx.y / g.nb2, z2 = (int)blockIdx.y % g.nb2;
    const float* A = g.A + z1 * g.a_s1 + z2 * g.a_s2;
    const float* W = g.W + z1 * g.w_s1 + z2 * g.w_s2;
    float* Y = g.Y + z1 * g.y_s1 + z2 * g.y_s2;
    const float* bias = g.bias ? g.bias + z2 * g.bias_s2 : nullptr;
    const float* res = g.res ? g.res + z1 * g.r_s1 + z2 * g.r_s2 : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = wave_id();
    const int wm = w / WAVES_N, wn = w % WAVES_N;
    const int lrow = tid / C4;  // row inside a pass
    const int lc4 = tid % C4;   // 16-byte column inside the K slab

    // tile-relative offsets (rows past the end are clamped; their results are never stored)
    const float* At = A + m0 * g.lda + lc4 * 4;
    const float* Wt = W + (int64_t)n0 * g.ldw + lc4 * 4;
    const int a_rows = (int)((M - m0) < BM ? (M - m0) : BM) - 1;
    const int b_rows = ((N - n0) < BN ? (N - n0) : BN) - 1;
    int64_t a_off[A_LOADS], b_off[B_LOADS];
#pragma unroll
    for (int p = 0; p < A_LOADS; ++p) a_off[p] = min(lrow + RPP * p, a_rows) * g.lda;
#pragma unroll
    for (int p = 0; p < B_LOADS; ++p) b_off[p] = min(lrow + RPP * p, b_rows) * g.ldw;

    f32x16 acc[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // K tail (K % 32 != 0): out-of-range 16-byte columns are replaced by zeros when the staged
    // registers are written to LDS (not at load time: a select right behind the loads would make
    // every K step wait for its own global loads).  The loads themselves always use a clamped,
    // valid offset.  K % 4 == 0 is required by the caller.
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[A_LOADS], rb[B_LOADS];
    bool in_cur = lc4 * 4 < K;
    {
        const int ko = in_cur ? 0 : -lc4 * 4;
#pragma unroll
        for (int p = 0; p < A_LOADS; ++p) ra[p] = *reinterpret_cast<const f32x4*>(At + ko + a_off[p]);
#pragma unroll
        for (int p = 0; p < B_LOADS; ++p) rb[p] = *reinterpret_cast<const f32x4*>(Wt + ko + b_off[p]);
    }

    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
    const int nk = (K + BKT - 1) / BKT;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int p = 0; p < A_LOADS; ++p)
            *reinterpret_cast<f32x4*>(&As[(lrow + RPP * p) * LDS_LD + lc4 * 4]) = in_cur ? ra[p] : zero4;
#pragma unroll
        for (int p = 0; p < B_LOADS; ++p)
            *reinterpret_cast<f32x4*>(&Bs[(lrow + RPP * p) * LDS_LD + lc4 * 4]) = in_cur ? rb[p] : zero4;
        __syncthreads();
        if (kt + 1 < nk) {
            in_cur = (kt + 1) * BKT + lc4 * 4 < K;
            const int ko = in_cur ? (kt + 1) * BKT : -lc4 * 4;
#pragma unroll
            for (int p = 0; p < A_LOADS; ++p) ra[p] = *reinterpret_cast<const f32x4*>(At + ko + a_off[p]);
#pragma unroll
            for (int p = 0; p < B_LOADS; ++p) rb[p] = *reinterpret_cast<const f32x4*>(Wt + ko + b_off[p]);
        }
        // Fragments of k-slice kk+1 are fetched from LDS (into a second register set) before the
        // MFMAs of slice kk issue.
        f32x4 fa[2], fb[2][NSUB];
        fa[0] = *reinterpret_cast<const f32x4*>(&As[wm * 32 * LDS_LD + frag_off]);
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
            fb[0][j] = *reinterpret_cast<const f32x4*>(&Bs[(wn * NSUB + j) * 32 * LDS_LD + frag_off]);
#pragma unroll
        for (int kk = 0; kk < BKT / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BKT / 8) {
                fa[nxt] = *reinterpret_cast<const f32x4*>(&As[wm * 32 * LDS_LD + frag_off + (kk + 1) * 8]);
#pragma unroll
                for (int j = 0; j < NSUB; ++j)
                    fb[nxt][j] = *reinterpret_cast<const f32x4*>(
                        &Bs[(wn * NSUB + j) * 32 * LDS_LD + frag_off + (kk + 1) * 8]);
            }
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur][j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur][j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur][j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur][j].w, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    gemm_epilogue<MODE, NSUB>(g, acc, lds, Y, bias, res, m0, n0, lane, w, wm, wn);
}

template <int WAVES_M, int NSUB, int BKT>
static void launch_tile(const GemmArgs& g, int mode, dim3 grid, hipStream_t s) {
    switch (mode) {
        case 0: hipLaunchKernelGGL((gemm_nt_f32_kernel<0, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
        case 1: hipLaunchKernelGGL((gemm_nt_f32_kernel<1, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
        case 2: hipLaunchKernelGGL((gemm_nt_f32_kernel<2, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
        case 3: hipLaunchKernelGGL((gemm_nt_f32_kernel<3, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL((gemm_nt_f32_kernel<4, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
    }
}

// ---------------------------------------------------------------------------------------------
// latency kernel for the decoder's short problems (M <= 512): 32 x 32 output tile per workgroup,
// the four waves split K between them (intra-workgroup split-K, reduced through LDS in a fixed
// order -> deterministic), operands stream HBM/L2 -> registers with every load of a wave's K range
// in flight at once.  A decode step is ~60 dependent launches of this size: what matters is the
// length of the dependent chain inside a launch (here: one memory round trip + K/8 MFMAs), and
// how many CUs share the weight stream ((M/32) x (N/32) workgroups instead of (M/32) x (N/128)).
// ---------------------------------------------------------------------------------------------
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_splitk4_kernel(const GemmArgs g) {
    // per wave: a [32 rows x 64 k] slab of each operand (272-byte row pitch: conflict-free for the 16-byte
    // row-major writes and for the fragment reads alike); the partial tiles reuse the front of it
    constexpr int PITCH = 68;                      // floats
    constexpr int SLAB = 32 * PITCH;
    __shared__ __attribute__((aligned(16))) float smem[NW * 2 * SLAB];
    float* part = smem;
    const int tiles_n = g.tiles_n;
    const int64_t m0 = (int64_t)(blockIdx.x / (unsigned)tiles_n) * 32;
    const int n0 = (int)(blockIdx.x % (unsigned)tiles_n) * 32;
    const int64_t M = g.M;
    const int N = g.N, K = g.K;
    const int z1 = (int)blockIdx.y / g.nb2, z2 = (int)blockIdx.y % g.nb2;
    const float* A = g.A + z1 * g.a_s1 + z2 * g.a_s2;
    const float* W = g.W + z1 * g.w_s1 + z2 * g.w_s2;
    float* Y = g.Y + z1 * g.y_s1 + z2 * g.y_s2;
    const float* bias = g.bias ? g.bias + z2 * g.bias_s2 : nullptr;
    const float* res = g.res ? g.res + z1 * g.r_s1 + z2 * g.r_s2 : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int r32 = lane & 31, kh = lane >> 5;
    // Loads: one instruction covers 4 rows x 256 contiguous bytes (lane -> row 4j + (lane >> 4), 16-byte column
    // lane & 15), i.e. 8 cache lines.  (Loading the MFMA fragments directly -- lane -> row lane & 31, 16 bytes --
    // touches 32 lines for 32 useful bytes each, and the CU's line-request rate, not latency, then sets the
    // kernel's duration: 7 us + 6.8 us per 1024 of K, scripts/bench_small_gemm.py.)
    const int lrow = lane >> 4, lcol = (lane & 15) * 4;
    const float* aptr[8];
    const float* wptr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t ar = m0 + 4 * j + lrow;
        const int wr = n0 + 4 * j + lrow;
        aptr[j] = A + (ar < M ? ar : M - 1) * g.lda + lcol;
        wptr[j] = W + (int64_t)(wr < N ? wr : N - 1) * g.ldw + lcol;
    }
    float* As = smem + w * (2 * SLAB);
    float* Ws = As + SLAB;

    const int steps = (K + 7) / 8;                 // one step = 8 consecutive k (two 16-byte halves)
    const int per = (steps + NW - 1) / NW;
    const int s0 = w * per;
    const int s1 = (s0 + per < steps) ? s0 + per : steps;
    const int kend = s1 * 8 < K ? s1 * 8 : K;      // this wave's K range is [8 s0, kend)
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int UN = 8;                          // steps per batch = 64 k = one slab row
    for (int sb = s0; sb < s1; sb += UN) {
        const int k0 = sb * 8 + lcol;
        const bool in = k0 < kend;                 // K % 4 == 0: a 16-byte column is in or out as a whole
        const int kc = in ? k0 - lcol : 0;         // (clamped, always valid address; zero-filled below)
        f32x4 la[8], lw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            la[j] = *reinterpret_cast<const f32x4*>(aptr[j] + kc);
            lw[j] = *reinterpret_cast<const f32x4*>(wptr[j] + kc);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *reinterpret_cast<f32x4*>(As + (4 * j + lrow) * PITCH + lcol) = in ? la[j] : zero4;
            *reinterpret_cast<f32x4*>(Ws + (4 * j + lrow) * PITCH + lcol) = in ? lw[j] : zero4;
        }
        // (each wave reads back only what it wrote: LDS operations of one wave are processed in order)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(As + r32 * PITCH + 8 * u + 4 * kh);
            const f32x4 b = *reinterpret_cast<const f32x4*>(Ws + r32 * PITCH + 8 * u + 4 * kh);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
    __syncthreads();                               // every wave is done with its slabs before they become `part`
    // partial tiles -> LDS as [wave][row][col]; C/D layout: col = lane & 31, row = (e&3)+8*(e>>2)+4*(lane>>5)
#pragma unroll
    for (int e = 0; e < 16; ++e) part[w * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + r32] = acc[e];
    __syncthreads();
    if (tid >= 256) return;                        // the first four waves add the partial tiles and write
    const int row = tid >> 3, c = (tid & 7) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(&part[row * 32 + c]);
#pragma unroll
    for (int q = 1; q < NW; ++q) v += *reinterpret_cast<const f32x4*>(&part[q * 1024 + row * 32 + c]);  // waves 0..NW-1 in order
    const int64_t gr = m0 + row;
    const int gc = n0 + c;
    if (gr >= M || gc >= N) return;
    const float alpha = g.alpha;
    const bool vec = gc + 3 < N && (g.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) &&
                     (MODE != 2 || (g.ldres % 4 == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0));
    if (vec) {
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + gc);
        if (MODE == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (MODE == 2) v = *reinterpret_cast<const f32x4*>(res + gr * g.ldres + gc) + alpha * v;
        if (MODE == 3 && (g.scale_cols == 0 || gc < g.scale_cols)) v = alpha * v;
        *reinterpret_cast<f32x4*>(Y + gr * g.ldy + gc) = v;
    } else {
        for (int q = 0; q < 4 && gc + q < N; ++q) {
            float x = v[q] + (bias ? bias[gc + q] : 0.f);
            if (MODE == 1) x = fmaxf(x, 0.f);
            if (MODE == 2) x = res[gr * g.ldres + gc + q] + alpha * x;
            if (MODE == 3 && (g.scale_cols == 0 || gc + q < g.scale_cols)) x = alpha * x;
            Y[gr * g.ldy + gc + q] = x;
        }
    }
}

static void launch_splitk4(const GemmArgs& g, int mode, dim3 grid, hipStream_t s) {
    // (8 waves splitting K were measured for the decoder's K = 2048 layer: 15.6 vs 16.4 us, not worth a variant)
    switch (mode) {
        case 0: hipLaunchKernelGGL((gemm_splitk4_kernel<0, 4>), grid, dim3(256), 0, s, g); break;
        case 1: hipLaunchKernelGGL((gemm_splitk4_kernel<1, 4>), grid, dim3(256), 0, s, g); break;
        case 2: hipLaunchKernelGGL((gemm_splitk4_kernel<2, 4>), grid, dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL((gemm_splitk4_kernel<3, 4>), grid, dim3(256), 0, s, g); break;
    }
}

// fix-up of the split-K tail: FIX_PARTS workgroups per tail tile (32 rows each), slices added in ascending
// order.  All slice loads of a position are issued before the first add; 16-byte loads and stores.
constexpr int FIX_PARTS = 4;
template <int MODE, int NSUB>
__global__ __launch_bounds__(256) void gemm_splitk_fixup_kernel(const GemmArgs g) {
    constexpr int BM = 128, BN = 32 * NSUB, RPB = BM / FIX_PARTS, MAXS = 8;
    const unsigned tile = blockIdx.x / FIX_PARTS, part = blockIdx.x % FIX_PARTS;
    const unsigned logical = (unsigned)g.tile_base + tile;
    const int64_t m0 = (int64_t)(logical / (unsigned)g.tiles_n) * BM + part * RPB;
    const int n0 = (int)(logical % (unsigned)g.tiles_n) * BN;
    const float* base = g.splitk_ws + (size_t)tile * g.split * (BM * BN) + part * RPB * BN;
    const bool vec_ok = (g.ldy % 4 == 0) && (MODE != 2 || g.ldres % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.Y) & 15) == 0) &&
                        (MODE != 2 || (reinterpret_cast<uintptr_t>(g.res) & 15) == 0) &&
                        (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < RPB * BN / 4; i += 256) {
        const int r = i / (BN / 4), c = (i - r * (BN / 4)) * 4;
        const int64_t row = m0 + r;
        const int col = n0 + c;
        if (row >= g.M || col >= g.N) continue;
        f32x4 p[MAXS];
#pragma unroll
        for (int q = 0; q < MAXS; ++q)
            p[q] = q < g.split ? *reinterpret_cast<const f32x4*>(base + (size_t)q * (BM * BN) + r * BN + c) : zero4;
        const bool full = vec_ok && col + 3 < g.N;
        f32x4 rv = zero4, bv = zero4;
        if (full) {
            if (MODE == 2 && g.res_split) {
                const _Float16* rr = reinterpret_cast<const _Float16*>(g.res + row * g.ldres) + (col >> 5) * 64 + (col & 31);
                const f16x4 h4 = *reinterpret_cast<const f16x4*>(rr), l4 = *reinterpret_cast<const f16x4*>(rr + 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) rv[q] = (float)h4[q] + (float)l4[q] * (1.0f / 2048.0f);
            } else if (MODE == 2)
                rv = *reinterpret_cast<const f32x4*>(g.res + row * g.ldres + col);
            if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
        }
        f32x4 v = p[0];
#pragma unroll
        for (int q = 1; q < MAXS; ++q)
            if (q < g.split) v += p[q];
        if (full) {
            v += bv;
            if (MODE == 1) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (MODE == 2) v = rv + g.alpha * v;
            if (MODE == 3 && (g.scale_cols == 0 || col < g.scale_cols)) v = g.alpha * v;
            if (g.out_split) {
                f16x4 hi, lo;
                note_range(amax4(0.f, v), g.range_flag);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    _Float16 h, l;
                    split_f16x3(v[q], h, l);
                    hi[q] = h;
                    lo[q] = l;
                }
                _Float16* yr = reinterpret_cast<_Float16*>(g.Y + row * g.ldy) + (col >> 5) * 64 + (col & 31);
                *reinterpret_cast<f16x4*>(yr) = hi;
                *reinterpret_cast<f16x4*>(yr + 32) = lo;
                continue;
            }
            *reinterpret_cast<f32x4*>(g.Y + row * g.ldy + col) = v;
            continue;
        }
        for (int e = 0; e < 4 && col + e < g.N; ++e) {
            float x = v[e] + (g.bias ? g.bias[col + e] : 0.f);
            if (MODE == 1) x = fmaxf(x, 0.f);
            if (MODE == 2) x = g.res[row * g.ldres + col + e] + g.alpha * x;
            if (MODE == 3 && (g.scale_cols == 0 || col + e < g.scale_cols)) x = g.alpha * x;
            g.Y[row * g.ldy + col + e] = x;
        }
    }
}

template <int NSUB>
static void launch_fixup(const GemmArgs& g, int mode, dim3 grid, hipStream_t s) {
    switch (mode) {
        case 0: hipLaunchKernelGGL((gemm_splitk_fixup_kernel<0, NSUB>), grid, dim3(256), 0, s, g); break;
        case 1: hipLaunchKernelGGL((gemm_splitk_fixup_kernel<1, NSUB>), grid, dim3(256), 0, s, g); break;
        case 2: hipLaunchKernelGGL((gemm_splitk_fixup_kernel<2, NSUB>), grid, dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL((gemm_splitk_fixup_kernel<3, NSUB>), grid, dim3(256), 0, s, g); break;
    }
}

static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

constexpr int SPLITK_MAX_SLICES = 1024;  // tail tiles x slices (84 MB of scratch)
size_t gemm_splitk_ws_bytes() { return (size_t)SPLITK_MAX_SLICES * 128 * 160 * sizeof(float); }

template <int MODE, int NSUB, bool SPLITK>
static void launch_glds_stage(const GemmArgs& g, dim3 grid, hipStream_t s) {
    // the buffer form of the operand loads (default) needs the tile's lane offsets to fit 32 bits
    const int want = opt(OPT_GEMM_GLOBAL_LOADS) ? 0 : 2;
    if (g.f16x3) {
        if constexpr (MODE == 1 || MODE == 2) {
            // the TDS block's layers: split-form output under the range guard (MODE 2: split-form residual too)
            const bool y_ok = g.ldy % 4 == 0 && g.ldy < (1 << 21) && (reinterpret_cast<uintptr_t>(g.Y) & 15) == 0 && g.N % (32 * NSUB) == 0;
            const bool r_ok = MODE != 2 || (g.res_split && g.ldres % 4 == 0 && g.ldres < (1 << 21) && (reinterpret_cast<uintptr_t>(g.res) & 15) == 0);
            if (g.out_split && g.range_flag && y_ok && r_ok && grid.y == 1) {
                hipLaunchKernelGGL((gemm_glds_kernel<MODE, NSUB, SPLITK, 2, true, 1>), grid, dim3(256), 0, s, g);
                return;
            }
        }
        if constexpr (MODE <= 3) hipLaunchKernelGGL((gemm_glds_kernel<MODE, NSUB, SPLITK, 2, true>), grid, dim3(256), 0, s, g);
        return;
    }
    if (want == 2 && g.lda < (1 << 21) && g.ldw < (1 << 21))
        hipLaunchKernelGGL((gemm_glds_kernel<MODE, NSUB, SPLITK, 2>), grid, dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_glds_kernel<MODE, NSUB, SPLITK, 0>), grid, dim3(256), 0, s, g);
}

// splitk: the grid is [whole-round tiles | K slices of the remaining tiles] (modes 0..3 only)
template <int NSUB>
static void launch_glds(const GemmArgs& g, int mode, bool splitk, dim3 grid, hipStream_t s) {
    if (splitk) {
        switch (mode) {
            case 0: launch_glds_stage<0, NSUB, true>(g, grid, s); break;
            case 1: launch_glds_stage<1, NSUB, true>(g, grid, s); break;
            case 2: launch_glds_stage<2, NSUB, true>(g, grid, s); break;
            default: launch_glds_stage<3, NSUB, true>(g, grid, s); break;
        }
        return;
    }
    switch (mode) {
        case 0: launch_glds_stage<0, NSUB, false>(g, grid, s); break;
        case 1: launch_glds_stage<1, NSUB, false>(g, grid, s); break;
        case 2: launch_glds_stage<2, NSUB, false>(g, grid, s); break;
        case 3: launch_glds_stage<3, NSUB, false>(g, grid, s); break;
        default: launch_glds_stage<4, NSUB, false>(g, grid, s); break;
    }
}

int launch_gemm(GemmArgs g, int mode, int nbatch, hipStream_t s) {
    TAL_CHECK_ARG(g.A && g.W && (g.Y || mode == 4), "gemm: null pointer");
    TAL_CHECK_ARG(mode != 4 || (g.part_val && g.part_idx && g.part_ld >= gemm_mode4_partials(g.M, g.N)),
                  "gemm: mode 4 needs partial buffers with part_ld >= %d", gemm_mode4_partials(g.M, g.N));
    TAL_CHECK_ARG(g.M >= 0 && g.N > 0 && g.K > 0, "gemm: bad shape M=%lld N=%d K=%d", (long long)g.M, g.N, g.K);
    TAL_CHECK_ARG(g.K % 4 == 0 && g.lda % 4 == 0 && g.ldw % 4 == 0, "gemm: K=%d lda=%lld ldw=%lld must be multiples of 4", g.K, (long long)g.lda, (long long)g.ldw);
    TAL_CHECK_ARG(mode >= 0 && mode <= 4, "gemm: mode %d", mode);
    TAL_CHECK_ARG(mode != 2 || g.res, "gemm: mode 2 needs a residual");
    TAL_CHECK_ARG(g.scale_cols % 4 == 0, "gemm: scale_cols must be a multiple of 4");
    TAL_CHECK_ARG(nbatch >= 1 && nbatch <= 65535 && g.nb2 >= 1, "gemm: batch %d", nbatch);
    if (g.M == 0) return TAL_OK;
    if (g.f16x3 || g.out_split) {
        TAL_CHECK_ARG(g.f16x3, "gemm: out_split needs the fp16x3 form");
        TAL_CHECK_ARG(g.M > 128 && g.K % BK == 0 && mode <= 3 && nbatch == 1, "gemm: the fp16x3 form needs M > 128, K %% 32 == 0, modes 0-3");
        TAL_CHECK_ARG(!g.out_split || (g.N % 160 == 0 && g.ldy % 4 == 0), "gemm: split output needs N %% 160 == 0");
        TAL_CHECK_ARG(g.lda < (1 << 21) && g.ldw < (1 << 21), "gemm: leading dimension too large for the fp16x3 form");
    }
    TAL_CHECK_ARG(!g.res_split || (g.f16x3 && mode == 2 && g.N % 160 == 0 && g.ldres % 32 == 0 && g.ldy < (1 << 21) && g.ldres < (1 << 21) &&
                                  (reinterpret_cast<uintptr_t>(g.res) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.Y) & 15) == 0),
                  "gemm: a split-form residual needs the fp16x3 form, mode 2, N %% 160 == 0");
    const bool small = g.M <= 512 && !g.f16x3;
    const int bm = small ? 32 : 128, bn = small ? 128 : 160;
    g.tiles_n = (int)cdiv(g.N, bn);
    g.tiles_n_magic = g.tiles_n > 1 ? (unsigned)((1ull << 32) / (unsigned)g.tiles_n) + 1u : 0u;
    const int64_t nb = cdiv(g.M, bm) * g.tiles_n;
    TAL_CHECK_ARG(nb < (1ll << 31) && nb * g.tiles_n < (1ll << 32), "gemm: grid too large");
    dim3 grid((unsigned)nb, (unsigned)nbatch);
    ProfScope prof(PROF_GEMM, 2.0 * (double)g.M * (double)g.N * (double)g.K * nbatch, s);
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15) == 0;
    // measurement switches (tal_set_option; three relaxed atomic loads per launch)
    const bool no_splitk4 = opt(OPT_GEMM_NO_SPLITK4) != 0, no_glds = opt(OPT_GEMM_NO_GLDS) != 0, no_tail = opt(OPT_GEMM_NO_SPLITK_TAIL) != 0;
    // small problems are latency-bound (a K step costs one L2 round trip, not its 16 MFMAs): a
    // 128-deep K slab quarters the number of dependent round trips
    if (small && mode != 4 && aligned16 && !no_splitk4) {
        GemmArgs h = g;
        h.tiles_n = (int)cdiv(g.N, 32);
        const int64_t nb2 = cdiv(g.M, 32) * h.tiles_n;
        TAL_CHECK_ARG(nb2 < (1ll << 31), "gemm: grid too large");
        launch_splitk4(h, mode, dim3((unsigned)nb2, (unsigned)nbatch), s);
    } else if (small && g.K >= 256)
        launch_tile<1, 1, 128>(g, mode, grid, s);
    else if (small)
        launch_tile<1, 1, 32>(g, mode, grid, s);
    else if (g.K % BK == 0 && aligned16 && !no_glds) {
        // (a 128 x 96 tile -- 10.3 instead of 6.2 rounds on the 1-hour stage-3 shape -- was measured at
        //  +2 %, inside run-to-run noise: the 160-wide tile stays)
        // Stream-K-lite: 2 workgroups per CU are resident, so a launch proceeds in rounds of 2*CUs
        // tiles; the tiles of the last, partial round are cut along K into `split` slices each, so that
        // they occupy the whole chip for a fraction of a round instead of part of it for a whole one
        // (1-hour stage-3 shape: 3168 tiles = 6.19 rounds).  The slices are blocks of the same launch,
        // dispatched behind the whole tiles; a fix-up kernel adds them in a fixed order.  `split` is
        // the one that minimises the tail, ceil(rem*split/slots)/split rounds, within the scratch size.
        const int64_t slots = 2 * (int64_t)device_cus();
        const int64_t rem = nb % slots, full = nb - rem;
        const int nk = g.K / BK;
        int split = 0;
        // (a launch with less than a quarter of a round of tiles -- a few hundred rows -- is cut along K as a whole: its
        //  few tiles become up to 8x as many K slices, so that most of the chip takes part)
        if (rem > 0 && (full > 0 || 4 * rem <= slots) && nbatch == 1 && mode <= 3 && g.splitk_ws && !no_tail) {
            // cost of a candidate in rounds: the slices' own rounds + their scratch traffic (each slice
            // writes and the fix-up re-reads a 128x160 fp32 tile, ~4 TB/s) relative to one round's duration
            // (measured: 4.45 us per K step + ~14 us per tile, scripts/bench_gemm_fit.py)
            // (fp16x3 form: 1.56 us per K step + ~10.5 us per tile, scripts/ubench/gemm_f16x3.hip)
            const double round_us = g.f16x3 ? 1.56 * nk + 10.5 : 4.45 * nk + 14.0;
            double best = 0.97;
            for (int sp = 2; sp <= 8 && sp <= nk / 4; ++sp) {
                if ((size_t)rem * sp * 128 * 160 * sizeof(float) > g.splitk_ws_bytes) break;
                const double traffic_us = (double)rem * sp * (2.0 * 128 * 160 * 4) / 4.0e6;
                const double tail = (double)cdiv(rem * sp, slots) / sp + traffic_us / round_us;
                if (tail < best) { best = tail; split = sp; }
            }
        }
        if (split < 2) {
            launch_glds<5>(g, mode, false, grid, s);
        } else {
            GemmArgs h = g;
            h.tile_base = (int)full;
            h.split = split;
            h.tail_tiles = (int)rem;
            launch_glds<5>(h, mode, true, dim3((unsigned)(full + rem * split)), s);   // whole tiles, then K slices -> scratch
            launch_fixup<5>(h, mode, dim3((unsigned)rem * FIX_PARTS), s);                         // ordered sum + epilogue
        }
    }
    else
        launch_tile<4, 5, 32>(g, mode, grid, s);
    TAL_CHECK_LAUNCH("gemm");
    return TAL_OK;
}

int launch_linear_ws(const float* x, const float* w, const float* b, const float* res, float alpha, int mode, int64_t M,
                     int N, int K, float* y, float* ws, size_t ws_bytes, hipStream_t s) {
    TAL_CHECK_ARG(x && w && y, "tal_linear_fwd: null pointer");
    GemmArgs g = {};
    g.A = x; g.W = w; g.bias = b; g.res = res; g.Y = y;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldy = N; g.ldres = N;
    g.nb2 = 1;
    g.alpha = alpha;
    g.splitk_ws = ws;
    g.splitk_ws_bytes = ws_bytes;
    return launch_gemm(g, mode, 1, s);
}

// fp32 -> hi / lo fp16 split in fp32-row geometry: per row and 32-wide K block, 32 hi halves then 32 lo halves
// (lo = (x - hi) * 2^11): one thread per 4 consecutive floats
__global__ __launch_bounds__(256) void split_f16x3_kernel(const float* __restrict__ x, _Float16* __restrict__ out, int64_t n4,
                                                         int* __restrict__ range_flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    note_range(amax4(0.f, v), range_flag);
    f16x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        _Float16 h, l;
        split_f16x3(v[q], h, l);
        hi[q] = h;
        lo[q] = l;
    }
    // element index e = 4 i inside a row-major [rows, K] array with K % 32 == 0: block e / 32 (64 halves), slot e % 32
    const int64_t e = 4 * i;
    _Float16* o = out + (e >> 5) * 64 + (e & 31);
    *reinterpret_cast<f16x4*>(o) = hi;
    *reinterpret_cast<f16x4*>(o + 32) = lo;
}

int launch_split_f16x3(const float* x, void* out, int64_t rows, int K, hipStream_t s, int* range_flag) {
    TAL_CHECK_ARG(x && out && rows >= 0 && K > 0 && K % 32 == 0, "split_f16x3: bad argument (K %% 32 == 0 required)");
    TAL_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "split_f16x3: 16-byte alignment required");
    const int64_t n4 = rows * K / 4;
    if (n4 == 0) return TAL_OK;
    ProfScope prof(PROF_OTHER, (double)rows * K * 8.0, s);
    hipLaunchKernelGGL(split_f16x3_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, s, x, reinterpret_cast<_Float16*>(out), n4, range_flag);
    TAL_CHECK_LAUNCH("split_f16x3");
    return TAL_OK;
}

int launch_linear_f16x3(const void* xs, const void* wsplit, const float* b, const float* res, float alpha, int mode, int64_t M,
                        int N, int K, void* y, int out_split, float* ws, size_t ws_bytes, hipStream_t s, int* range_flag, int res_split) {
    TAL_CHECK_ARG(xs && wsplit && y, "linear_f16x3: null pointer");
    GemmArgs g = {};
    g.A = reinterpret_cast<const float*>(xs); g.W = reinterpret_cast<const float*>(wsplit); g.bias = b; g.res = res;
    g.Y = reinterpret_cast<float*>(y);
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldy = N; g.ldres = N;
    g.nb2 = 1;
    g.alpha = alpha;
    g.splitk_ws = ws;
    g.splitk_ws_bytes = ws_bytes;
    g.f16x3 = 1;
    g.out_split = out_split;
    g.range_flag = range_flag;
    g.res_split = res_split;
    return launch_gemm(g, mode, 1, s);
}

// number of (value, index) partials per row a mode-4 launch of this shape produces
int gemm_mode4_partials(int64_t M, int N) { return M <= 512 ? (int)cdiv(N, 128) * 4 : (int)cdiv(N, 160); }

int launch_linear(const float* x, const float* w, const float* b, const float* res, float alpha, int mode, int64_t M,
                  int N, int K, float* y, hipStream_t s) {
    TAL_CHECK_ARG(x && w && y, "tal_linear_fwd: null pointer");
    GemmArgs g = {};
    g.A = x; g.W = w; g.bias = b; g.res = res; g.Y = y;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldy = N; g.ldres = N;
    g.nb2 = 1;
    g.alpha = alpha;
    return launch_gemm(g, mode, 1, s);
}

}  // namespace tal
