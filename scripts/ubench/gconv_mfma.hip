// Experiment (not product code): the TDSBlock grouped conv (k = 21, groups = 80, C/G = 10 / 14 / 18, "same" padding,
// + bias + ReLU + ReZero residual; tal/asr/models.py:304-308,329) on the fp16 matrix cores in the fp16x3 form.
//
// Per group the conv is a GEMM  out[co, t] = sum_k W[co, k] * X[k, t],  k = tap * CG + ci,  X[k, t] = x[t + tap - 10][ci].
// With the group's input slab stored TIME-major and compact in LDS ([t][CG] halves), X[., t] is simply the 21*CG
// consecutive halves starting at slab[t * CG] (a Hankel matrix): an MFMA operand fragment (8 consecutive k of one
// column) is one 16-byte LDS read, no im2col.  M = output channels (weights, register-resident, zero rows past CG),
// N = 16 time steps, K = 21*CG rounded up to 32 (zero weight columns; they multiply finite slab bytes).
// v_mfma_f32_16x16x32_f16, three per product block (hi*hi, hi*lo, lo*hi), fp32 accumulation.
// The C layout gives a lane 4 consecutive channels of one time step: one 16-byte load (residual) and store per block.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/gconv_mfma.hip -o scripts/ubench/gconv_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8u __attribute__((ext_vector_type(8), aligned(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int KS = 21, PAD = 10;

// hi = fp16(x), lo = fp16((x - hi) * 2^11); x clamped to the finite fp16 range first (|lo| <= 2^15 then needs no clamp)
__device__ __forceinline__ void split_f16x3(float x, _Float16& hi, _Float16& lo) {
    const float xc = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    hi = (_Float16)xc;
    lo = (_Float16)((xc - (float)hi) * 2048.f);
}

// P = LDS row pitch in halves (>= CG, a multiple of 4): rows are 8- or 16-byte aligned, so a fragment is one aligned
// ds_read_b128 (P % 8 == 0) or two ds_read_b64 (the compact pitch P = CG needs two ds_read2_b32 at half the LDS rate).
// k = tap * P + ci; the P - CG pad channels of a row hold zeros and meet zero weights.
template <int P>
__device__ __forceinline__ h8 lds_frag(const _Float16* p, const _Float16* p_plus4) {
    if (P % 8 == 0) return *reinterpret_cast<const h8*>(p);
    if (P % 4 != 0) return *reinterpret_cast<const h8u*>(p);       // 4-byte aligned: two ds_read2_b32
    // two ds_read_b64 (2 LDS cycles each); the second pointer is opaque so they are not fused into a ds_read2_b64 (8)
    const h4 a = *reinterpret_cast<const h4*>(p), b = *reinterpret_cast<const h4*>(p_plus4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// TAIL (CG > 16): the channels past the first 16 do not get an M tile of their own (2 of 16 rows used); instead one
// M tile holds (channel, time shift) rows -- CT = CG - 16 channels x 16 / CT shifts -- against columns that are 16 / CT
// time steps apart: row (c, s) carries channel c's weights moved s taps down the K axis, so one tile yields CT
// channels x 128 time steps.  K grows by (16 / CT - 1) taps; MFMAs per group and 128 steps: 8 x 16 x 3 + 21 x 3 = 447
// instead of 768.
template <int CG, int P, int GB, int TT, int WGPC, bool TAIL = false>
__global__ __launch_bounds__(256, WGPC) void gconv_mfma_kernel(const float* __restrict__ x, const _Float16* __restrict__ wfrag, const _Float16* __restrict__ wtail,
                                                           const float* __restrict__ bias, float alpha, float* __restrict__ y,
                                                           int64_t T, int C, long long* clk) {
    constexpr int KTOT = (KS - 1) * P + CG, NKS = (KTOT + 31) / 32, MT = TAIL ? 1 : (CG + 15) / 16, MTW = (CG + 15) / 16;
    constexpr int TIN = TT + 2 * PAD;
    constexpr int SLAB = (TIN * P + (32 * NKS > KS * P ? 32 * NKS - KS * P : 0) + 7) & ~7;
    constexpr int CH = GB * CG, CH4 = CH / 4;
    constexpr int RPP = 256 / CH4;                    // time steps per pass of the slab load
    static_assert(CH % 4 == 0 && CG % 2 == 0 && TT % 64 == 0 && P % 2 == 0 && P >= CG, "shape");
    extern __shared__ __attribute__((aligned(16))) _Float16 slab[];   // [2 (hi, lo)][GB][SLAB]
    _Float16* s_hi = slab;
    _Float16* s_lo = slab + GB * SLAB;

    const int b = blockIdx.z, g0 = blockIdx.y * GB;
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* xb = x + (int64_t)b * T * C;
    float* yb = y + (int64_t)b * T * C;

    const long long c_start = clock64(), w_start = wall_clock64();
#ifndef NOPRIO
    __builtin_amdgcn_s_setprio(3);     // the slab phase is VALU work: at equal priority every instruction queues behind the partner workgroup's MFMAs
#endif
    // the first unit's weights are requested before the slab (their latency hides behind the slab load and split)
    constexpr int NU = GB * MT, NB = TT / 16;
    constexpr int UPW = NU >= 4 ? NU / 4 : 1;         // units per wave
    constexpr int WPU = NU >= 4 ? 1 : 4 / NU;         // waves per unit
    constexpr int NBW = NB / WPU;
    static_assert(NU == 1 || NU == 2 || NU % 4 == 0, "units");
    h8 wh[NKS], wl[NKS];
    auto load_w = [&](int u) {
        const h8* wf = reinterpret_cast<const h8*>(wfrag) + (int64_t)((g0 + u / MT) * MTW + u % MT) * (NKS * 2 * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            wh[ks] = wf[(ks * 2 + 0) * 64];
            wl[ks] = wf[(ks * 2 + 1) * 64];
        }
    };
    load_w(NU >= 4 ? w : w / WPU);
    // ---- slab load: thread = (time step inside a pass, 16-byte column piece): 16-byte global loads (GB*CG contiguous
    // floats per time step), split, 4-byte LDS stores at per-thread constant offsets ----
    {
        const int r0 = tid / CH4, c4 = tid - r0 * CH4;
        const bool active = r0 < RPP;
        int so[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ch = c4 * 4 + 2 * q, gl = ch / CG;
            so[q] = gl * SLAB + (ch - gl * CG);
        }
        const float* xc = xb + g0 * CG + c4 * 4;
        constexpr int NPASS = (TIN + RPP - 1) / RPP, UNR = NPASS > 20 ? (NPASS + 1) / 2 : NPASS;   // one memory round trip
        for (int p0 = 0; p0 < NPASS; p0 += UNR) {
            f32x4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ti = r0 + (p0 + u) * RPP;
                const int64_t t = t0 - PAD + ti;
                const bool in = t >= 0 && t < T;
                const int64_t tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
                v[u] = *reinterpret_cast<const f32x4*>(xc + tc * C);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ti = r0 + (p0 + u) * RPP;
                const int64_t t = t0 - PAD + ti;
                const bool in = t >= 0 && t < T;
                if (active && ti < TIN) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        _Float16 h0, l0, h1, l1;
                        split_f16x3(in ? v[u][2 * q] : 0.f, h0, l0);
                        split_f16x3(in ? v[u][2 * q + 1] : 0.f, h1, l1);
                        const h2 hh = {h0, h1}, ll = {l0, l1};
                        *reinterpret_cast<h2*>(s_hi + so[q] + ti * P) = hh;
                        *reinterpret_cast<h2*>(s_lo + so[q] + ti * P) = ll;
                    }
                }
            }
        }
        // zeros in the pad channels of every row and behind the last row (finite bytes under zero weights)
        const h2 z2 = {(_Float16)0.f, (_Float16)0.f};
        if (P > CG)
            for (int i = tid; i < 2 * GB * TIN; i += 256)
#pragma unroll
                for (int c = CG; c < P; c += 2) *reinterpret_cast<h2*>(slab + (i / TIN) * SLAB + (i % TIN) * P + c) = z2;
        for (int i = tid; i < GB * (SLAB - TIN * P); i += 256) {   // (2 GB arrays, two halves per store)
            const int a = i / ((SLAB - TIN * P) / 2), r = i - a * ((SLAB - TIN * P) / 2);
            *reinterpret_cast<h2*>(slab + a * SLAB + TIN * P + 2 * r) = z2;
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    const long long c_load = clock64();

    // ---- units = (group, 16-channel M tile); a wave owns one unit (or a time slice of one) at a time ----
    const int col = lane & 15, kg = lane >> 4;
    for (int uu = 0; uu < UPW; ++uu) {
        const int u = NU >= 4 ? w + 4 * uu : w / WPU;
        const int part = NU >= 4 ? 0 : w % WPU;
        const int gl = u / MT, mt = u - gl * MT, g = g0 + gl;
        if (uu > 0) load_w(u);
        const int ch0 = mt * 16 + 4 * kg;
        const int nvalid = CG - ch0;                  // >= 4: four channels, 2: two, <= 0: none
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nvalid) bv[i] = bias[g * CG + ch0 + i];
        const _Float16* hs = s_hi + gl * SLAB + (part * NBW * 16 + col) * P + 8 * kg;
        const _Float16* ls = s_lo + gl * SLAB + (part * NBW * 16 + col) * P + 8 * kg;
        const _Float16* hs4 = hs + 4;
        const _Float16* ls4 = ls + 4;
        asm volatile("" : "+v"(hs4), "+v"(ls4));
        int boff = 0;
        h8 c0h = lds_frag<P>(hs, hs4), c0l = lds_frag<P>(ls, ls4);
        h8 c1h = lds_frag<P>(hs + 32, hs4 + 32), c1l = lds_frag<P>(ls + 32, ls4 + 32);
        // residual x of block tb is fetched XD blocks ahead (an L2 round trip is longer than one block of MFMAs);
        // the block loop is unrolled by XD + 1 so the ring of in-flight registers is indexed statically
        constexpr int XD = 3;
        static_assert(NBW % (XD + 1) == 0, "blocks per wave");
        const int tbeg = part * NBW, tend = (part + 1) * NBW;
        const float* xcol = xb + g * CG + ch0;
        // branch-free: every lane loads 16 bytes; a lane with two valid channels loads from two floats earlier and keeps
        // the upper half, a lane with none loads the group's first channels (all addresses stay inside the row)
        const int xshift = nvalid >= 4 ? 0 : (nvalid == 2 ? -2 : -ch0);
        auto load_x = [&](int tb) {
            int64_t t = t0 + tb * 16 + col;
            t = t < T ? t : T - 1;
            return *reinterpret_cast<const f32x4u*>(xcol + t * C + xshift);       // (selected at the point of use)
        };
        f32x4 xr[XD + 1];
#pragma unroll
        for (int j = 0; j < XD; ++j) xr[j] = load_x(tbeg + j);
        for (int tb0 = tbeg; tb0 < tend; tb0 += XD + 1) {
#pragma unroll
            for (int j = 0; j <= XD; ++j) {
                const int tb = tb0 + j;
                xr[(j + XD) % (XD + 1)] = load_x(tb + XD < tend ? tb + XD : tb);
                f32x4 acc = bv, ax1 = {0.f, 0.f, 0.f, 0.f}, ax2 = {0.f, 0.f, 0.f, 0.f};
                const int nboff = tb + 1 < tend ? boff + 16 * P : boff;
                h8 bh[3], bl[3];
                bh[0] = c0h; bl[0] = c0l; bh[1] = c1h; bl[1] = c1l;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const int pk = ks + 2;
                    const int po = pk < NKS ? boff + 32 * pk : nboff + 32 * (pk - NKS);
                    bh[pk % 3] = lds_frag<P>(hs + po, hs4 + po);
                    bl[pk % 3] = lds_frag<P>(ls + po, ls4 + po);
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], bh[ks % 3], acc, 0, 0, 0);
                    ax1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], bl[ks % 3], ax1, 0, 0, 0);
                    ax2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ks], bh[ks % 3], ax2, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                c0h = bh[NKS % 3]; c0l = bl[NKS % 3]; c1h = bh[(NKS + 1) % 3]; c1l = bl[(NKS + 1) % 3];
                boff = nboff;
                const int64_t t = t0 + tb * 16 + col;
                const f32x4 xs2 = {xr[j][2], xr[j][3], 0.f, 0.f};
                const f32x4 xv = nvalid >= 4 ? xr[j] : xs2;
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = xv[i] + alpha * fmaxf(acc[i] + (ax1[i] + ax2[i]) * (1.0f / 2048.0f), 0.f);
                float* yp = yb + t * C + g * CG + ch0;
                if (t < T) {
                    if (nvalid >= 4) *reinterpret_cast<f32x4u*>(yp) = o;
                    else if (nvalid == 2) { f32x2u q2 = {o[0], o[1]}; *reinterpret_cast<f32x2u*>(yp) = q2; }
                }
            }
        }
    }
    if constexpr (TAIL) {
        constexpr int CT = CG - 16, SH = 16 / CT, SPAN = 16 * SH;       // 2 channels x 8 shifts, 128 time steps per tile
        constexpr int NKT = ((KS - 1 + SH - 1) * P + CG + 31) / 32;
        static_assert(CT == 2 && GB * (TT / SPAN) == 4, "tail units");
        const int gl = w / (TT / SPAN), sb = w % (TT / SPAN), g = g0 + gl;
        h8 th[NKT], tl[NKT];
        const h8* wf = reinterpret_cast<const h8*>(wtail) + (int64_t)g * (NKT * 2 * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < NKT; ++ks) {
            th[ks] = wf[(ks * 2 + 0) * 64];
            tl[ks] = wf[(ks * 2 + 1) * 64];
        }
        const int cch = g * CG + 16 + (kg >> 1);                      // this lane's channel; its 4 values are 4 consecutive time steps
        const int64_t tl0 = t0 + sb * SPAN + SH * col + 4 * (kg & 1);
        float xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = xb[(tl0 + i < T ? tl0 + i : T - 1) * C + cch];
        const float bvt = bias[cch];
        f32x4 acc = {bvt, bvt, bvt, bvt}, ax1 = {0.f, 0.f, 0.f, 0.f}, ax2 = {0.f, 0.f, 0.f, 0.f};
        const _Float16* hs = s_hi + gl * SLAB + (sb * SPAN + SH * col) * P + 8 * kg;
        const _Float16* ls = s_lo + gl * SLAB + (sb * SPAN + SH * col) * P + 8 * kg;
        h8 bh[3], bl[3];
        bh[0] = lds_frag<P>(hs, hs + 4); bl[0] = lds_frag<P>(ls, ls + 4);
        bh[1] = lds_frag<P>(hs + 32, hs + 36); bl[1] = lds_frag<P>(ls + 32, ls + 36);
#pragma unroll
        for (int ks = 0; ks < NKT; ++ks) {
            const int pk = ks + 2 < NKT ? ks + 2 : NKT - 1;
            bh[(ks + 2) % 3] = lds_frag<P>(hs + 32 * pk, hs + 32 * pk + 4);
            bl[(ks + 2) % 3] = lds_frag<P>(ls + 32 * pk, ls + 32 * pk + 4);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(th[ks], bh[ks % 3], acc, 0, 0, 0);
            ax1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(th[ks], bl[ks % 3], ax1, 0, 0, 0);
            ax2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl[ks], bh[ks % 3], ax2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (tl0 + i < T) yb[(tl0 + i) * C + cch] = xv[i] + alpha * fmaxf(acc[i] + (ax1[i] + ax2[i]) * (1.0f / 2048.0f), 0.f);
    }
    if (clk && tid == 0) {
        const int64_t wg = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
        if (wg < 8192) {
            clk[3 * wg] = c_load - c_start;
            clk[3 * wg + 1] = clock64() - c_load;
            clk[3 * wg + 2] = wall_clock64() - w_start;
        }
    }
}

// ---- variant 2: the fp32 centre rows of the slab stay in LDS ([TT][GB*CG] floats): the MFMA waves take the residual
// from there and write x + alpha relu(conv) back in place (no second global read of x), then -- after a barrier -- the
// whole workgroup stores the tile with 16-byte row-contiguous writes, plus the hi / lo split of the same values for
// the next dense layer (8-byte pieces, GB*CG/4 of them contiguous per row).
template <int CG, int P, int GB, int TT, int WGPC, bool SPLIT>
__global__ __launch_bounds__(256, WGPC) void gconv_mfma2_kernel(const float* __restrict__ x, const _Float16* __restrict__ wfrag,
                                                            const float* __restrict__ bias, float alpha, float* __restrict__ y,
                                                            _Float16* __restrict__ ysplit, int64_t T, int C) {
    constexpr int KTOT = (KS - 1) * P + CG, NKS = (KTOT + 31) / 32, MT = (CG + 15) / 16;
    constexpr int TIN = TT + 2 * PAD;
    constexpr int SLAB = (TIN * P + (32 * NKS > KS * P ? 32 * NKS - KS * P : 0) + 7) & ~7;
    constexpr int CH = GB * CG, CH4 = CH / 4;
    constexpr int RPP = 256 / CH4;
    static_assert(CH % 4 == 0 && CG % 2 == 0 && TT % 16 == 0 && P % 2 == 0 && P >= CG, "shape");
    extern __shared__ __attribute__((aligned(16))) _Float16 slab[];   // [2 (hi, lo)][GB][SLAB] halves, then [TT][CH] floats
    _Float16* s_hi = slab;
    _Float16* s_lo = slab + GB * SLAB;
    float* xres = reinterpret_cast<float*>(slab + 2 * GB * SLAB);

    const int b = blockIdx.z, g0 = blockIdx.y * GB;
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* xb = x + (int64_t)b * T * C;
    float* yb = y + (int64_t)b * T * C;
    const int r0 = tid / CH4, c4 = tid - r0 * CH4;
    const bool active = r0 < RPP;
    {
        int so[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ch = c4 * 4 + 2 * q, gl = ch / CG;
            so[q] = gl * SLAB + (ch - gl * CG);
        }
        const float* xc = xb + g0 * CG + (active ? c4 * 4 : 0);
        constexpr int NPASS = (TIN + RPP - 1) / RPP, UNR = NPASS > 12 ? (NPASS + 1) / 2 : NPASS;
        for (int p0 = 0; p0 < NPASS; p0 += UNR) {
            f32x4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int64_t t = t0 - PAD + r0 + (p0 + u) * RPP;
                const int64_t tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
                v[u] = *reinterpret_cast<const f32x4*>(xc + tc * C);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ti = r0 + (p0 + u) * RPP;
                const int64_t t = t0 - PAD + ti;
                const bool in = t >= 0 && t < T;
                if (active && ti < TIN) {
                    if (ti >= PAD && ti < PAD + TT) *reinterpret_cast<f32x4*>(xres + (ti - PAD) * CH + c4 * 4) = v[u];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        _Float16 h0, l0, h1, l1;
                        split_f16x3(in ? v[u][2 * q] : 0.f, h0, l0);
                        split_f16x3(in ? v[u][2 * q + 1] : 0.f, h1, l1);
                        const h2 hh = {h0, h1}, ll = {l0, l1};
                        *reinterpret_cast<h2*>(s_hi + so[q] + ti * P) = hh;
                        *reinterpret_cast<h2*>(s_lo + so[q] + ti * P) = ll;
                    }
                }
            }
        }
        const h2 z2 = {(_Float16)0.f, (_Float16)0.f};
        if (P > CG)
            for (int i = tid; i < 2 * GB * TIN; i += 256)
#pragma unroll
                for (int c = CG; c < P; c += 2) *reinterpret_cast<h2*>(slab + (i / TIN) * SLAB + (i % TIN) * P + c) = z2;
        for (int i = tid; i < GB * (SLAB - TIN * P); i += 256) {
            const int a = i / ((SLAB - TIN * P) / 2), r = i - a * ((SLAB - TIN * P) / 2);
            *reinterpret_cast<h2*>(slab + a * SLAB + TIN * P + 2 * r) = z2;
        }
    }
    __syncthreads();

    constexpr int NU = GB * MT, NB = TT / 16;
    constexpr int UPW = NU >= 4 ? NU / 4 : 1;
    constexpr int WPU = NU >= 4 ? 1 : 4 / NU;
    constexpr int NBW = NB / WPU;
    static_assert(NU == 1 || NU == 2 || NU % 4 == 0, "units");
    static_assert(NB % WPU == 0, "blocks per wave");
    const int col = lane & 15, kg = lane >> 4;
    for (int uu = 0; uu < UPW; ++uu) {
        const int u = NU >= 4 ? w + 4 * uu : w / WPU;
        const int part = NU >= 4 ? 0 : w % WPU;
        const int gl = u / MT, mt = u - gl * MT, g = g0 + gl;
        h8 wh[NKS], wl[NKS];
        const h8* wf = reinterpret_cast<const h8*>(wfrag) + (int64_t)(g * MT + mt) * (NKS * 2 * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            wh[ks] = wf[(ks * 2 + 0) * 64];
            wl[ks] = wf[(ks * 2 + 1) * 64];
        }
        const int ch0 = mt * 16 + 4 * kg;
        const int nvalid = CG - ch0;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nvalid) bv[i] = bias[g * CG + ch0 + i];
        const _Float16* hs = s_hi + gl * SLAB + (part * NBW * 16 + col) * P + 8 * kg;
        const _Float16* ls = s_lo + gl * SLAB + (part * NBW * 16 + col) * P + 8 * kg;
        const _Float16* hs4 = hs + 4;
        const _Float16* ls4 = ls + 4;
        int boff = 0;
        h8 c0h = lds_frag<P>(hs, hs4), c0l = lds_frag<P>(ls, ls4);
        h8 c1h = lds_frag<P>(hs + 32, hs4 + 32), c1l = lds_frag<P>(ls + 32, ls4 + 32);
        const int tbeg = part * NBW, tend = (part + 1) * NBW;
        // this lane's residual / output slot: row (16 tb + col), channels gl*CG + ch0 .. + 3 (8-byte aligned)
        float* rp = xres + (tbeg * 16 + col) * CH + gl * CG + (nvalid > 0 ? ch0 : 0);
        for (int tb = tbeg; tb < tend; ++tb) {
            f32x4 acc = bv, ax1 = {0.f, 0.f, 0.f, 0.f}, ax2 = {0.f, 0.f, 0.f, 0.f};
            const int nboff = tb + 1 < tend ? boff + 16 * P : boff;
            h8 bh[3], bl[3];
            bh[0] = c0h; bl[0] = c0l; bh[1] = c1h; bl[1] = c1l;
            const f32x2 x01 = *reinterpret_cast<const f32x2*>(rp), x23 = *reinterpret_cast<const f32x2*>(rp + 2);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int pk = ks + 2;
                const int po = pk < NKS ? boff + 32 * pk : nboff + 32 * (pk - NKS);
                bh[pk % 3] = lds_frag<P>(hs + po, hs4 + po);
                bl[pk % 3] = lds_frag<P>(ls + po, ls4 + po);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], bh[ks % 3], acc, 0, 0, 0);
                ax1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], bl[ks % 3], ax1, 0, 0, 0);
                ax2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ks], bh[ks % 3], ax2, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            c0h = bh[NKS % 3]; c0l = bl[NKS % 3]; c1h = bh[(NKS + 1) % 3]; c1l = bl[(NKS + 1) % 3];
            boff = nboff;
            const f32x4 xv = {x01[0], x01[1], x23[0], x23[1]};
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = xv[i] + alpha * fmaxf(acc[i] + (ax1[i] + ax2[i]) * (1.0f / 2048.0f), 0.f);
            if (nvalid > 0) {
                const f32x2 o01 = {o[0], o[1]}, o23 = {o[2], o[3]};
                *reinterpret_cast<f32x2*>(rp) = o01;
                if (nvalid >= 4) *reinterpret_cast<f32x2*>(rp + 2) = o23;
            }
            rp += 16 * CH;
        }
    }
    __syncthreads();

    // ---- store phase: thread = (row inside a pass, 16-byte piece), whole rows of GB*CG floats ----
    if (active) {
        float* yc = yb + g0 * CG + c4 * 4;
        const int cb = g0 * CG + c4 * 4;                  // first channel of this thread's piece (a multiple of 4)
        _Float16* sc = SPLIT ? ysplit + (int64_t)b * T * 2 * C + (cb >> 5) * 64 + (cb & 31) : nullptr;   // (row stride: C halves x 2 = 2C bytes x 2)
        constexpr int NPS = (TT + RPP - 1) / RPP;
#pragma unroll 4
        for (int p = 0; p < NPS; ++p) {
            const int row = r0 + p * RPP;
            const int64_t t = t0 + row;
            if (row < TT && t < T) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(xres + row * CH + c4 * 4);
                *reinterpret_cast<f32x4*>(yc + t * C) = o;
                if (SPLIT) {
                    _Float16 h[4], l[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) split_f16x3(o[i], h[i], l[i]);
                    const h4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
                    _Float16* sp = sc + t * (2 * (int64_t)C);
                    *reinterpret_cast<h4*>(sp) = hv;
                    *reinterpret_cast<h4*>(sp + 32) = lv;
                }
            }
        }
    }
}

// reference Conv1d weight [C, CG, 21] -> MFMA A fragments [g][mt][ks][hi, lo][lane][8]
static void pack_weights(const std::vector<float>& w, int C, int CG, int P, std::vector<_Float16>& out) {
    const int G = C / CG, KTOT = (KS - 1) * P + CG, NKS = (KTOT + 31) / 32, MT = (CG + 15) / 16;
    out.assign((size_t)G * MT * NKS * 2 * 64 * 8, (_Float16)0.f);
    for (int g = 0; g < G; ++g)
        for (int mt = 0; mt < MT; ++mt)
            for (int ks = 0; ks < NKS; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 8; ++i) {
                        const int co = mt * 16 + (l & 15), k = 32 * ks + 8 * (l >> 4) + i;
                        float v = 0.f;
                        if (co < CG && k < KTOT && k % P < CG) v = w[((size_t)(g * CG + co) * CG + k % P) * KS + k / P];
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)((v - (float)hi) * 2048.f);
                        const size_t base = ((((size_t)(g * MT + mt) * NKS + ks) * 2) * 64 + l) * 8 + i;
                        out[base] = hi;
                        out[base + 64 * 8] = lo;
                    }
}

// tail tile of a group (CG > 16): row r = c * SH + s -> channel 16 + c moved s taps down: A[r][k] = W[16 + c][k - s P]
static void pack_tail(const std::vector<float>& w, int C, int CG, int P, std::vector<_Float16>& out) {
    const int G = C / CG, CT = CG - 16, SH = 16 / CT, KTOT = (KS - 1) * P + CG, NKT = ((KS - 1 + SH - 1) * P + CG + 31) / 32;
    out.assign((size_t)G * NKT * 2 * 64 * 8, (_Float16)0.f);
    for (int g = 0; g < G; ++g)
        for (int ks = 0; ks < NKT; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 8; ++i) {
                    const int r = l & 15, c = r / SH, sft = r % SH, k = 32 * ks + 8 * (l >> 4) + i - sft * P;
                    float v = 0.f;
                    if (k >= 0 && k < KTOT && k % P < CG) v = w[((size_t)(g * CG + 16 + c) * CG + k % P) * KS + k / P];
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)((v - (float)hi) * 2048.f);
                    const size_t base = ((((size_t)g * NKT + ks) * 2) * 64 + l) * 8 + i;
                    out[base] = hi;
                    out[base + 64 * 8] = lo;
                }
}

template <int CG, int P, int GB, int TT, int WGPC = 2, bool TAIL = false>
static void run(int64_t T, bool check) {
    const int G = 80, C = G * CG;
    std::vector<float> hx((size_t)T * C), hw((size_t)C * CG * KS), hb(C);
    uint64_t s = 777;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 33) & 0xffffff) / 8388608.0f - 1.0f; };
    for (auto& v : hx) v = rnd() * 2.0f;
    for (auto& v : hw) v = rnd() * 0.1f;
    for (auto& v : hb) v = rnd() * 0.2f;
    std::vector<_Float16> hf;
    pack_weights(hw, C, CG, P, hf);
    float *dx, *dy, *db;
    _Float16* df;
    (void)hipMalloc(&dx, hx.size() * 4);
    (void)hipMalloc(&dy, hx.size() * 4);
    (void)hipMalloc(&db, hb.size() * 4);
    (void)hipMalloc(&df, hf.size() * 2);
    (void)hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(df, hf.data(), hf.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(dy, 0xff, hx.size() * 4);
    long long* dclk;
    (void)hipMalloc(&dclk, 3 * 8192 * 8);
    (void)hipMemset(dclk, 0, 3 * 8192 * 8);
    constexpr int KTOT = (KS - 1) * P + CG, NKS = (KTOT + 31) / 32, TIN = TT + 2 * PAD;
    constexpr int SLAB = (TIN * P + (32 * NKS > KS * P ? 32 * NKS - KS * P : 0) + 7) & ~7;
    constexpr size_t lds = (size_t)2 * GB * SLAB * 2;
    auto kern = gconv_mfma_kernel<CG, P, GB, TT, WGPC, TAIL>;
    _Float16* dt = nullptr;
    if (TAIL) {
        std::vector<_Float16> ht;
        pack_tail(hw, C, CG, P, ht);
        (void)hipMalloc(&dt, ht.size() * 2);
        (void)hipMemcpy(dt, ht.data(), ht.size() * 2, hipMemcpyHostToDevice);
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)((T + TT - 1) / TT), G / GB, 1);
    const float alpha = 0.37f;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < (check ? 1 : 20); ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, dx, df, dt, db, alpha, dy, T, C, dclk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("CG=%d P=%d GB=%d TT=%d WGPC=%d tail=%d T=%lld: lds %zu B, %.3f ms, %.1f fp32-equivalent TFLOP/s, %.2f TB/s in+out (%s)\n", CG, P, GB, TT, WGPC, (int)TAIL, (long long)T,
           lds, best, 2.0 * T * C * CG * KS / best / 1e9, 2.0 * T * C * 4 / best / 1e9, hipGetErrorString(hipGetLastError()));
    if (!check) {
        std::vector<long long> hc(3 * 8192);
        (void)hipMemcpy(hc.data(), dclk, hc.size() * 8, hipMemcpyDeviceToHost);
        const int64_t nwg = (int64_t)grid.x * grid.y < 8192 ? (int64_t)grid.x * grid.y : 8192;
        double a = 0, bb = 0, c = 0;
        for (int64_t i = 0; i < nwg; ++i) { a += hc[3 * i]; bb += hc[3 * i + 1]; c += hc[3 * i + 2]; }
        printf("  per workgroup (wave 0, mean of %lld): slab load %.0f cycles, MFMA phase %.0f cycles, %.2f us wall -> %.2f GHz; %u workgroups\n",
               (long long)nwg, a / nwg, bb / nwg, c / nwg / 100.0, (a + bb) / c / 10.0, grid.x * grid.y);
    }
    if (check) {
        std::vector<float> hy(hx.size());
        (void)hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, vmax = 0;
        int64_t bad = 0, n = 0;
        for (int64_t t = 0; t < T; t += (t < 40 || t > T - 40) ? 1 : 13)
            for (int c = 0; c < C; c += 1) {
                const int g = c / CG;
                double a = hb[c];
                for (int ci = 0; ci < CG; ++ci)
                    for (int k = 0; k < KS; ++k) {
                        const int64_t ti = t + k - PAD;
                        if (ti >= 0 && ti < T) a += (double)hw[((size_t)c * CG + ci) * KS + k] * (double)hx[ti * C + g * CG + ci];
                    }
                const double ref = hx[t * C + c] + alpha * (a > 0 ? a : 0);
                const double e = fabs(ref - (double)hy[t * C + c]);
                if (!(e < 1e-4)) ++bad;
                emax = fmax(emax, e);
                vmax = fmax(vmax, fabs(ref));
                ++n;
            }
        printf("  checked %lld outputs: max |err| vs float64 %.3e (max |value| %.2f), %lld beyond 1e-4\n", (long long)n, emax, vmax, (long long)bad);
    }
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(db); (void)hipFree(df);
}

template <int CG, int P, int GB, int TT, int WGPC, bool SPLIT>
static void run2(int64_t T, bool check) {
    const int G = 80, C = G * CG;
    std::vector<float> hx((size_t)T * C), hw((size_t)C * CG * KS), hb(C);
    uint64_t s = 4242;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 33) & 0xffffff) / 8388608.0f - 1.0f; };
    for (auto& v : hx) v = rnd() * 2.0f;
    for (auto& v : hw) v = rnd() * 0.1f;
    for (auto& v : hb) v = rnd() * 0.2f;
    std::vector<_Float16> hf;
    pack_weights(hw, C, CG, P, hf);
    float *dx, *dy, *db;
    _Float16 *df, *ds;
    (void)hipMalloc(&dx, hx.size() * 4);
    (void)hipMalloc(&dy, hx.size() * 4);
    (void)hipMalloc(&ds, hx.size() * 4);
    (void)hipMalloc(&db, hb.size() * 4);
    (void)hipMalloc(&df, hf.size() * 2);
    (void)hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(df, hf.data(), hf.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(dy, 0xff, hx.size() * 4);
    (void)hipMemset(ds, 0xff, hx.size() * 4);
    constexpr int KTOT = (KS - 1) * P + CG, NKS = (KTOT + 31) / 32, TIN = TT + 2 * PAD;
    constexpr int SLAB = (TIN * P + (32 * NKS > KS * P ? 32 * NKS - KS * P : 0) + 7) & ~7;
    constexpr size_t lds = (size_t)2 * GB * SLAB * 2 + (size_t)TT * GB * CG * 4 + 16;
    auto kern = gconv_mfma2_kernel<CG, P, GB, TT, WGPC, SPLIT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)((T + TT - 1) / TT), G / GB, 1);
    const float alpha = 0.37f;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < (check ? 1 : 20); ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, dx, df, db, alpha, dy, ds, T, C);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("v2 CG=%d P=%d GB=%d TT=%d WGPC=%d split=%d T=%lld: lds %zu B, %.3f ms, %.1f fp32-equivalent TFLOP/s (%s)\n", CG, P, GB, TT, WGPC,
           (int)SPLIT, (long long)T, lds, best, 2.0 * T * C * CG * KS / best / 1e9, hipGetErrorString(hipGetLastError()));
    if (check) {
        std::vector<float> hy(hx.size());
        std::vector<_Float16> hs(hx.size() * 2);
        (void)hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hs.data(), ds, hs.size() * 2, hipMemcpyDeviceToHost);
        double emax = 0;
        int64_t bad = 0, n = 0, sbad = 0;
        for (int64_t t = 0; t < T; t += (t < 40 || t > T - 40) ? 1 : 7)
            for (int c = 0; c < C; c += 1) {
                const int g = c / CG;
                double a = hb[c];
                for (int ci = 0; ci < CG; ++ci)
                    for (int k = 0; k < KS; ++k) {
                        const int64_t ti = t + k - PAD;
                        if (ti >= 0 && ti < T) a += (double)hw[((size_t)c * CG + ci) * KS + k] * (double)hx[ti * C + g * CG + ci];
                    }
                const double ref = hx[t * C + c] + alpha * (a > 0 ? a : 0);
                const double e = fabs(ref - (double)hy[t * C + c]);
                if (!(e < 1e-4)) ++bad;
                emax = fmax(emax, e);
                ++n;
                if (SPLIT) {
                    const float v = hy[t * C + c];
                    const _Float16 hi = (_Float16)v, lo = (_Float16)((v - (float)hi) * 2048.f);
                    const size_t o = ((size_t)t * (C / 32) + c / 32) * 64 + c % 32;
                    if (!((float)hs[o] == (float)hi && (float)hs[o + 32] == (float)lo)) ++sbad;
                }
            }
        printf("  checked %lld outputs: max |err| vs float64 %.3e, %lld beyond 1e-4, %lld split mismatches\n", (long long)n, emax, (long long)bad,
               (long long)sbad);
    }
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(db); (void)hipFree(df); (void)hipFree(ds);
}

int main(int argc, char** argv) {
    if (argc > 1) {          // profiling: one shape, few launches
        run<14, 16, 4, 256>(89986, false);
        run<18, 24, 2, 256, 2, true>(44983, false);
        return 0;
    }
    run<18, 24, 2, 256, 2, true>(1500, true);
    run<18, 24, 2, 256, 2, true>(77, true);
    run<18, 24, 2, 256>(44983, false);
    run<18, 24, 2, 256, 2, true>(44983, false);
    return 0;
}
