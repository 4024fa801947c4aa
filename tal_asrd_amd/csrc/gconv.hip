// Grouped temporal convolutions of the TDS encoder (k = 21, groups = n_mels = 80).
//
//   stride-2 "resize" conv, padding 0 ........ tal/asr/models.py:363-364
//   TDSBlock grouped conv + ReLU + ReZero ..... tal/asr/models.py:304-308,329
//
// Activations are time-major [B, T, C].  Three kernels:
//   gconv_mfma_kernel   long inputs, 10 / 14 / 18 channels per group (and the stride-2 convs between them): on the fp16
//                       matrix cores in the fp16x3 form, through a Hankel view of a time-major LDS slab (second half of
//                       this file);
//   gconv_s2_c1_kernel  the first resize conv (1 mel bin per group -> 10 channels): store-bound, channel-major lanes;
//   gconv_kernel        fp32 VALU kernel for everything else (short inputs, TAL_TDS_F32=1, other widths): a workgroup
//                       stages a [GB groups x C/G channels] x [time tile + halo] slab of x into LDS, channel-major with
//                       an odd row pitch (conflict-free both for the transposing store and for lanes that walk
//                       consecutive time steps); each wave owns one group at a time and a lane accumulates all C_out/G
//                       output channels of R time steps, so the weights of a group are wave-uniform and travel through
//                       SGPRs (s_load + v_fmac with a scalar operand) while the x taps come from LDS.
#include "common.h"

namespace tal {

constexpr int KS = 21;

template <int CIG, int COG, int STRIDE, int GB, int TT, bool RESID>
__global__ __launch_bounds__(256) void gconv_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                   const float* __restrict__ bias, float alpha,
                                                   float* __restrict__ y, int64_t T_in, int64_t T_out, int C_in,
                                                   int C_out) {
    constexpr int PAD = RESID ? KS / 2 : 0;
    constexpr int TIN = (TT - 1) * STRIDE + KS;
    constexpr int TINP = TIN | 1;
    // GB >= 4: wave w walks groups w, w+4, ... over the whole time tile.  GB < 4: 4/GB waves share a
    // group and split the time tile (smaller LDS slab per workgroup -> more workgroups per CU, so
    // one workgroup's slab load overlaps another's FMA phase).
    constexpr int WPG = GB >= 4 ? 1 : 4 / GB;      // waves per group
    constexpr int TW = TT / WPG;                   // time steps per wave
    constexpr int R = TW / 64;
    static_assert(TW % 64 == 0 && R >= 1, "time tile per wave must be a multiple of 64");
    constexpr int CH = GB * CIG;
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [CH][TINP]

    const int b = blockIdx.z;
    const int g0 = blockIdx.y * GB;
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = wave_id();

    const float* xb = x + (int64_t)b * T_in * C_in + g0 * CIG;
    const int64_t tin0 = t0 * STRIDE - PAD;
    // Slab load: 16-byte global loads (the CH channels of a time step are contiguous and 16-byte
    // aligned), four in flight per thread with clamped addresses + select (no branch in front of a
    // load), then the transposing scalar LDS stores.
    static_assert(CH % 4 == 0, "a group slab must be a whole number of 16-byte columns");
    constexpr int CH4 = CH / 4;
    constexpr int NV = TIN * CH4;
    constexpr int UNR = 4;
    for (int base = 0; base < NV; base += 256 * UNR) {
        f32x4 v[UNR];
        int tiv[UNR], cv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = base + u * 256 + tid;
            const int ti = idx / CH4;
            tiv[u] = idx < NV ? ti : -1;
            cv[u] = (idx - ti * CH4) * 4;
            const int64_t t = tin0 + ti;
            const bool in = idx < NV && t >= 0 && t < T_in;
            const int64_t tc = in ? t : (T_in - 1 < 0 ? 0 : (t < 0 ? 0 : T_in - 1));
            const f32x4 ld = *reinterpret_cast<const f32x4*>(xb + tc * C_in + (idx < NV ? cv[u] : 0));
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v[u] = in ? ld : z;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (tiv[u] >= 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xs[(cv[u] + q) * TINP + tiv[u]] = v[u][q];
            }
    }
    __syncthreads();

    const int part = w % WPG;                      // which slice of the time tile this wave owns
    const int tl0 = part * TW + lane;              // first local time step of this lane
    for (int gl = w / WPG; gl < GB; gl += 4 / WPG) {
        const int g = g0 + gl;
        float acc[R][COG];
#pragma unroll
        for (int co = 0; co < COG; ++co) {
            const float bv = bias[g * COG + co];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][co] = bv;
        }
        const float* wg = wp + (int64_t)g * (CIG * KS * COG);
        const float* xl = xs + gl * CIG * TINP + tl0 * STRIDE;
#pragma unroll 1
        for (int ci = 0; ci < CIG; ++ci) {
            const float* wc = wg + ci * (KS * COG);
            const float* xc = xl + ci * TINP;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                float xv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) xv[r] = xc[r * 64 * STRIDE + k];
#pragma unroll
                for (int co = 0; co < COG; ++co) {
                    const float wv = wc[k * COG + co];
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][co] = fmaf(wv, xv[r], acc[r][co]);
                }
            }
        }
        float* yb = y + (int64_t)b * T_out * C_out + g * COG;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t t = t0 + tl0 + 64 * r;
            if (t < T_out) {
#pragma unroll
                for (int co = 0; co < COG; ++co) {
                    float v = acc[r][co];
                    if (RESID) {
                        const float xin = xs[(gl * CIG + co) * TINP + tl0 + 64 * r + PAD];
                        v = xin + alpha * fmaxf(v, 0.f);
                    }
                    yb[t * C_out + co] = v;
                }
            }
        }
    }
}

// Shape-generic fallback (any channels-per-group): one thread per output element.
// Only used for widths the specialised kernels do not cover (small unit-test models).
template <bool RESID>
__global__ void gconv_generic_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                     const float* __restrict__ bias, float alpha, float* __restrict__ y,
                                     int64_t T_in, int64_t T_out, int C_in, int C_out, int cig, int cog, int stride,
                                     int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C_out);
    const int64_t bt = i / C_out;
    const int64_t t = bt % T_out;
    const int64_t b = bt / T_out;
    const int g = c / cog, co = c - g * cog;
    const int pad = RESID ? KS / 2 : 0;
    float acc = bias[c];
    const float* xb = x + b * T_in * C_in + g * cig;
    const float* wg = wp + (int64_t)g * cig * KS * cog + co;
    for (int ci = 0; ci < cig; ++ci)
        for (int k = 0; k < KS; ++k) {
            const int64_t ti = t * stride - pad + k;
            if (ti >= 0 && ti < T_in) acc = fmaf(wg[(ci * KS + k) * cog], xb[ti * C_in + ci], acc);
        }
    if (RESID) acc = x[(b * T_in + t) * C_in + c] + alpha * fmaxf(acc, 0.f);
    y[i] = acc;
}

// First resize conv of the encoder: 1 input channel per group (the 80 mel bins) -> COG outputs per group, stride 2.
// 21 MACs per output: the launch is bound by its stores (576 MB per hour of audio), so the lane <-> channel mapping is
// what matters: a lane owns ONE output channel (its 21 weights + bias live in registers) and walks the time axis, so a
// wave stores 64 consecutive channels of a time step = 256 contiguous bytes.  (With lane = time step, as in the generic
// kernel above, a lane's 10 outputs are a 40-byte piece of a 3200-byte row: 2 TB/s.)  The mel slab of the tile sits in
// LDS time-major ([row][groups of the workgroup]); lanes of one group read the same word (broadcast).  Four output
// steps per iteration share their 27 input samples.  Same fmaf order as the generic kernel: bit-identical results.
// YSPLIT: the output is written in the hi / lo split form of the fp16x3 dense layers only: lane pairs exchange halves so
// that even lanes store two hi halves and odd lanes two lo halves (4-byte stores, 64 contiguous bytes per 32 channels).
// in_mean != NULL: x is the log-mel BEFORE LogMelSpec's global-mean subtraction (tal/asr/models.py:52) and in_mean[0] the scalar:
// the conv has no padding, so every output sees all 21 taps and conv(x - m) = conv(x) - m * sum_k w[k] exactly in real
// arithmetic -- the subtraction becomes a correction of the channel's bias and the separate pass over the log-mel disappears.
template <int COG, int NG, int TT, bool YSPLIT>
__global__ __launch_bounds__(256) void gconv_s2_c1_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                         const float* __restrict__ bias, float* __restrict__ y, int64_t T_in,
                                                         int64_t T_out, int C_in, int C_out, int* __restrict__ range_flag,
                                                         const float* __restrict__ in_mean) {
    constexpr int NC = NG * COG;                       // output channels of this workgroup (<= 256)
    constexpr int TIN = (TT - 1) * 2 + KS;
    static_assert(NC <= 256 && NG % 4 == 0 && TT % 4 == 0, "shape");
    __shared__ __attribute__((aligned(16))) float xs[TIN * NG + 8];
    const int b = blockIdx.z, g0 = blockIdx.y * NG;
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    const int tid = threadIdx.x;
    const float* xb = x + (int64_t)b * T_in * C_in + g0;
    for (int i = tid; i < TIN * (NG / 4); i += 256) {
        const int r = i / (NG / 4), c = (i - r * (NG / 4)) * 4;
        int64_t t = t0 * 2 + r;
        t = t < T_in ? t : T_in - 1;                   // rows past the end feed outputs past T_out only
        *reinterpret_cast<f32x4*>(xs + r * NG + c) = *reinterpret_cast<const f32x4*>(xb + t * C_in + c);
    }
    __syncthreads();
    if (tid >= NC) return;
    const int ch = g0 * COG + tid;                     // output channel; packed weight layout [G][1][21][COG]
    float wk[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) wk[k] = wp[((int64_t)(g0 + tid / COG) * KS + k) * COG + tid % COG];
    float bv = bias[ch];
    if (in_mean) {
        float wsum = wk[0];
#pragma unroll
        for (int k = 1; k < KS; ++k) wsum += wk[k];
        bv = fmaf(-in_mean[0], wsum, bv);
    }
    const float* xc = xs + tid / COG;
    float* yc = y + (int64_t)b * T_out * C_out + ch;
    // split form: row = C_out * 4 bytes; channel ch -> 32-channel block ch / 32 (128 bytes), hi half at slot ch % 32, lo 64 bytes on.
    // This lane stores the pair (ch & ~1): the hi halves when ch is even, the lo halves when odd (ch and the lane index have
    // the same parity: g0 * COG is even).
    _Float16* ysp = reinterpret_cast<_Float16*>(y) + (int64_t)b * T_out * C_out * 2 + ((ch & ~1) >> 5) * 64 + ((ch & ~1) & 31) + (ch & 1) * 32;
    float amax = 0.f;
    for (int tl = 0; tl < TT; tl += 4) {
        float xv[KS + 6];
#pragma unroll
        for (int i = 0; i < KS + 6; ++i) xv[i] = xc[(2 * tl + i) * NG];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float acc = bv;
#pragma unroll
            for (int k = 0; k < KS; ++k) acc = fmaf(wk[k], xv[2 * q + k], acc);
            const int64_t t = t0 + tl + q;
            if (YSPLIT) {
                _Float16 hi, lo;
                split_f16x3(acc, hi, lo);
                amax = fmaxf(amax, fabsf(acc));
                const unsigned short hb = __builtin_bit_cast(unsigned short, hi), lb = __builtin_bit_cast(unsigned short, lo);
                const unsigned mine = (unsigned)hb | ((unsigned)lb << 16);
                const unsigned other = __shfl_xor(mine, 1, 64);
                // even lane: {hi(ch), hi(ch + 1)}; odd lane: {lo(ch - 1), lo(ch)}
                const unsigned word = (tid & 1) ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
                if (t < T_out) *reinterpret_cast<unsigned*>(ysp + t * C_out * 2) = word;
            } else if (t < T_out) yc[t * C_out] = acc;
        }
    }
    if (YSPLIT) note_range(amax, range_flag);
}

template <int CIG, int COG, int STRIDE, int GB, int TT, bool RESID>
static int launch_spec(const float* x, const float* wp, const float* bias, float alpha, float* y, int B, int64_t T_in,
                       int64_t T_out, int C_in, int C_out, int groups, hipStream_t s) {
    constexpr int TIN = (TT - 1) * STRIDE + KS;
    constexpr int TINP = TIN | 1;
    constexpr size_t lds = (size_t)GB * CIG * TINP * sizeof(float);
    auto kern = gconv_kernel<CIG, COG, STRIDE, GB, TT, RESID>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            set_error("gconv: cannot reserve %zu bytes of LDS", lds);
            return TAL_EHIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)cdiv(T_out, TT), (unsigned)(groups / GB), (unsigned)B);
    ProfScope prof(RESID ? PROF_GCONV_RES : PROF_GCONV_S2, 2.0 * (double)B * (double)T_out * C_out * CIG * KS, s);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, wp, bias, alpha, y, T_in, T_out, C_in, C_out);
    TAL_CHECK_LAUNCH("gconv");
    return TAL_OK;
}

template <bool RESID>
static int launch_generic(const float* x, const float* wp, const float* bias, float alpha, float* y, int B,
                          int64_t T_in, int64_t T_out, int C_in, int C_out, int groups, int stride, hipStream_t s) {
    const int64_t total = (int64_t)B * T_out * C_out;
    if (total == 0) return TAL_OK;
    hipLaunchKernelGGL(gconv_generic_kernel<RESID>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, x, wp, bias,
                       alpha, y, T_in, T_out, C_in, C_out, C_in / groups, C_out / groups, stride, total);
    TAL_CHECK_LAUNCH("gconv_generic");
    return TAL_OK;
}

bool gconv_s2_can_split(int C_in, int C_out, int groups, const float* x) {
    const bool c1_generic = opt(OPT_GCONV_C1_GENERIC) != 0;
    return groups > 0 && C_in == groups && C_out == 10 * groups && groups % 20 == 0 && C_in % 4 == 0 && C_out % 32 == 0 &&
           (reinterpret_cast<uintptr_t>(x) & 15) == 0 && !c1_generic;
}

// the shapes whose first resize conv can take the mean of its input as a bias correction (gconv_s2_c1_kernel)
bool gconv_s2_can_fold_mean(int C_in, int C_out, int groups, const float* x) {
    return groups > 0 && C_in == groups && C_out == 10 * groups && groups % 20 == 0 && C_in % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(x) & 15) == 0 && !opt(OPT_GCONV_C1_GENERIC);
}

int launch_gconv_s2(const float* x, const float* wp, const float* bias, int B, int64_t T_in, int C_in, int C_out,
                    int groups, float* y, hipStream_t s, void* y_split, int* range_flag, const float* in_mean) {
    TAL_CHECK_ARG(x && wp && bias && (y || y_split), "tal_gconv_s2_fwd: null pointer");
    TAL_CHECK_ARG(!in_mean || gconv_s2_can_fold_mean(C_in, C_out, groups, x), "tal_gconv_s2_fwd: no mean-folding kernel for this shape");
    TAL_CHECK_ARG(!y_split || gconv_s2_can_split(C_in, C_out, groups, x), "tal_gconv_s2_fwd: no split-output kernel for this shape");
    TAL_CHECK_ARG(groups > 0 && C_in % groups == 0 && C_out % groups == 0, "tal_gconv_s2_fwd: channels %d->%d not divisible by groups %d", C_in, C_out, groups);
    TAL_CHECK_ARG(B > 0 && T_in >= KS, "tal_gconv_s2_fwd: T_in=%lld shorter than the kernel", (long long)T_in);
    const int64_t T_out = (T_in - KS) / 2 + 1;
    const int cig = C_in / groups, cog = C_out / groups;
    const bool c1_generic = opt(OPT_GCONV_C1_GENERIC) != 0;
    if (cig == 1 && cog == 10 && groups % 20 == 0 && C_in % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && !c1_generic) {
        // channel-major lanes: coalesced 256-byte stores (0.29 -> 0.16 ms on the 1-hour shape)
        constexpr int NG = 20, TT = 256, TTS = 32;
        ProfScope prof(PROF_GCONV_S2, 2.0 * (double)B * (double)T_out * C_out * KS, s);
        // a lane walks its tile's time steps one after the other: short inputs (a 30-second clip is 6 tiles of 256 steps x 4
        // group blocks = 24 workgroups, 42 us) take 32-step tiles so that the launch covers the chip
        const bool small = cdiv(T_out, TT) * (groups / NG) * B < 256;
        dim3 grid((unsigned)cdiv(T_out, small ? TTS : TT), (unsigned)(groups / NG), (unsigned)B);
        float* yo = y_split ? reinterpret_cast<float*>(y_split) : y;
        if (y_split && small)
            hipLaunchKernelGGL((gconv_s2_c1_kernel<10, NG, TTS, true>), grid, dim3(256), 0, s, x, wp, bias, yo, T_in, T_out, C_in, C_out, range_flag, in_mean);
        else if (y_split)
            hipLaunchKernelGGL((gconv_s2_c1_kernel<10, NG, TT, true>), grid, dim3(256), 0, s, x, wp, bias, yo, T_in, T_out, C_in, C_out, range_flag, in_mean);
        else if (small)
            hipLaunchKernelGGL((gconv_s2_c1_kernel<10, NG, TTS, false>), grid, dim3(256), 0, s, x, wp, bias, yo, T_in, T_out, C_in, C_out, range_flag, in_mean);
        else
            hipLaunchKernelGGL((gconv_s2_c1_kernel<10, NG, TT, false>), grid, dim3(256), 0, s, x, wp, bias, yo, T_in, T_out, C_in, C_out, range_flag, in_mean);
        TAL_CHECK_LAUNCH("gconv (1 channel per group)");
        return TAL_OK;
    }
    if (cig == 1 && cog == 10 && groups % 16 == 0)
        return launch_spec<1, 10, 2, 16, 128, false>(x, wp, bias, 0.f, y, B, T_in, T_out, C_in, C_out, groups, s);
    // Measured on the 1-hour shapes (TFLOP/s): 10->14: 4 groups x 128 outputs 73, 2 x 128 72, 2 x 256 67;
    // 14->18: 2 x 128 75 (31 KB slab, 5 workgroups / CU), 4 x 128 53 (62 KB, 2 / CU), 2 x 256 46.
    // (De-interleaving the slab into even / odd time steps, which removes the 2-way bank conflict of the
    //  stride-2 tap reads, changed nothing: the LDS reads are not the limiter.)
    if (cig == 10 && cog == 14 && groups % 4 == 0)
        return launch_spec<10, 14, 2, 4, 128, false>(x, wp, bias, 0.f, y, B, T_in, T_out, C_in, C_out, groups, s);
    if (cig == 14 && cog == 18 && groups % 2 == 0)
        return launch_spec<14, 18, 2, 2, 128, false>(x, wp, bias, 0.f, y, B, T_in, T_out, C_in, C_out, groups, s);
    return launch_generic<false>(x, wp, bias, 0.f, y, B, T_in, T_out, C_in, C_out, groups, 2, s);
}

int launch_gconv_res(const float* x, const float* wp, const float* bias, float alpha, int B, int64_t T, int C,
                     int groups, float* y, hipStream_t s) {
    TAL_CHECK_ARG(x && wp && bias && y, "tal_gconv_res_fwd: null pointer");
    TAL_CHECK_ARG(x != y, "tal_gconv_res_fwd: in-place not supported (halo reads)");
    TAL_CHECK_ARG(groups > 0 && C % groups == 0, "tal_gconv_res_fwd: C=%d not divisible by groups %d", C, groups);
    TAL_CHECK_ARG(B > 0 && T > 0, "tal_gconv_res_fwd: bad shape");
    const int cg = C / groups;
    // 2 groups x 256 time steps per workgroup (2 waves per group): 22-40 KB slabs, 4+ workgroups
    // per CU.  Measured on the 1-hour shapes: 66 / 85 / 85 TFLOP/s vs 61 / 67 / 70 with 4 groups.
    if (cg == 10 && groups % 2 == 0)
        return launch_spec<10, 10, 1, 2, 256, true>(x, wp, bias, alpha, y, B, T, T, C, C, groups, s);
    if (cg == 14 && groups % 2 == 0)
        return launch_spec<14, 14, 1, 2, 256, true>(x, wp, bias, alpha, y, B, T, T, C, C, groups, s);
    if (cg == 18 && groups % 2 == 0)
        return launch_spec<18, 18, 1, 2, 256, true>(x, wp, bias, alpha, y, B, T, T, C, C, groups, s);
    return launch_generic<true>(x, wp, bias, alpha, y, B, T, T, C, C, groups, 1, s);
}

// ---------------------------------------------------------------------------------------------
// Grouped convs on the fp16 matrix cores (fp16x3 form, as the dense layers of the block)
// ---------------------------------------------------------------------------------------------
// Per group the conv is a small GEMM  out[co, t] = sum_k W[co, k] X[k, t],  k = j * P + c  (j-th tap of a K segment,
// channel c).  A segment's input slab sits in LDS TIME-major with row pitch P halves ([row][P], hi and lo arrays), so
// column t of X is simply the ntap * P consecutive halves that start at slab[t * P] (a Hankel matrix): one MFMA operand
// fragment (8 consecutive k of one column) is one LDS read, no im2col.  M = output channels (weights: register-resident
// MFMA A fragments, zero rows past C_out/G, zero columns for pad channels and past the last tap), N = 16 output steps,
// K rounded up to 32 per segment.  v_mfma_f32_16x16x32_f16 x 3 per product block (hi*hi, hi*lo, lo*hi; fp32 accumulate,
// three independent accumulators).  Segments keep every fragment 16-byte aligned and bank-conflict-free:
//   stride 1, C_in/G <= 16 : one segment, 21 taps (P = 16; P = 10 compact for 10 channels: that stage is HBM-bound)
//   stride 1, C_in/G = 18  : channels 0-15 (P = 16) + channels 16-17 (P = 8); a single P = 24 slab measured 37 % of
//                            the kernel time in LDS bank-conflict cycles (SQ_LDS_BANK_CONFLICT)
//   stride 2               : even taps read the slab of even input rows, odd taps the slab of odd rows (P = 16 each), so
//                            consecutive output steps stay one row apart
// The C layout hands a lane 4 consecutive channels of one output step: the TDSBlock residual is one 16-byte load (fetched
// three blocks ahead: an L2 round trip outlasts a block), the output one 16-byte store; optionally the hi / lo split of
// the output for the next dense layer is stored too (measured slower than the separate split pass: see tal_tds_fwd).
// Measured (1-hour shapes): TDSBlock convs 0.60 / 0.33 / 0.31 ms per launch against 0.65 / 0.59 / 0.54 for the VALU
// kernel above; max error against float64 5e-7 (values of a few units).  scripts/ubench/gconv_mfma.hip is the test bed.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef _Float16 f16x8u __attribute__((ext_vector_type(8), aligned(4)));
typedef _Float16 f16x8a8 __attribute__((ext_vector_type(8), aligned(8)));
// slab row pitch (halves) of a 10-channel group: 10 = compact (20-byte rows: every fragment read is 4-byte aligned and half
// of the LDS cycles are bank conflicts), 12 = 24-byte rows (8-byte aligned reads, one more K chunk of 32)
#ifndef GC_P10
#define GC_P10 12
#endif
#ifndef GC_P18
#define GC_P18 20
#endif
// workgroups per CU the register budget is held to (ablation: GC_LB18 = 3 for the 18-channel stride-1 kernels)
#ifndef GC_LB18
#define GC_LB18 2
#endif
// how many K chunks ahead of their MFMAs the operand fragments are read from LDS
#ifndef GC_PF
#define GC_PF 2
#endif
// K permutation of the fragment reads (GcLayout::PERM); 0: plain order (ablation)
#ifndef GC_KPERM
#define GC_KPERM 1
#endif
#define GC_LB(CIG, STRIDE) (((CIG) > 16 && (STRIDE) == 1) ? GC_LB18 : 2)
typedef _Float16 f16x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// K-segment layout (host + device, compile time)
template <int CIG, int STRIDE>
struct GcLayout {
    // 18 channels per group, stride 1: one segment of 40-byte rows (GC_P18 = 20: 14 K chunks, 8-byte aligned fragment reads;
    // -10 % on the kernel alone, -0.5 % on the 1-hour step) or, with GC_P18 = 0, two K segments (16 channels at pitch 16 + 2 at
    // pitch 8: 17 K chunks, every fragment one aligned 16-byte read -- round 1's form; a 48-byte pitch lost 37 % to bank conflicts)
    static constexpr bool SPLIT18 = CIG > 16 && STRIDE == 1 && GC_P18 == 0;
    static constexpr int NSEG = (STRIDE == 2 || SPLIT18) ? 2 : 1;
    static constexpr int pitch(int s) { return STRIDE == 2 ? 16 : (SPLIT18 ? (s == 0 ? 16 : 8) : (CIG > 16 ? GC_P18 : (CIG == 10 ? GC_P10 : 16))); }
    static constexpr int ntap(int s) { return STRIDE == 2 ? (s == 0 ? (KS + 1) / 2 : KS / 2) : KS; }
    static constexpr int nch(int s) { return SPLIT18 ? (s == 0 ? 16 : CIG - 16) : CIG; }
    static constexpr int choff(int s) { return (SPLIT18 && s == 1) ? 16 : 0; }
    static constexpr int tap0(int s) { return STRIDE == 2 ? s : 0; }
    static constexpr int nk(int s) { return s < NSEG ? ((ntap(s) - 1) * pitch(s) + nch(s) + 31) / 32 : 0; }
    static constexpr int NKS = nk(0) + nk(1);
    // K permutation against LDS bank conflicts (single-segment stride-1 layouts whose rows are an ODD number of 8-byte bank
    // pairs apart: 12 or 20 halves).  Lanes 0-31 of a fragment read are 16 output steps x 2 k groups; with the k groups 16
    // bytes apart every such read is a 2-way conflict (p * d = 4 mod 64 banks has a solution |d| <= 15 for p = 6, 10: SQ_LDS_
    // BANK_CONFLICT = 48 % of the LDS cycles, the LDS array 70-75 % busy); 128 bytes apart it is conflict-free -- the 16 rows
    // fill every other bank pair, the second k group the rest.  So inside a block of 128 slab halves (4 K chunks) MFMA j takes
    // its k groups 0..3 from the 16-byte pieces j, j + 8, j + 4, j + 12; the weights are packed with the same permutation
    // (pack_gconv_mfma_kernel).  Chunks behind the last whole block of four keep the plain order.
    static constexpr bool PERM = NSEG == 1 && STRIDE == 1 && pitch(0) % 8 == 4 && GC_KPERM;
    static constexpr int NKP = PERM ? (nk(0) / 4) * 4 : 0;
    static constexpr int rows(int s, int tt) { return STRIDE == 2 ? tt + ntap(s) - 1 : tt + KS - 1; }
    static constexpr int slab(int s, int tt) {       // halves, incl. the K round-up the last output step reads past its window
        if (s >= NSEG) return 0;
        const int over = 32 * nk(s) - ntap(s) * pitch(s);
        return (rows(s, tt) * pitch(s) + (over > 0 ? over : 0) + 7) & ~7;
    }
};

// runtime mirror for the weight packer
struct GcPackDesc {
    int nseg, pitch[2], ntap[2], nch[2], choff[2], tap0[2], nk[2], tapstep, cig, cog, mt_n, groups;
    int perm_nk;      // K chunks of segment 0 in the permuted order (GcLayout::PERM)
};
template <int CIG, int STRIDE>
static GcPackDesc make_pack_desc(int cog, int groups) {
    using LY = GcLayout<CIG, STRIDE>;
    GcPackDesc d = {};
    d.nseg = LY::NSEG;
    for (int s2 = 0; s2 < 2; ++s2) {
        d.pitch[s2] = LY::pitch(s2); d.ntap[s2] = LY::ntap(s2); d.nch[s2] = LY::nch(s2);
        d.choff[s2] = LY::choff(s2); d.tap0[s2] = LY::tap0(s2); d.nk[s2] = LY::nk(s2);
    }
    d.tapstep = STRIDE; d.cig = CIG; d.cog = cog; d.mt_n = (cog + 15) / 16; d.groups = groups;
    d.perm_nk = LY::NKP;
    return d;
}
// the (C_in/G, C_out/G, stride) combinations with a kernel; false: none
static bool gconv_mfma_desc(int cig, int cog, int stride, int groups, GcPackDesc& d) {
    if (groups <= 0 || groups % 4) return false;
    if (stride == 1 && cig == cog) {
        if (cig == 10) { d = make_pack_desc<10, 1>(cog, groups); return true; }
        if (cig == 14) { d = make_pack_desc<14, 1>(cog, groups); return true; }
        if (cig == 18) { d = make_pack_desc<18, 1>(cog, groups); return true; }
    }
    if (stride == 2) {
        if (cig == 10 && cog == 14) { d = make_pack_desc<10, 2>(cog, groups); return true; }
        if (cig == 14 && cog == 18) { d = make_pack_desc<14, 2>(cog, groups); return true; }
    }
    return false;
}

// `p2` = p + 4 halves through a pointer the compiler cannot relate to `p` (8-byte-aligned pitches only): the fragment is then
// read as TWO ds_read_b64 (2 LDS cycles each).  Left to itself hipcc merges the pair into one ds_read2_b64, which the LDS
// serves at half the rate (8-16 cycles per wave instruction, MI355X_MICROARCH.md section LDS) -- with 20- or 12-halves rows the
// LDS array was ~70 % busy at ~40 % matrix-pipe utilisation (profiles/r2_pmc_inst_all_kernels.txt, SQ_LDS_IDX_ACTIVE).
template <int P>
__device__ __forceinline__ f16x8 gconv_frag(const _Float16* p, const _Float16* p2) {
    if (P % 8 == 0) return *reinterpret_cast<const f16x8*>(p);
#ifdef GC_MERGED_READS      // (layout ablation build, scripts/build_ablation.sh: one 8-byte-aligned 16-byte read = ds_read2_b64)
    if (P % 4 == 0) return *reinterpret_cast<const f16x8a8*>(p);
#endif
    if (P % 4 == 0) {
        typedef _Float16 f16x4a8 __attribute__((ext_vector_type(4), aligned(8)));
        const f16x4a8 a = *reinterpret_cast<const f16x4a8*>(p), b = *reinterpret_cast<const f16x4a8*>(p2);
        return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    }
    return *reinterpret_cast<const f16x8u*>(p);
}

// (ablation build -DGC_TIMELINE: lane 0 of every wave of the first 4096 workgroups stamps the 100 MHz wall clock at the phase
//  boundaries; scripts/gconv_timeline.py reads them through tal_debug_gconv_timeline.  Not part of the product library.)
#ifdef GC_TIMELINE
__device__ unsigned long long g_gc_timeline[8 * 4 * 4096];
#define GC_STAMP(i)                                                                                   \
    do {                                                                                              \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                             \
            g_gc_timeline[((i) * 4 + (threadIdx.x >> 6)) * 4096 + blockIdx.x] = wall_clock64();       \
    } while (0)
#else
#define GC_STAMP(i)
#endif

// Which (batch item, time tile, group block) a workgroup takes.  n_tt > 0: a 1-D grid in the XCD-aware order -- group blocks
// fastest, an XCD walks whole time tiles: a row of the activations is C * 4 bytes shared by all group blocks, and a workgroup's
// 36-56 channels are 144-224 bytes that start and end inside 128-byte lines, so neighbouring group blocks READ and WRITE the same
// lines.  On the plain (time tile, group block, item) grid they sit on different XCDs (block b runs on XCD b % 8): every
// boundary line is fetched by two L2s and leaves two L2s as a partial write (measured 1.85x the input fetched, 1.35x the output
// written, profiles/r3_pmc_traffic_all_kernels.txt).  n_tt == 0: the plain 3-D grid (option gconv_grid_xyz, ablation).
#define GC_BLOCK_INDEX(n_tt, n_gb)                                                           \
    int b, g0;                                                                               \
    int64_t t0;                                                                              \
    if (n_tt > 0) {                                                                          \
        const unsigned L = xcd_logical_block(gridDim.x, blockIdx.x);                         \
        const unsigned gbi = L % (unsigned)n_gb, rest = L / (unsigned)n_gb;                  \
        g0 = (int)gbi * GB;                                                                  \
        t0 = (int64_t)(rest % (unsigned)n_tt) * TT;                                          \
        b = (int)(rest / (unsigned)n_tt);                                                    \
    } else {                                                                                 \
        b = blockIdx.z;                                                                      \
        g0 = blockIdx.y * GB;                                                                \
        t0 = (int64_t)blockIdx.x * TT;                                                       \
    }

// FROMC1 (round 6; 10 channels per group, split form out): the slab is not loaded but COMPUTED -- the workgroup runs the first resize
// conv of the encoder (1 mel bin -> 10 channels per group, stride 2, no padding: tal/asr/models.py:363-364) for its groups and its
// tile's rows (halo included) straight into the slab, with gconv_s2_c1_kernel's arithmetic (same fmaf chain, same mean fold, same
// split), and the TDSBlock conv follows from there.  The stage's first activation (576 MB per hour of audio) is never written or
// read, and the resize conv's launch is gone; results are bit-identical to the two launches.
struct GcC1 {
    const float* mel;     // [B][T_mel][C_mel] log-mel (before its mean subtraction when `mean` is given)
    const float* w1;      // packed resize-conv weights [G][1][21][10]
    const float* b1;      // [G * 10]
    const float* mean;    // device scalar or NULL
    int64_t T_mel;
    int C_mel;
};

// SPLIT: 0 = fp32 output, 1 = fp32 output + its hi / lo split, 2 = the split form only.  XSPLIT: the input is in the split
// form (per row and 32-channel block: 32 hi halves, 32 lo halves; same bytes as fp32) -- the slab is then filled without
// any conversion arithmetic and the TDSBlock residual is rebuilt from the slab as hi + lo * 2^-11.
template <int CIG, int COG, int STRIDE, bool RESID, int GB, int TT, int SPLIT, bool XSPLIT = false, bool FROMC1 = false>
__global__ __launch_bounds__(256, GC_LB(CIG, STRIDE)) void gconv_mfma_kernel(const float* __restrict__ x, const _Float16* __restrict__ wfrag,
                                                           const float* __restrict__ bias, float alpha, float* __restrict__ y,
                                                           _Float16* __restrict__ ysplit, int64_t T_in, int64_t T_out, int C_in,
                                                           int C_out, int* __restrict__ range_flag, int n_tt, int n_gb, const GcC1 c1) {
    using LY = GcLayout<CIG, STRIDE>;
    constexpr int NKS = LY::NKS, NK0 = LY::nk(0), MT = (COG + 15) / 16;
    constexpr int P0 = LY::pitch(0), P1 = LY::pitch(1);
    constexpr int PADT = RESID ? KS / 2 : 0;
    // STAGED: the output tile goes through LDS and leaves as 16-byte (8-byte) pieces.  Rows below (tb + 1) * 16 of a group's
    // slab are dead once block tb has its operands, so the block's output halves are written over them (channels 0-15 into
    // the row of segment 0, 16-17 into the row of segment 1); with one wave per group that needs no synchronisation, with two
    // (MT == 2) one workgroup barrier per block.  Why: a store instruction costs by the 128-byte lines it touches, not by
    // its bytes -- the direct path (lane = time step) pays 16 lines per instruction, 4-6 instructions per block.
    constexpr bool STAGED = SPLIT == 2 && GB * MT == 4 && (GB * COG) % 4 == 0 && P0 % 4 == 0 && P0 >= (COG < 16 ? 4 * ((COG + 3) / 4) : 16) &&
                            (MT == 1 || (MT == 2 && ((LY::NSEG == 2 && P1 >= 4) || (LY::NSEG == 1 && P0 >= 4 * ((COG + 3) / 4)))));
    constexpr int TIN = (TT - 1) * STRIDE + KS;                     // input rows a tile of TT outputs reads
    constexpr int SL0 = LY::slab(0, TT), SL1 = LY::slab(1, TT), GS = SL0 + SL1;   // halves per group and (hi | lo) array
    constexpr int CH = GB * CIG, CH4 = CH / 4;
    constexpr int RPP = 256 / CH4;                                  // input rows per pass of the slab load
    static_assert(CH % 4 == 0 && CIG % 2 == 0 && TT % 64 == 0 && (!RESID || (CIG == COG && STRIDE == 1)), "shape");
    extern __shared__ __attribute__((aligned(16))) _Float16 slab[];   // [2 (hi, lo)][GB][GS]
    _Float16* s_hi = slab;
    _Float16* s_lo = slab + GB * GS;

    GC_BLOCK_INDEX(n_tt, n_gb)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = wave_id();
    const float* xb = x + (int64_t)b * T_in * C_in;
    float* yb = y + (int64_t)b * T_out * C_out;
    GC_STAMP(0);

    if constexpr (FROMC1) {
        static_assert(CIG == 10 && COG == 10 && STRIDE == 1 && RESID && GB == 4 && XSPLIT && SPLIT == 2 && LY::NSEG == 1 && TIN % 4 == 0 && P0 % 2 == 0, "shape");
        // ---- the slab from the log-mel: stage the tile's mel rows ([row][4 bins], 16-byte loads), then thread = (channel of the
        // workgroup, row phase): a channel's 21 weights + bias in registers, four consecutive rows share their 27 samples ----
        constexpr int NR = 2 * (TIN - 1) + KS;                       // mel rows behind TIN rows of the stage
        float* xs = reinterpret_cast<float*>(slab + 2 * GB * GS);    // [NR][GB] floats behind the slabs
        const float* mb = c1.mel + (int64_t)b * c1.T_mel * c1.C_mel + g0;
        const int64_t m0 = 2 * (t0 - PADT);
        for (int i = tid; i < NR; i += 256) {
            int64_t t = m0 + i;
            t = t < 0 ? 0 : (t < c1.T_mel ? t : c1.T_mel - 1);       // (clamped rows only feed rows outside [0, T_in): forced to zero below)
            *reinterpret_cast<f32x4*>(xs + i * GB) = *reinterpret_cast<const f32x4*>(mb + t * c1.C_mel);
        }
        __syncthreads();
        float amax = 0.f;
        // A thread owns a channel PAIR (packed fp32 multiply-adds: on this chip a SIMD's vector ALU time is what the conv phase
        // behind competes for -- the first version, one channel per thread, made the launch 0.27 ms slower than the two it replaces)
        // and one of 12 row phases; per element the fmaf chain of gconv_s2_c1_kernel, so the halves are bit-identical to its output.
        constexpr int NPAIR = GB * CIG / 2, NPH = 256 / NPAIR;          // 20 pairs x 12 phases = 240 threads
        if (tid < NPAIR * NPH) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const int pr = tid % NPAIR, ph = tid / NPAIR;
            const int gl = pr / (CIG / 2), ci = 2 * (pr - gl * (CIG / 2)), g = g0 + gl;
            f32x2 wk[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) wk[k] = *reinterpret_cast<const f32x2*>(c1.w1 + ((int64_t)g * KS + k) * COG + ci);
            f32x2 bv = *reinterpret_cast<const f32x2*>(c1.b1 + g * COG + ci);
            if (c1.mean) {
                f32x2 wsum = wk[0];
#pragma unroll
                for (int k = 1; k < KS; ++k) wsum += wk[k];
                const float m = -c1.mean[0];
                bv[0] = fmaf(m, wsum[0], bv[0]);
                bv[1] = fmaf(m, wsum[1], bv[1]);
            }
            const float* xc = xs + gl;
            _Float16* dh = s_hi + gl * GS + ci;
            _Float16* dl = s_lo + gl * GS + ci;
            for (int rq = ph; rq < TIN / 4; rq += NPH) {
                const int r = 4 * rq;
                float xv[KS + 6];
#pragma unroll
                for (int i = 0; i < KS + 6; ++i) xv[i] = xc[(2 * r + i) * GB];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x2 acc = bv;
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        const f32x2 xx = {xv[2 * q + k], xv[2 * q + k]};
                        acc = __builtin_elementwise_fma(wk[k], xx, acc);
                    }
                    const int64_t tg = t0 - PADT + r + q;
                    const bool inside = tg >= 0 && tg < T_in;         // outside: the TDSBlock conv's zero padding
                    _Float16 h0, l0, h1, l1;
                    split_f16x3(acc[0], h0, l0);
                    split_f16x3(acc[1], h1, l1);
                    if (inside) amax = fmaxf(amax, fmaxf(fabsf(acc[0]), fabsf(acc[1])));
                    const f16x2 zz = {(_Float16)0.f, (_Float16)0.f};
                    const f16x2 hh = {h0, h1}, ll = {l0, l1};
                    *reinterpret_cast<f16x2*>(dh + (r + q) * P0) = inside ? hh : zz;
                    *reinterpret_cast<f16x2*>(dl + (r + q) * P0) = inside ? ll : zz;
                    if (ci == CIG - 2) {                              // this row's pad columns (finite bytes under zero weights)
#pragma unroll
                        for (int j = 2; j + 1 < P0 - CIG + 2; j += 2) {
                            *reinterpret_cast<f16x2*>(dh + (r + q) * P0 + j) = zz;
                            *reinterpret_cast<f16x2*>(dl + (r + q) * P0 + j) = zz;
                        }
                    }
                }
            }
        }
        note_range(amax, range_flag);
        const f16x2 z2 = {(_Float16)0.f, (_Float16)0.f};
        constexpr int tail2 = (SL0 - LY::rows(0, TT) * P0) / 2;
        if (tail2 > 0)
            for (int i = tid; i < 2 * GB * tail2; i += 256)
                *reinterpret_cast<f16x2*>(slab + (i / tail2) * GS + LY::rows(0, TT) * P0 + 2 * (i % tail2)) = z2;
    } else
    // ---- slab load: thread = (input row inside a pass, 16-byte column piece): 16-byte global loads (GB*CIG contiguous
    // floats per row), all passes in flight, split, 4-byte LDS stores ----
    {
        const int r0 = tid / CH4, c4 = tid - r0 * CH4;
        const bool active = r0 < RPP;
        int so[2], sp[2];       // per channel pair: constant part of the LDS offset, row pitch
        int npad[2];            // pad pairs behind this pair: the thread that stores a segment's LAST channel pair of a row also
                                // zeroes the row's pad columns (finite bytes under zero weights) -- no separate pass over the slab
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ch = c4 * 4 + 2 * q, gl = ch / CIG, ci = ch - gl * CIG;
            const int seg = (STRIDE == 1 && ci >= LY::nch(0)) ? 1 : 0;          // channel split (18 channels per group)
            so[q] = gl * GS + (seg ? SL0 + ci - LY::nch(0) : ci);
            sp[q] = seg ? P1 : P0;
            const int cis = seg ? ci - LY::nch(0) : ci, nc = LY::nch(seg);
            npad[q] = cis + 2 == nc ? (sp[q] - nc) / 2 : 0;
        }
        constexpr int MAXPAD = (P0 - LY::nch(0)) / 2 > (LY::NSEG == 2 ? (P1 - LY::nch(1)) / 2 : 0) ? (P0 - LY::nch(0)) / 2
                                                                                                 : (LY::NSEG == 2 ? (P1 - LY::nch(1)) / 2 : 0);
        const unsigned zero_pair = 0u;
        auto pad_row = [&](int q, int off) {          // off: LDS offset (halves) of the pair just stored
#pragma unroll
            for (int j = 0; j < MAXPAD; ++j)
                if (j < npad[q]) {
                    *reinterpret_cast<unsigned*>(s_hi + off + 2 + 2 * j) = zero_pair;
                    *reinterpret_cast<unsigned*>(s_lo + off + 2 + 2 * j) = zero_pair;
                }
        };
        // Buffer loads over this batch item ([T_in, C_in] floats): a row before the first or past the last one is out of
        // the descriptor's range and reads as zeros -- the conv's zero padding without clamps or selects.  (The pass offset
        // has to travel in the lane offset: a scalar offset is not part of the range check.)
        __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)(T_in * C_in * 4), 0x00020000);
        const int64_t row0 = t0 * STRIDE - PADT + r0;
        // (32-bit byte offset: launch_* checks that a batch item stays below 2 GiB; a negative row wraps to an offset past the range)
        const int voff = (int)((row0 * C_in + g0 * CIG + (active ? c4 * 4 : 0)) * 4);
        constexpr int NPASS = (TIN + RPP - 1) / RPP, UNR = NPASS > 20 ? (NPASS + 1) / 2 : NPASS;
        float amax = 0.f;      // fp16-range guard: the largest |x| this thread turns into halves
        if constexpr (XSPLIT) {
            // split input: channels c .. c + 3 (c % 4 == 0, so never across a 32-channel block) are 8 bytes of hi halves at
            // block * 128 + (c % 32) * 2 of the row and 8 bytes of lo halves 64 bytes further
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const int cgl = g0 * CIG + (active ? c4 * 4 : 0);
            const int voffs = (int)(row0 * C_in * 4) + (cgl >> 5) * 128 + (cgl & 31) * 2;
            for (int p0 = 0; p0 < NPASS; p0 += UNR) {
                u32x2 vh[UNR], vl[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    vh[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, voffs + (p0 + u) * RPP * C_in * 4, 0, 0);
                    vl[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, voffs + 64 + (p0 + u) * RPP * C_in * 4, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int ti = r0 + (p0 + u) * RPP;
                    const int row = STRIDE == 2 ? ti >> 1 : ti;
                    const int tbase = (STRIDE == 2 && (ti & 1)) ? SL0 : 0;
                    if (active && ti < TIN) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            *reinterpret_cast<unsigned*>(s_hi + so[q] + tbase + row * sp[q]) = vh[u][q];
                            *reinterpret_cast<unsigned*>(s_lo + so[q] + tbase + row * sp[q]) = vl[u][q];
                            if constexpr (MAXPAD > 0) pad_row(q, so[q] + tbase + row * sp[q]);
                        }
                    }
                }
            }
        } else
        for (int p0 = 0; p0 < NPASS; p0 += UNR) {
            f32x4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff + (p0 + u) * RPP * C_in * 4, 0, 0));
#pragma unroll
            for (int u = 0; u < UNR; ++u) amax = amax4(amax, v[u]);
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ti = r0 + (p0 + u) * RPP;
                // stride 2: even input rows -> segment 0, odd rows -> segment 1, row ti / 2 of its slab
                const int row = STRIDE == 2 ? ti >> 1 : ti;
                const int tbase = (STRIDE == 2 && (ti & 1)) ? SL0 : 0;
                if (active && ti < TIN) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // hi = fp16(x), lo = fp16((x - hi) * 2^11) on a channel pair (x clamped to the fp16 range first:
                        // |lo| <= 2^15 then needs no clamp of its own; identical to split_f16x3 for |x| <= 65504)
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        const f32x2 xc2 = {__builtin_amdgcn_fmed3f(v[u][2 * q], -65504.f, 65504.f),
                                           __builtin_amdgcn_fmed3f(v[u][2 * q + 1], -65504.f, 65504.f)};
                        const f16x2 hh = __builtin_convertvector(xc2, f16x2);
                        const f32x2 hf = __builtin_convertvector(hh, f32x2);
                        const f16x2 ll = __builtin_convertvector((xc2 - hf) * 2048.f, f16x2);
                        *reinterpret_cast<f16x2*>(s_hi + so[q] + tbase + row * sp[q]) = hh;
                        *reinterpret_cast<f16x2*>(s_lo + so[q] + tbase + row * sp[q]) = ll;
                        if constexpr (MAXPAD > 0) pad_row(q, so[q] + tbase + row * sp[q]);
                    }
                }
            }
        }
        note_range(amax, range_flag);
        // zeros behind the last row of each segment (the K round-up the last output steps read past their window)
        const f16x2 z2 = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int sg = 0; sg < LY::NSEG; ++sg) {
            const int pt = sg ? P1 : P0, rows = LY::rows(sg, TT), sl = sg ? SL1 : SL0, base = sg ? SL0 : 0;
            const int tail2 = (sl - rows * pt) / 2;
            if (tail2 > 0)
                for (int i = tid; i < 2 * GB * tail2; i += 256)
                    *reinterpret_cast<f16x2*>(slab + (i / tail2) * GS + base + rows * pt + 2 * (i % tail2)) = z2;
        }
    }
    GC_STAMP(1);
    __syncthreads();
    GC_STAMP(2);

    // ---- units = (group, 16-channel M tile); a wave owns one unit (or a time slice of one) at a time ----
    constexpr int NU = GB * MT, NB = TT / 16;
    constexpr int UPW = NU >= 4 ? NU / 4 : 1;         // units per wave
    constexpr int WPU = NU >= 4 ? 1 : 4 / NU;         // waves per unit
    constexpr int NBW = NB / WPU;
    static_assert(NU == 1 || NU == 2 || NU % 4 == 0, "units");
    const int col = lane & 15, kg = lane >> 4;
    for (int uu = 0; uu < UPW; ++uu) {
        const int u = NU >= 4 ? w + 4 * uu : w / WPU;
        const int part = NU >= 4 ? 0 : w % WPU;
        const int gl = u / MT, mt = u - gl * MT, g = g0 + gl;
        f16x8 wh[NKS], wl[NKS];
        const f16x8* wf = reinterpret_cast<const f16x8*>(wfrag) + (int64_t)(g * MT + mt) * (NKS * 2 * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            wh[ks] = wf[(ks * 2 + 0) * 64];
            wl[ks] = wf[(ks * 2 + 1) * 64];
        }
        const int ch0 = mt * 16 + 4 * kg;
        const int nvalid = COG - ch0;                 // >= 4: four channels, 2: two, <= 0: none
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nvalid) bv[i] = bias[g * COG + ch0 + i];
        const int tbeg = part * NBW, tend = (part + 1) * NBW;
        asm volatile("" :: "v"(wh[NKS - 1]), "v"(wl[NKS - 1]));
        GC_STAMP(4);
        // fragment of K chunk c for output column (16 tb + col): segment base + (16 tb + col) * pitch + 8 kg + 32 ks
        // (LY::PERM: k group kg of the chunks below LY::NKP sits 64 (kg & 1) + 32 (kg >> 1) halves into its block of 128)
        constexpr int NKP = LY::NKP;
        const int kgo = LY::PERM ? 64 * (kg & 1) + 32 * (kg >> 1) : 8 * kg;
        const _Float16* hs = s_hi + gl * GS + kgo;
        const _Float16* ls = s_lo + gl * GS + kgo;
        int four = 4;                        // (opaque to the compiler: hs2 / ls2 become separate base registers, still LDS pointers)
        asm volatile("" : "+v"(four));
        const _Float16* hs2 = hs + four;     // second halves of the fragments (gconv_frag)
        const _Float16* ls2 = ls + four;
        // chunks behind the last whole block of four: plain order
        const _Float16* hst = s_hi + gl * GS + 8 * kg;
        const _Float16* lst = s_lo + gl * GS + 8 * kg;
        const _Float16* hst2 = hst + four;
        const _Float16* lst2 = lst + four;
        int boff0 = (tbeg * 16 + col) * P0, boff1 = SL0 + (tbeg * 16 + col) * P1;
        auto frag_off = [&](int c, int b0, int b1) {
            return c < NKP ? b0 + 128 * (c >> 2) + 8 * (c & 3) : (c < NK0 ? b0 + 32 * c : b1 + 32 * (c - NK0));
        };
        auto frag = [&](const _Float16* base, int c, int b0, int b1) {
            const bool is_hi = base == hs;
            const bool plain = LY::PERM && c >= NKP;
            const _Float16* b1p = plain ? (is_hi ? hst : lst) : base;
            const _Float16* b2p = plain ? (is_hi ? hst2 : lst2) : (is_hi ? hs2 : ls2);
            return c < NK0 ? gconv_frag<P0>(b1p + frag_off(c, b0, b1), b2p + frag_off(c, b0, b1))
                           : gconv_frag<P1>(b1p + frag_off(c, b0, b1), b2p + frag_off(c, b0, b1));
        };
        // operand fragments are read PF K chunks ahead of their MFMAs (a ring of PF + 1 register slots; the first PF chunks of the
        // next block are read under the last MFMAs of the current one and carried across the epilogue)
        constexpr int PF = GC_PF, RING = PF + 1;
        static_assert(PF >= 1 && PF < NKS, "prefetch depth");
        f16x8 ch[PF], cl[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            ch[q] = frag(hs, q, boff0, boff1);
            cl[q] = frag(ls, q, boff0, boff1);
        }
        // TDSBlock residual: x of block tb is fetched XD blocks ahead; the block loop is unrolled by XD + 1 so the ring of
        // in-flight registers is indexed statically.  Branch-free: every lane loads 16 bytes; a lane with two valid
        // channels loads from two floats earlier and keeps the upper half, a lane with none re-reads the group's first
        // channels (all addresses stay inside the row); the choice is applied at the point of use, so no wait sits behind
        // the load.
        constexpr int XD = 3;
        static_assert(NBW % (XD + 1) == 0, "blocks per wave");
        const int cbase = g * COG + ch0;
        const float* xcol = xb + cbase + (nvalid >= 4 ? 0 : (nvalid == 2 ? -2 : -ch0));
        auto load_x = [&](int tb) {
            int64_t t = t0 + tb * 16 + col;
            t = t < T_out ? t : T_out - 1;
            return *reinterpret_cast<const f32x4u*>(xcol + t * C_in);
        };
        f32x4 xr[XD + 1];
        if (RESID && !XSPLIT) {
#pragma unroll
            for (int j = 0; j < XD; ++j) xr[j] = load_x(tbeg + j);
        }
        // XSPLIT: the residual x[t][ch0 .. ch0 + 3] sits in this group's slab (row t - t0 + PADT) as hi / lo halves
        const int rseg = (STRIDE == 1 && LY::NSEG == 2 && ch0 >= LY::nch(0)) ? 1 : 0;
        const _Float16* rh = s_hi + gl * GS + (rseg ? SL0 + ch0 - LY::nch(0) : ch0);
        const _Float16* rl = s_lo + gl * GS + (rseg ? SL0 + ch0 - LY::nch(0) : ch0);
        const int rp = rseg ? P1 : P0;
        constexpr int r01 = 0;
        const int r23 = nvalid >= 4 ? 2 : 0;   // (a lane without valid channels reads initialised slab bytes it never uses)
        // rows of this tile that exist, seen from this lane: block tb holds one iff 16 tb < rows_left
        const int rows_left = (int)(T_out - t0 < (int64_t)TT ? T_out - t0 : (int64_t)TT) - col;
        float am = 0.f;        // fp16-range guard: the largest |y| this lane stores in the split form
        for (int tb0 = tbeg; tb0 < tend; tb0 += XD + 1) {
#pragma unroll
            for (int j = 0; j <= XD; ++j) {
                const int tb = tb0 + j;
                if (RESID && !XSPLIT) xr[(j + XD) % (XD + 1)] = load_x(tb + XD < tend ? tb + XD : tb);
                f32x4 acc = bv, ax1 = {0.f, 0.f, 0.f, 0.f}, ax2 = {0.f, 0.f, 0.f, 0.f};
                const bool more = tb + 1 < tend;
                const int nb0 = more ? boff0 + 16 * P0 : boff0, nb1 = more ? boff1 + 16 * P1 : boff1;
                f16x8 bh[RING], bl[RING];
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    bh[q] = ch[q];
                    bl[q] = cl[q];
                }
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    // fragments of K chunk ks + PF (of the next block at the end) are read before the MFMAs of chunk ks
                    const int pk = ks + PF;
                    if (pk < NKS) {
                        bh[pk % RING] = frag(hs, pk, boff0, boff1);
                        bl[pk % RING] = frag(ls, pk, boff0, boff1);
                    } else {
                        bh[pk % RING] = frag(hs, pk - NKS, nb0, nb1);
                        bl[pk % RING] = frag(ls, pk - NKS, nb0, nb1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], bh[ks % RING], acc, 0, 0, 0);
                    ax1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], bl[ks % RING], ax1, 0, 0, 0);
                    ax2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ks], bh[ks % RING], ax2, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    ch[q] = bh[(NKS + q) % RING];
                    cl[q] = bl[(NKS + q) % RING];
                }
                boff0 = nb0;
                boff1 = nb1;
                // ---- epilogue (every vector instruction here is time taken from the matrix pipe: kept to ~40) ----
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 s11 = {1.0f / 2048.0f, 1.0f / 2048.0f}, z0 = {0.f, 0.f}, al2 = {alpha, alpha};
                f32x2 oa = {acc[0], acc[1]}, ob = {acc[2], acc[3]};
                {
                    const f32x2 xa = {ax1[0] + ax2[0], ax1[1] + ax2[1]}, xb2 = {ax1[2] + ax2[2], ax1[3] + ax2[3]};
                    oa = xa * s11 + oa;
                    ob = xb2 * s11 + ob;
                }
                if (RESID && XSPLIT) {
                    // branch-free: every lane reads two channel pairs (a lane with fewer valid channels re-reads valid
                    // ones, r01 / r23 below), so its unused results stay ordinary finite activations
                    const int ro = (tb * 16 + col + PADT) * rp;
                    const f16x2 h01 = *reinterpret_cast<const f16x2*>(rh + ro + r01), l01 = *reinterpret_cast<const f16x2*>(rl + ro + r01);
                    const f16x2 h23 = *reinterpret_cast<const f16x2*>(rh + ro + r23), l23 = *reinterpret_cast<const f16x2*>(rl + ro + r23);
                    const f32x2 xa = __builtin_convertvector(l01, f32x2) * s11 + __builtin_convertvector(h01, f32x2);
                    const f32x2 xb2 = __builtin_convertvector(l23, f32x2) * s11 + __builtin_convertvector(h23, f32x2);
                    oa = al2 * __builtin_elementwise_max(oa, z0) + xa;
                    ob = al2 * __builtin_elementwise_max(ob, z0) + xb2;
                } else if (RESID) {
                    const f32x4 xs2 = {xr[j][2], xr[j][3], 0.f, 0.f};
                    const f32x4 xv = nvalid >= 4 ? xr[j] : xs2;
                    const f32x2 xa = {xv[0], xv[1]}, xb2 = {xv[2], xv[3]};
                    oa = al2 * __builtin_elementwise_max(oa, z0) + xa;
                    ob = al2 * __builtin_elementwise_max(ob, z0) + xb2;
                }
                if constexpr (STAGED) {
                    if constexpr (MT > 1) __syncthreads();     // the partner wave has read its operands of this block
                    if (nvalid > 0) {
                        float am2 = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(oa[0]), __builtin_fabsf(oa[1])));
                        if (nvalid >= 4) am2 = __builtin_fmaxf(am2, __builtin_fmaxf(__builtin_fabsf(ob[0]), __builtin_fabsf(ob[1])));
                        am = tb * 16 < rows_left ? am2 : am;       // (rows past the end of the sequence are never stored)
                        f16x2p h01p, l01p, h23p, l23p;
                        if (range_flag) {
                            split_f16x3_pair(oa[0], oa[1], h01p, l01p);
                            split_f16x3_pair(ob[0], ob[1], h23p, l23p);
                        } else {
                            _Float16 hh[4], ll[4];
                            split_f16x3(oa[0], hh[0], ll[0]); split_f16x3(oa[1], hh[1], ll[1]);
                            split_f16x3(ob[0], hh[2], ll[2]); split_f16x3(ob[1], hh[3], ll[3]);
                            h01p = {hh[0], hh[1]}; l01p = {ll[0], ll[1]}; h23p = {hh[2], hh[3]}; l23p = {ll[2], ll[3]};
                        }
                        typedef _Float16 f16x4a8 __attribute__((ext_vector_type(4), aligned(8)));
                        const int so2 = gl * GS + ((mt == 0 || LY::NSEG == 1) ? (tb * 16 + col) * P0 + mt * 16 + 4 * kg : SL0 + (tb * 16 + col) * P1 + 4 * kg);
                        *reinterpret_cast<f16x4a8*>(s_hi + so2) = f16x4a8{h01p[0], h01p[1], h23p[0], h23p[1]};
                        *reinterpret_cast<f16x4a8*>(s_lo + so2) = f16x4a8{l01p[0], l01p[1], l23p[0], l23p[1]};
                    }
                } else
                if (tb * 16 < rows_left && nvalid > 0) {
                    const int64_t t = t0 + tb * 16 + col;
                    if (SPLIT != 2) {
                        float* yp = yb + t * C_out + cbase;
                        const f32x4 o = {oa[0], oa[1], ob[0], ob[1]};
                        if (nvalid >= 4) *reinterpret_cast<f32x4u*>(yp) = o;
                        else *reinterpret_cast<f32x2u*>(yp) = f32x2u{oa[0], oa[1]};
                    }
                    if (SPLIT) {
                        // the same values as hi / lo halves in the dense layers' operand geometry: per row and 32-channel
                        // block 32 hi halves, then 32 lo halves (channels cbase .. cbase + 3; cbase is even)
                        // (under the range guard: no clamps, packed conversions)
                        am = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(oa[0]), __builtin_fabsf(oa[1])));
                        if (nvalid >= 4) am = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(ob[0]), __builtin_fabsf(ob[1])));
                        f16x2p h01p, l01p, h23p, l23p;
                        if (range_flag) {
                            split_f16x3_pair(oa[0], oa[1], h01p, l01p);
                            split_f16x3_pair(ob[0], ob[1], h23p, l23p);
                        } else {      // unguarded call: clamped halves (a value beyond the fp16 range degrades instead of becoming inf)
                            _Float16 hh[4], ll[4];
                            split_f16x3(oa[0], hh[0], ll[0]); split_f16x3(oa[1], hh[1], ll[1]);
                            split_f16x3(ob[0], hh[2], ll[2]); split_f16x3(ob[1], hh[3], ll[3]);
                            h01p = {hh[0], hh[1]}; l01p = {ll[0], ll[1]}; h23p = {hh[2], hh[3]}; l23p = {ll[2], ll[3]};
                        }
                        _Float16* sp = ysplit + (((int64_t)b * T_out + t) * (C_out >> 5) + (cbase >> 5)) * 64 + (cbase & 31);
                        const f16x2 h01 = {h01p[0], h01p[1]}, l01 = {l01p[0], l01p[1]}, h23 = {h23p[0], h23p[1]}, l23 = {l23p[0], l23p[1]};
                        if (nvalid >= 4 && (cbase & 31) != 30) {
                            const f16x4u h4 = {h01[0], h01[1], h23[0], h23[1]}, l4 = {l01[0], l01[1], l23[0], l23[1]};
                            *reinterpret_cast<f16x4u*>(sp) = h4;
                            *reinterpret_cast<f16x4u*>(sp + 32) = l4;
                        } else {
                            *reinterpret_cast<f16x2*>(sp) = h01;
                            *reinterpret_cast<f16x2*>(sp + 32) = l01;
                            if (nvalid >= 4) {        // the second pair opens the next 32-channel block
                                *reinterpret_cast<f16x2*>(sp + 34) = h23;
                                *reinterpret_cast<f16x2*>(sp + 66) = l23;
                            }
                        }
                    }
                }
            }
        }
        if (SPLIT) note_range(am, range_flag);
    }
    GC_STAMP(5);
    if constexpr (STAGED) {
        // ---- the tile leaves LDS as pieces of PW words (8 channels from a multiple of 8, or 4 from a multiple of 4 -- g0 * COG
        // is one -- so a piece never crosses a 32-channel block).  thread = (row inside a pass, piece): the lanes of a store
        // instruction walk along rows, a row's pieces sit in its 2-3 consecutive 128-byte lines (one row per lane, 64 lines per
        // instruction, made this phase 23-36 % of the kernel) ----
        __syncthreads();
        GC_STAMP(6);
        constexpr int CW = GB * COG, PW = CW % 8 == 0 ? 4 : 2, NPC = CW / (2 * PW), WPG = COG / 2, NPR = 2 * NPC;
        constexpr int RP = 256 / NPR, NPO = (TT + RP - 1) / RP;
        typedef unsigned u32xp __attribute__((ext_vector_type(PW)));
        const int nrows = (int)(T_out - t0 < (int64_t)TT ? T_out - t0 : (int64_t)TT);
        const int r = tid / NPR, q = tid - r * NPR;
        const int a = q / NPC, p = q - a * NPC;          // a: 0 = hi halves, 1 = lo halves
        int wo[PW], wp[PW];                              // per word of the piece: LDS offset (halves) in row 0, row pitch
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int idx = PW * p + j, gl = idx / WPG, i = idx - gl * WPG;
            const bool first = i < 8 || (LY::NSEG == 1);      // channels 16, 17 of a two-segment slab sit in segment 1
            wp[j] = first ? P0 : P1;
            wo[j] = (a * GB + gl) * GS + (first ? 2 * i : SL0 + 2 * (i - 8)) + r * wp[j];
        }
        const int c = g0 * COG + 2 * PW * p;
        char* op = reinterpret_cast<char*>(ysplit) + ((int64_t)b * T_out + t0 + r) * C_out * 4 + (c >> 5) * 128 + (c & 31) * 2 + a * 64;
        const int64_t pstep = (int64_t)RP * C_out * 4;
        if (r < RP) {
            constexpr int CHK = 4;
            for (int p0 = 0; p0 < NPO; p0 += CHK) {
                u32xp v[CHK];
#pragma unroll
                for (int u = 0; u < CHK; ++u)
#pragma unroll
                    for (int j = 0; j < PW; ++j)
                        v[u][j] = *reinterpret_cast<const unsigned*>(slab + wo[j] + (p0 + u) * RP * wp[j]);
#pragma unroll
                for (int u = 0; u < CHK; ++u)
                    if ((p0 + u) * RP + r < nrows) *reinterpret_cast<u32xp*>(op + (p0 + u) * pstep) = v[u];
            }
        }
    }
    GC_STAMP(7);
}

// ---------------------------------------------------------------------------------------------
// 18 channels per group (the six TDSBlocks of the last stage), split form in and out: TIME-SHIFT PACKING
// ---------------------------------------------------------------------------------------------
// 18 = 16 + 2: in gconv_mfma_kernel the second 16-row M tile of a group multiplies two output channels -- 42 of the 84 MFMAs
// (and fragment reads) per 16 output steps for a ninth of the outputs.  The Hankel view offers a denser packing: moving a
// channel's weights s slab rows down (k -> k + s P) yields the SAME channel's output s steps later,
//     sum_k W[co][k - s P] slab[t P + k] = out(co, t + s),
// so ONE M tile of rows (channel c in {16, 17}) x (shift s in 0..7), run over K = (21 + 7) rows and columns 8 steps apart,
// produces the two channels for 128 consecutive steps in 18 K chunks: 54 MFMAs per 128 steps instead of 336.
// Work split: a workgroup = 2 groups x TT steps, wave w = (group w / 2, role h = w % 2).
//   phase A  wave (g, h) computes the shifted tile of steps [128 h, 128 h + 128) (TT = 64: h = 0 only, 8 columns), finishes it
//            (bias, fp16x3 combine, ReLU, ReZero residual, hi / lo split) and parks the halves in the slab's pad columns
//            (18, 19 of the 20-halves rows: never read under a non-zero weight) -- before any output overwrites slab rows;
//   phase B  the 16-channel tile as before, wave h over the blocks of its half of the tile -- no barrier inside: a block's halves
//            overwrite its own (dead) slab rows; wave 1 holds back the two blocks whose rows wave 0 still reads.
// Per 16 output steps and group: 42 + 6.75 MFMAs instead of 84, both waves of a group carry equal work, and the 16 workgroup
// barriers of a tile's block loop become one.
// An output's chain of MFMAs depends on its shift s = (t - tile start) % 8 only, and tile starts are multiples of 64: long
// (256) and short (64) tiles give bit-identical results.  Channels 16, 17 differ from gconv_mfma_kernel's in the last bits
// (other grouping of the products into K chunks), channels 0-15 are the same chains.
constexpr int S18_NKA = 18;      // K chunks of the shifted tile: (21 + 7) rows x 20 halves = 560 -> 576
constexpr int S18_SH = 8;        // shifts per channel
// The shifted tile's A fragments are NOT loaded from memory (18 K chunks x hi / lo x 1 KB per wave, mostly zeros, at the 25-60
// GB/s a CU pulls from L2: measured 6.8 us of a 22 us workgroup): a group's two left-over channels are 2 x 420 weights; they sit
// in LDS as a zero-padded line per channel and hi / lo, [S18_WPRE zeros | 420 weights in slab order tap * 20 + channel | zeros],
// and row (c, s), k group kg of chunk ch is the 8 halves at S18_WPRE + 32 ch + 8 kg - 20 s of line c.
constexpr int S18_WPRE = 20 * (S18_SH - 1);                   // 140
constexpr int S18_WL = S18_WPRE + 32 * S18_NKA + 8;           // 724 halves per line, rounded up so that a workgroup's eight lines are
constexpr int S18_WLP = (S18_WL + 255) & ~255;                // a whole number of 256 x 16-byte pieces (768)

template <int TT>
__global__ __launch_bounds__(256, 2) void gconv18_shift_kernel(const float* __restrict__ x, const _Float16* __restrict__ wfrag,
                                                             const _Float16* __restrict__ wshift, const float* __restrict__ bias, float alpha,
                                                             _Float16* __restrict__ ysplit, int64_t T, int C, int* __restrict__ range_flag,
                                                             int n_tt, int n_gb, int nb_total) {
    using LY = GcLayout<18, 1>;
    static_assert(LY::NSEG == 1 && LY::pitch(0) == 20 && LY::PERM && LY::nk(0) == 14, "the shift-packed kernel is built for 40-byte slab rows");
    constexpr int CG = 18, GB = 2, P = 20, NK = 14, NKP = LY::NKP, PADT = KS / 2;
    constexpr int TIN = TT + KS - 1, GS = LY::slab(0, TT);
    static_assert(GS * 2 >= ((TT >= 128 ? TT - 8 : TT - 8) * P + 32 * S18_NKA) * 2, "slab tail");
    constexpr int CH = GB * CG, CH4 = CH / 4, RPP = 256 / CH4;
    extern __shared__ __attribute__((aligned(16))) _Float16 slab[];   // [2 (hi, lo)][GB][GS] | weight lines [GB][2 (channel)][2 (hi, lo)][S18_WLP]
    _Float16* s_hi = slab;
    _Float16* s_lo = slab + GB * GS;
    _Float16* s_wl = slab + 2 * GB * GS;

#ifdef GC_S18_PERSIST
    // persistent form (ablation): two resident workgroups per CU walk the blocks their slots would have been dealt, in the same order
    // (physical block p runs on XCD p % 8; p, p + gridDim.x, ... keep that), so no workgroup is dispatched after the first wave of them
    for (unsigned pb_ = blockIdx.x; pb_ < (unsigned)nb_total; pb_ += gridDim.x) {
    int b, g0;
    int64_t t0;
    {
        const unsigned L = xcd_logical_block((unsigned)nb_total, pb_);
        const unsigned gbi = L % (unsigned)n_gb, rest = L / (unsigned)n_gb;
        g0 = (int)gbi * GB;
        t0 = (int64_t)(rest % (unsigned)n_tt) * TT;
        b = (int)(rest / (unsigned)n_tt);
    }
#else
    GC_BLOCK_INDEX(n_tt, n_gb)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = wave_id();
    const float* xb = x + (int64_t)b * T * C;
    // the two groups' weight lines (contiguous in memory: g0, g0 + 1) as 16-byte pieces: requested first, written to LDS behind
    // the slab rows' requests (loads return in order: a wait placed here would hold up everything behind it)
    constexpr int NPL = GB * 4 * S18_WLP / 8, NLP = NPL / 256;
    static_assert(NPL % 256 == 0, "weight lines");
    f16x8 wline[NLP];
    {
        const f16x8* src = reinterpret_cast<const f16x8*>(wshift + (int64_t)g0 * (4 * S18_WLP));
#pragma unroll
        for (int j = 0; j < NLP; ++j) wline[j] = src[tid + 256 * j];
        __builtin_amdgcn_sched_barrier(0);
    }
    const int nrows = (int)(T - t0 < (int64_t)TT ? T - t0 : (int64_t)TT);
    GC_STAMP(0);

    // ---- slab fill from the split-form input (as gconv_mfma_kernel<.., XSPLIT = true>) ----
    {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const int r0 = tid / CH4, c4 = tid - r0 * CH4;
        const bool active = r0 < RPP;
        int so[2];
        bool lastpair[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ch = c4 * 4 + 2 * q, gl = ch / CG;
            so[q] = gl * GS + ch - gl * CG;
            lastpair[q] = ch - gl * CG == CG - 2;
        }
        __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)(T * C * 4), 0x00020000);
        const int64_t row0 = t0 - PADT + r0;
        const int cgl = g0 * CG + (active ? c4 * 4 : 0);
        const int voffs = (int)(row0 * C * 4) + (cgl >> 5) * 128 + (cgl & 31) * 2;
        constexpr int NPASS = (TIN + RPP - 1) / RPP;
        u32x2 vh[NPASS], vl[NPASS];
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            vh[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, voffs + u * RPP * C * 4, 0, 0);
            vl[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, voffs + 64 + u * RPP * C * 4, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NLP; ++j)
            reinterpret_cast<f16x8*>(s_wl)[tid + 256 * j] = wline[j];
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            const int ti = r0 + u * RPP;
            if (active && ti < TIN) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    *reinterpret_cast<unsigned*>(s_hi + so[q] + ti * P) = vh[u][q];
                    *reinterpret_cast<unsigned*>(s_lo + so[q] + ti * P) = vl[u][q];
                    if (lastpair[q]) {       // channels 16, 17 of a row: its pad columns 18, 19 too (finite bytes under zero weights)
                        *reinterpret_cast<unsigned*>(s_hi + so[q] + ti * P + 2) = 0u;
                        *reinterpret_cast<unsigned*>(s_lo + so[q] + ti * P + 2) = 0u;
                    }
                }
            }
        }
        // zeros behind the last row (the K round-up the last columns read past their window)
        const f16x2 z2 = {(_Float16)0.f, (_Float16)0.f};
        constexpr int tail2 = (GS - TIN * P) / 2;
        for (int i = tid; i < 2 * GB * tail2; i += 256)
            *reinterpret_cast<f16x2*>(slab + (i / tail2) * GS + TIN * P + 2 * (i % tail2)) = z2;
    }
    GC_STAMP(1);
    __syncthreads();
    GC_STAMP(2);

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int col = lane & 15, kg = lane >> 4;
    const int gl = w >> 1, h = w & 1, g = g0 + gl;
    const f32x2 s11 = {1.0f / 2048.0f, 1.0f / 2048.0f}, z0 = {0.f, 0.f}, al2 = {alpha, alpha};
    float am = 0.f;        // fp16-range guard: the largest |y| this lane turns into halves

    // ---- phase A: channels 16, 17 of 128 (64) steps as one shifted M tile ----
    constexpr int NRANGE = TT >= 128 ? TT / 128 : 1, NCOLV = TT >= 128 ? 16 : TT / S18_SH;
    if (h < NRANGE) {
        const int cn = col & (NCOLV - 1);                          // (TT = 64: columns 8-15 repeat 0-7 and are dropped)
        const int base = h * 128 + S18_SH * cn;                    // first output step of this column
        const _Float16* hs = s_hi + gl * GS + base * P + 8 * kg;   // 16-byte aligned: 8 rows = 320 bytes
        const _Float16* ls = s_lo + gl * GS + base * P + 8 * kg;
        // this lane's A row = (channel 16 + (col & 1), shift col >> 1): 8 halves at S18_WPRE + 32 c + 8 kg - 20 shift of its line
        typedef _Float16 f16x4a8 __attribute__((ext_vector_type(4), aligned(8)));
        const _Float16* wh_l = s_wl + ((gl * 2 + (col & 1)) * 2 + 0) * S18_WLP + S18_WPRE + 8 * kg - 20 * (col >> 1);
        const _Float16* wl_l = wh_l + S18_WLP;
        const float b16 = bias[g * CG + 16], b17 = bias[g * CG + 17];
        f32x4 acc = {b16, b17, b16, b17}, ax1 = {0.f, 0.f, 0.f, 0.f}, ax2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < S18_NKA; ++c) {
            const f16x4a8 a0 = *reinterpret_cast<const f16x4a8*>(wh_l + 32 * c), a1 = *reinterpret_cast<const f16x4a8*>(wh_l + 32 * c + 4);
            const f16x4a8 l0 = *reinterpret_cast<const f16x4a8*>(wl_l + 32 * c), l1 = *reinterpret_cast<const f16x4a8*>(wl_l + 32 * c + 4);
            const f16x8 ah = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, al = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            const f16x8 bh = *reinterpret_cast<const f16x8*>(hs + 32 * c), bl = *reinterpret_cast<const f16x8*>(ls + 32 * c);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            ax1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, ax1, 0, 0, 0);
            ax2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, ax2, 0, 0, 0);
        }
        // rows 4 kg + e of the tile = (shift 2 kg + e / 2, channel 16 + e % 2): this lane holds steps tl, tl + 1 x channels 16, 17
        const int tl = base + 2 * kg;
        f32x2 oa = {acc[0], acc[1]}, ob = {acc[2], acc[3]};
        oa = f32x2{ax1[0] + ax2[0], ax1[1] + ax2[1]} * s11 + oa;
        ob = f32x2{ax1[2] + ax2[2], ax1[3] + ax2[3]} * s11 + ob;
        const _Float16* rh = s_hi + gl * GS + (tl + PADT) * P + 16;
        const _Float16* rl = s_lo + gl * GS + (tl + PADT) * P + 16;
        const f16x2 ha = *reinterpret_cast<const f16x2*>(rh), la = *reinterpret_cast<const f16x2*>(rl);
        const f16x2 hb = *reinterpret_cast<const f16x2*>(rh + P), lb = *reinterpret_cast<const f16x2*>(rl + P);
        oa = al2 * __builtin_elementwise_max(oa, z0) + (__builtin_convertvector(la, f32x2) * s11 + __builtin_convertvector(ha, f32x2));
        ob = al2 * __builtin_elementwise_max(ob, z0) + (__builtin_convertvector(lb, f32x2) * s11 + __builtin_convertvector(hb, f32x2));
        if (col < NCOLV) {
            if (tl < nrows) am = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(oa[0]), __builtin_fabsf(oa[1])));
            if (tl + 1 < nrows) am = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(ob[0]), __builtin_fabsf(ob[1])));
            f16x2p hap, lap, hbp, lbp;
            if (range_flag) {
                split_f16x3_pair(oa[0], oa[1], hap, lap);
                split_f16x3_pair(ob[0], ob[1], hbp, lbp);
            } else {
                _Float16 hh[4], ll[4];
                split_f16x3(oa[0], hh[0], ll[0]); split_f16x3(oa[1], hh[1], ll[1]);
                split_f16x3(ob[0], hh[2], ll[2]); split_f16x3(ob[1], hh[3], ll[3]);
                hap = {hh[0], hh[1]}; lap = {ll[0], ll[1]}; hbp = {hh[2], hh[3]}; lbp = {ll[2], ll[3]};
            }
            _Float16* oh = s_hi + gl * GS + tl * P + CG;          // pad columns 18, 19 of the OUTPUT step's row
            _Float16* ol = s_lo + gl * GS + tl * P + CG;
            *reinterpret_cast<f16x2p*>(oh) = hap;
            *reinterpret_cast<f16x2p*>(ol) = lap;
            *reinterpret_cast<f16x2p*>(oh + P) = hbp;
            *reinterpret_cast<f16x2p*>(ol + P) = lbp;
        }
    }

    GC_STAMP(3);
    // ---- phase B: channels 0-15; wave h takes the blocks of its half of the tile, one after the other, WITHOUT barriers ----
    // A block's output halves overwrite the slab rows of its own 16 steps (dead once the block has its operands: the next block
    // reads from 16 rows further on).  The only rows two waves of a group contend for are the first 20 of the second half -- wave
    // 0's last blocks still read them --, so wave 1 keeps the output of its first two blocks in registers (8 VGPRs) until the
    // barrier that precedes the store phase anyway.
    typedef _Float16 f16x4a8 __attribute__((ext_vector_type(4), aligned(8)));
    f16x4a8 keep_h[2], keep_l[2];
    constexpr int NB = TT / 16, NBH = NB / 2;
    static_assert(NB % 2 == 0 && NBH >= 2, "blocks per tile");
    {
        // (requesting these 28 fragments earlier -- behind the slab rows, held in 112 registers across the fill and phase A --
        //  measured SLOWER: 0.282 against 0.255 ms, profiles/r4_gconv_time_shift_and_xcd_order.txt)
        f16x8 wh[NK], wl[NK];
        const f16x8* wf = reinterpret_cast<const f16x8*>(wfrag) + (int64_t)(g * 2) * (NK * 2 * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            wh[ks] = wf[(ks * 2 + 0) * 64];
            wl[ks] = wf[(ks * 2 + 1) * 64];
        }
        const int ch0 = 4 * kg;
        f32x4 bv;
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = bias[g * CG + ch0 + i];
        // fragment of K chunk c for output column (16 tb + col): (16 tb + col) * P + [permuted: 128 (c / 4) + 8 (c % 4), k group kg at
        // 64 (kg & 1) + 32 (kg >> 1) | plain (c >= NKP): 32 c + 8 kg]; two ds_read_b64 per fragment (gconv_frag)
        const int kgo = 64 * (kg & 1) + 32 * (kg >> 1);
        const _Float16* hs = s_hi + gl * GS + kgo;
        const _Float16* ls = s_lo + gl * GS + kgo;
        int four = 4;
        asm volatile("" : "+v"(four));
        const _Float16* hs2 = hs + four;
        const _Float16* ls2 = ls + four;
        const _Float16* hst = s_hi + gl * GS + 8 * kg;
        const _Float16* lst = s_lo + gl * GS + 8 * kg;
        const _Float16* hst2 = hst + four;
        const _Float16* lst2 = lst + four;
        auto foff = [&](int c, int b0) { return c < NKP ? b0 + 128 * (c >> 2) + 8 * (c & 3) : b0 + 32 * c; };
        auto frag_h = [&](int c, int b0) { return c < NKP ? gconv_frag<P>(hs + foff(c, b0), hs2 + foff(c, b0)) : gconv_frag<P>(hst + foff(c, b0), hst2 + foff(c, b0)); };
        auto frag_l = [&](int c, int b0) { return c < NKP ? gconv_frag<P>(ls + foff(c, b0), ls2 + foff(c, b0)) : gconv_frag<P>(lst + foff(c, b0), lst2 + foff(c, b0)); };
        constexpr int PF = GC_PF, RING = PF + 1;
        const int tb0 = h * NBH;
        int boff = (tb0 * 16 + col) * P;
        f16x8 chh[PF], cll[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            chh[q] = frag_h(q, boff);
            cll[q] = frag_l(q, boff);
        }
        const _Float16* rh = s_hi + gl * GS + ch0;
        const _Float16* rl = s_lo + gl * GS + ch0;
        GC_STAMP(4);
        auto block = [&](int tb, bool last, f16x4a8& oh, f16x4a8& ol) {
            f32x4 acc = bv, ax1 = {0.f, 0.f, 0.f, 0.f}, ax2 = {0.f, 0.f, 0.f, 0.f};
            const int nb = last ? boff : boff + 16 * P;
            f16x8 bh[RING], bl[RING];
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                bh[q] = chh[q];
                bl[q] = cll[q];
            }
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int pk = ks + PF;
                if (pk < NK) {
                    bh[pk % RING] = frag_h(pk, boff);
                    bl[pk % RING] = frag_l(pk, boff);
                } else {
                    bh[pk % RING] = frag_h(pk - NK, nb);
                    bl[pk % RING] = frag_l(pk - NK, nb);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], bh[ks % RING], acc, 0, 0, 0);
                ax1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], bl[ks % RING], ax1, 0, 0, 0);
                ax2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ks], bh[ks % RING], ax2, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                chh[q] = bh[(NK + q) % RING];
                cll[q] = bl[(NK + q) % RING];
            }
            boff = nb;
            f32x2 oa = {acc[0], acc[1]}, ob = {acc[2], acc[3]};
            oa = f32x2{ax1[0] + ax2[0], ax1[1] + ax2[1]} * s11 + oa;
            ob = f32x2{ax1[2] + ax2[2], ax1[3] + ax2[3]} * s11 + ob;
            const int ro = (tb * 16 + col + PADT) * P;
            const f16x2 h01 = *reinterpret_cast<const f16x2*>(rh + ro), l01 = *reinterpret_cast<const f16x2*>(rl + ro);
            const f16x2 h23 = *reinterpret_cast<const f16x2*>(rh + ro + 2), l23 = *reinterpret_cast<const f16x2*>(rl + ro + 2);
            oa = al2 * __builtin_elementwise_max(oa, z0) + (__builtin_convertvector(l01, f32x2) * s11 + __builtin_convertvector(h01, f32x2));
            ob = al2 * __builtin_elementwise_max(ob, z0) + (__builtin_convertvector(l23, f32x2) * s11 + __builtin_convertvector(h23, f32x2));
            const float am2 = __builtin_fmaxf(__builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(oa[0]), __builtin_fabsf(oa[1]))),
                                              __builtin_fmaxf(__builtin_fabsf(ob[0]), __builtin_fabsf(ob[1])));
            am = tb * 16 + col < nrows ? am2 : am;                 // (rows past the end of the sequence are never stored)
            f16x2p h01p, l01p, h23p, l23p;
            if (range_flag) {
                split_f16x3_pair(oa[0], oa[1], h01p, l01p);
                split_f16x3_pair(ob[0], ob[1], h23p, l23p);
            } else {
                _Float16 hh[4], ll[4];
                split_f16x3(oa[0], hh[0], ll[0]); split_f16x3(oa[1], hh[1], ll[1]);
                split_f16x3(ob[0], hh[2], ll[2]); split_f16x3(ob[1], hh[3], ll[3]);
                h01p = {hh[0], hh[1]}; l01p = {ll[0], ll[1]}; h23p = {hh[2], hh[3]}; l23p = {ll[2], ll[3]};
            }
            oh = f16x4a8{h01p[0], h01p[1], h23p[0], h23p[1]};
            ol = f16x4a8{l01p[0], l01p[1], l23p[0], l23p[1]};
        };
        auto put = [&](int tb, const f16x4a8& oh, const f16x4a8& ol) {
            const int so2 = gl * GS + (tb * 16 + col) * P + ch0;
            *reinterpret_cast<f16x4a8*>(s_hi + so2) = oh;
            *reinterpret_cast<f16x4a8*>(s_lo + so2) = ol;
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            block(tb0 + i, NBH == 2 && i == 1, keep_h[i], keep_l[i]);
            if (h == 0) put(tb0 + i, keep_h[i], keep_l[i]);
        }
        for (int i = 2; i < NBH; ++i) {
            f16x4a8 oh, ol;
            block(tb0 + i, i == NBH - 1, oh, ol);
            put(tb0 + i, oh, ol);
        }
    }
    note_range(am, range_flag);
    GC_STAMP(5);

    // ---- the tile leaves LDS along rows (as gconv_mfma_kernel's STAGED phase); channels 16, 17 come from the pad columns ----
    __syncthreads();
    if (h == 1) {         // every wave has read its operands: the rows wave 0 was still reading can take wave 1's first two blocks
        const int ch0 = 4 * kg;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int so2 = gl * GS + ((NBH + i) * 16 + col) * P + ch0;
            *reinterpret_cast<f16x4a8*>(s_hi + so2) = keep_h[i];
            *reinterpret_cast<f16x4a8*>(s_lo + so2) = keep_l[i];
        }
    }
    __syncthreads();
    GC_STAMP(6);
    {
        constexpr int CW = GB * CG, PW = 2, NPC = CW / (2 * PW), WPG = CG / 2, NPR = 2 * NPC;
        constexpr int RP = 256 / NPR, NPO = (TT + RP - 1) / RP;
        typedef unsigned u32xp __attribute__((ext_vector_type(PW)));
        const int r = tid / NPR, q = tid - r * NPR;
        const int a = q / NPC, p = q - a * NPC;          // a: 0 = hi halves, 1 = lo halves
        int wo[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int idx = PW * p + j, gq = idx / WPG, i = idx - gq * WPG;
            wo[j] = (a * GB + gq) * GS + (i < 8 ? 2 * i : CG) + r * P;
        }
        const int c = g0 * CG + 2 * PW * p;
        char* op = reinterpret_cast<char*>(ysplit) + ((int64_t)b * T + t0 + r) * C * 4 + (c >> 5) * 128 + (c & 31) * 2 + a * 64;
        const int64_t pstep = (int64_t)RP * C * 4;
        if (r < RP) {
            constexpr int CHK = 4;
            for (int p0 = 0; p0 < NPO; p0 += CHK) {
                u32xp v[CHK];
#pragma unroll
                for (int u = 0; u < CHK; ++u)
#pragma unroll
                    for (int j = 0; j < PW; ++j)
                        v[u][j] = *reinterpret_cast<const unsigned*>(slab + wo[j] + (p0 + u) * RP * P);
#pragma unroll
                for (int u = 0; u < CHK; ++u)
                    if ((p0 + u) * RP + r < nrows) *reinterpret_cast<u32xp*>(op + (p0 + u) * pstep) = v[u];
            }
        }
    }
    GC_STAMP(7);
#ifdef GC_S18_PERSIST
    __syncthreads();       // (the next block's fill overwrites the slab the store phase has just read)
    }
#endif
}

// weight lines of the shifted tile: [g][channel 16 / 17][hi, lo][S18_WLP] halves = S18_WPRE zeros | the channel's 420 weights in
// slab order (tap * 20 + input channel, pad channels 18, 19 zero) | zeros
__global__ void pack_gconv_shift18_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int groups) {
    const int64_t total = (int64_t)groups * 2 * S18_WLP;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % S18_WLP), c = (int)((idx / S18_WLP) & 1), g = (int)(idx / (2 * S18_WLP));
    const int k = i - S18_WPRE;
    float v = 0.f;
    if (k >= 0 && k < KS * 20) {
        const int j = k / 20, cc = k - 20 * j;
        if (cc < 18) v = src[((int64_t)(g * 18 + 16 + c) * 18 + cc) * KS + j];
    }
    _Float16 hi, lo;
    split_f16x3(v, hi, lo);
    _Float16* line = dst + ((int64_t)(g * 2 + c) * 2) * S18_WLP + i;
    line[0] = hi;
    line[S18_WLP] = lo;
}

// reference Conv1d weight [C_out, C_in/G, 21] -> MFMA A fragments [g][mt][K chunk][hi, lo][lane][8 halves]:
// row = mt * 16 + (lane & 15) (output channel); chunk c of segment s, k = 32 ks + 8 (lane >> 4) + i = j * pitch + cc
// -> tap tap0 + j * stride, input channel choff + cc
__global__ void pack_gconv_mfma_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, const GcPackDesc d) {
    const int nks = d.nk[0] + d.nk[1];
    const int64_t total = (int64_t)d.groups * d.mt_n * nks * 64 * 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx & 7), l = (int)((idx >> 3) & 63);
    int64_t r = idx >> 9;
    const int c = (int)(r % nks);
    r /= nks;
    const int mt = (int)(r % d.mt_n), g = (int)(r / d.mt_n);
    const int sg = c < d.nk[0] ? 0 : 1, ks = c - (sg ? d.nk[0] : 0);
    const int kg = l >> 4;
    const int co = mt * 16 + (l & 15);
    const int k = (sg == 0 && ks < d.perm_nk) ? 128 * (ks >> 2) + 8 * (ks & 3) + 64 * (kg & 1) + 32 * (kg >> 1) + i : 32 * ks + 8 * kg + i;
    const int j = k / d.pitch[sg], cc = k - j * d.pitch[sg];
    float v = 0.f;
    if (co < d.cog && j < d.ntap[sg] && cc < d.nch[sg])
        v = src[((int64_t)(g * d.cog + co) * d.cig + d.choff[sg] + cc) * KS + d.tap0[sg] + j * d.tapstep];
    _Float16 hi, lo;
    split_f16x3(v, hi, lo);
    const int64_t base = ((((int64_t)(g * d.mt_n + mt) * nks + c) * 2) * 64 + l) * 8 + i;
    dst[base] = hi;
    dst[base + 64 * 8] = lo;
}

// (the split-input slab load moves 4-channel pieces that must not straddle a 32-channel block: the workgroup's first channel,
//  g0 * CIG with g0 a multiple of GB, has to be a multiple of 4)
template <int CIG, int GB>
static constexpr bool g0_aligned() { return (CIG * GB) % 4 == 0; }

// grid of the matrix-core kernels: 1-D in the XCD-aware order (GC_BLOCK_INDEX) unless it would not fit 32 bits or the plain
// grid is asked for
static dim3 gconv_grid(int64_t tiles, int group_blocks, int B, int& n_tt, int& n_gb) {
    const int64_t nb = tiles * group_blocks * B;
    if (nb < (1ll << 31) && tiles < (1ll << 31) && !opt(OPT_GCONV_GRID_XYZ)) {
        n_tt = (int)tiles;
        n_gb = group_blocks;
        return dim3((unsigned)nb);
    }
    n_tt = n_gb = 0;
    return dim3((unsigned)tiles, (unsigned)group_blocks, (unsigned)B);
}

// x_split: the input is in the split form; y == NULL with ysplit: only the split form of the output is written
template <int CIG, int COG, int STRIDE, bool RESID, int GB, int TT>
static int launch_mfma_spec(const float* x, const void* wfrag, const float* bias, float alpha, float* y, void* ysplit, int B,
                            int64_t T_in, int64_t T_out, int C_in, int C_out, int groups, hipStream_t s, int* range_flag,
                            bool x_split = false) {
    using LY = GcLayout<CIG, STRIDE>;
    constexpr size_t lds = (size_t)2 * GB * (LY::slab(0, TT) + LY::slab(1, TT)) * sizeof(_Float16);
    typedef void (*kern_t)(const float*, const _Float16*, const float*, float, float*, _Float16*, int64_t, int64_t, int, int, int*, int, int, const GcC1);
    kern_t k;
    if (x_split) k = y ? (ysplit ? (kern_t)gconv_mfma_kernel<CIG, COG, STRIDE, RESID, GB, TT, 1, true> : (kern_t)gconv_mfma_kernel<CIG, COG, STRIDE, RESID, GB, TT, 0, true>)
                       : (kern_t)gconv_mfma_kernel<CIG, COG, STRIDE, RESID, GB, TT, 2, true>;
    else k = y ? (ysplit ? (kern_t)gconv_mfma_kernel<CIG, COG, STRIDE, RESID, GB, TT, 1, false> : (kern_t)gconv_mfma_kernel<CIG, COG, STRIDE, RESID, GB, TT, 0, false>)
               : (kern_t)gconv_mfma_kernel<CIG, COG, STRIDE, RESID, GB, TT, 2, false>;
    TAL_CHECK_ARG(y || ysplit, "gconv (fp16x3): no output buffer");
    TAL_CHECK_ARG(!ysplit || C_out % 32 == 0, "gconv (fp16x3): the split output needs C_out %% 32 == 0");
    TAL_CHECK_ARG(!x_split || (C_in % 32 == 0 && (g0_aligned<CIG, GB>())), "gconv (fp16x3): the split input needs C_in %% 32 == 0");
    static const void* attr_done[6] = {};          // (one slot per variant of this instantiation)
    const int slot = (x_split ? 3 : 0) + (y ? (ysplit ? 1 : 0) : 2);
    if (attr_done[slot] != reinterpret_cast<const void*>(k)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("gconv (fp16x3): cannot reserve %zu bytes of LDS", lds);
            return TAL_EHIP;
        }
        attr_done[slot] = reinterpret_cast<const void*>(k);
    }
    int n_tt, n_gb;
    const dim3 grid = gconv_grid(cdiv(T_out, TT), groups / GB, B, n_tt, n_gb);
    ProfScope prof(RESID ? PROF_GCONV_RES : PROF_GCONV_S2, 2.0 * (double)B * (double)T_out * C_out * CIG * KS, s);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, x, reinterpret_cast<const _Float16*>(wfrag), bias, alpha, y,
                       reinterpret_cast<_Float16*>(ysplit), T_in, T_out, C_in, C_out, range_flag, n_tt, n_gb, GcC1{});
    TAL_CHECK_LAUNCH("gconv (fp16x3)");
    return TAL_OK;
}

// Short inputs: tiles of 64 output steps instead of 256 (128 for stride 2) when the long tiles would not give every CU
// gconv_short_below workgroups (option "gconv_short_below", default 4: two rounds of two resident workgroups) -- a wave then
// walks 4 blocks of 16 steps instead of 16 behind its slab fill, on 4x the workgroups.  Same arithmetic per output: results
// do not depend on the tile length.
static bool gconv_short_tiles(int64_t T_out, int tt_long, int group_blocks, int B) {
    return cdiv(T_out, tt_long) * group_blocks * B < (int64_t)opt(OPT_GCONV_SHORT_BELOW) * device_cus();
}

// the slab loads address one batch item ([T, C] floats) through a buffer descriptor with 32-bit byte offsets
bool gconv_f16x3_fits(int64_t T, int C) { return T * C * 4 + (int64_t)300 * C * 4 < ((int64_t)1 << 31); }

// halves of the [g][mt][K chunk][hi, lo][lane][8] fragment table; the 18-channel stride-1 conv keeps a second table behind it:
// the weight lines of its two left-over channels (gconv18_shift_kernel, pack_gconv_shift18_kernel)
static size_t gconv_f16x3_main_halves(const GcPackDesc& d) { return (size_t)d.groups * d.mt_n * (d.nk[0] + d.nk[1]) * 2 * 64 * 8; }
static bool gconv_has_shift18(const GcPackDesc& d) { return d.cig == 18 && d.cog == 18 && d.tapstep == 1 && d.nseg == 1 && d.pitch[0] == 20; }

size_t gconv_f16x3_weight_bytes(int C_in, int C_out, int groups, int stride) {
    if (groups <= 0 || C_in % groups || C_out % groups) return 0;
    GcPackDesc d;
    if (!gconv_mfma_desc(C_in / groups, C_out / groups, stride, groups, d)) return 0;
    return gconv_f16x3_main_halves(d) * sizeof(_Float16) + (gconv_has_shift18(d) ? (size_t)groups * 4 * S18_WLP * sizeof(_Float16) : 0);
}

int launch_pack_gconv_f16x3(const float* w_ref, void* w_frag, int C_in, int C_out, int groups, int stride, hipStream_t s) {
    TAL_CHECK_ARG(w_ref && w_frag, "tal_pack_gconv_f16x3_weight: null pointer");
    GcPackDesc d;
    TAL_CHECK_ARG(groups > 0 && C_in % groups == 0 && C_out % groups == 0 && gconv_mfma_desc(C_in / groups, C_out / groups, stride, groups, d),
                  "tal_pack_gconv_f16x3_weight: no fp16x3 kernel for %d -> %d channels, groups=%d, stride %d", C_in, C_out, groups, stride);
    const int64_t total = (int64_t)groups * d.mt_n * (d.nk[0] + d.nk[1]) * 64 * 8;
    hipLaunchKernelGGL(pack_gconv_mfma_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w_ref, reinterpret_cast<_Float16*>(w_frag), d);
    TAL_CHECK_LAUNCH("tal_pack_gconv_f16x3_weight");
    if (gconv_has_shift18(d)) {
        const int64_t ts = (int64_t)groups * 2 * S18_WLP;
        hipLaunchKernelGGL(pack_gconv_shift18_kernel, dim3((unsigned)cdiv(ts, 256)), dim3(256), 0, s, w_ref,
                           reinterpret_cast<_Float16*>(w_frag) + gconv_f16x3_main_halves(d), groups);
        TAL_CHECK_LAUNCH("tal_pack_gconv_f16x3_weight (shifted tile)");
    }
    return TAL_OK;
}

// The first resize conv (1 mel bin -> 10 channels per group) and the first TDSBlock conv of the stage as ONE launch (FROMC1):
// mel [B][T_mel][groups] -> y_split [B][T1][10 groups] in the split form, T1 = (T_mel - 21) / 2 + 1.
bool gconv_c1_res_fusable(int C_in, int C_out, int groups, const float* mel) {
    return groups > 0 && C_in == groups && C_out == 10 * groups && groups % 4 == 0 && C_out % 32 == 0 &&
           (reinterpret_cast<uintptr_t>(mel) & 15) == 0 && gconv_f16x3_weight_bytes(C_out, C_out, groups, 1) > 0;
}
template <int TT>
static int launch_c1_res_tt(const GcC1& c1, const void* wfrag, const float* bias, float alpha, int B, int64_t T1, int C, int groups,
                            void* ysplit, hipStream_t s, int* range_flag) {
    using LY = GcLayout<10, 1>;
    constexpr int GB = 4, TIN = TT + KS - 1, NR = 2 * (TIN - 1) + KS;
    constexpr size_t lds = (size_t)2 * GB * LY::slab(0, TT) * sizeof(_Float16) + (size_t)NR * GB * sizeof(float);
    auto k = gconv_mfma_kernel<10, 10, 1, true, GB, TT, 2, true, true>;
    static const void* attr_done = nullptr;
    if (attr_done != reinterpret_cast<const void*>(k)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("gconv (resize conv + TDSBlock conv): cannot reserve %zu bytes of LDS", lds);
            return TAL_EHIP;
        }
        attr_done = reinterpret_cast<const void*>(k);
    }
    int n_tt, n_gb;
    const dim3 grid = gconv_grid(cdiv(T1, TT), groups / GB, B, n_tt, n_gb);
    ProfScope prof(PROF_GCONV_RES, 2.0 * (double)B * (double)T1 * C * (10 + 1) * KS, s);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, (const float*)nullptr, reinterpret_cast<const _Float16*>(wfrag), bias, alpha, (float*)nullptr,
                       reinterpret_cast<_Float16*>(ysplit), T1, T1, C, C, range_flag, n_tt, n_gb, c1);
    TAL_CHECK_LAUNCH("gconv (resize conv + TDSBlock conv)");
    return TAL_OK;
}
int launch_gconv_c1_res_f16x3(const float* mel, const float* w1, const float* b1, const float* in_mean, const void* w_frag, const float* bias,
                              float alpha, int B, int64_t T_mel, int groups, void* y_split, hipStream_t s, int* range_flag) {
    TAL_CHECK_ARG(mel && w1 && b1 && w_frag && bias && y_split, "gconv (resize conv + TDSBlock conv): null pointer");
    TAL_CHECK_ARG(B > 0 && T_mel >= KS && gconv_c1_res_fusable(groups, 10 * groups, groups, mel), "gconv (resize conv + TDSBlock conv): shape not supported");
    const int64_t T1 = (T_mel - KS) / 2 + 1;
    const int C = 10 * groups;
    TAL_CHECK_ARG(gconv_f16x3_fits(T1, C), "gconv (resize conv + TDSBlock conv): one batch item must stay below 2 GiB");
    GcC1 c1 = {mel, w1, b1, in_mean, T_mel, groups};
    if (gconv_short_tiles(T1, 256, groups / 4, B)) return launch_c1_res_tt<64>(c1, w_frag, bias, alpha, B, T1, C, groups, y_split, s, range_flag);
    return launch_c1_res_tt<256>(c1, w_frag, bias, alpha, B, T1, C, groups, y_split, s, range_flag);
}

int launch_gconv_res_f16x3(const float* x, const void* w_frag, const float* bias, float alpha, int B, int64_t T, int C, int groups,
                           float* y, void* y_split, hipStream_t s, int* range_flag, bool x_split) {
    TAL_CHECK_ARG(x && w_frag && bias && (y || y_split), "tal_gconv_res_f16x3_fwd: null pointer");
    TAL_CHECK_ARG((const void*)x != (const void*)y && (const void*)x != y_split, "tal_gconv_res_f16x3_fwd: in-place not supported (halo reads)");
    TAL_CHECK_ARG(gconv_f16x3_weight_bytes(C, C, groups, 1) > 0, "tal_gconv_res_f16x3_fwd: no fp16x3 kernel for C=%d groups=%d", C, groups);
    TAL_CHECK_ARG(B > 0 && T > 0, "tal_gconv_res_f16x3_fwd: bad shape");
    TAL_CHECK_ARG(!y_split || C % 32 == 0, "tal_gconv_res_f16x3_fwd: the split output needs C %% 32 == 0 (C=%d)", C);
    TAL_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && C % 4 == 0, "tal_gconv_res_f16x3_fwd: x must be 16-byte aligned");
    TAL_CHECK_ARG(gconv_f16x3_fits(T, C), "tal_gconv_res_f16x3_fwd: one batch item must stay below 2 GiB (T=%lld, C=%d)", (long long)T, C);
    const int cg = C / groups;
    if (cg == 18 && x_split && !y && y_split && GC_P18 == 20 && GC_KPERM && !opt(OPT_GCONV_NO_SHIFT18)) {
        // the product path of the last stage: split form in and out -> the time-shift-packed kernel
        GcPackDesc d;
        gconv_mfma_desc(18, 18, 1, groups, d);
        const _Float16* wshift = reinterpret_cast<const _Float16*>(w_frag) + gconv_f16x3_main_halves(d);
        const bool shortt = gconv_short_tiles(T, 256, groups / 2, B);
        using LY = GcLayout<18, 1>;
        const size_t lds = ((size_t)2 * 2 * LY::slab(0, shortt ? 64 : 256) + (size_t)2 * 4 * S18_WLP) * sizeof(_Float16);
        auto kern = shortt ? gconv18_shift_kernel<64> : gconv18_shift_kernel<256>;
        static const void* attr_done[2] = {};
        if (attr_done[shortt] != reinterpret_cast<const void*>(kern)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                set_error("gconv (fp16x3, shifted tile): cannot reserve %zu bytes of LDS", lds);
                return TAL_EHIP;
            }
            attr_done[shortt] = reinterpret_cast<const void*>(kern);
        }
        int n_tt, n_gb;
        dim3 grid = gconv_grid(cdiv(T, shortt ? 64 : 256), groups / 2, B, n_tt, n_gb);
        const int nb_total = n_tt > 0 ? (int)grid.x : 0;
#ifdef GC_S18_PERSIST
        if (n_tt > 0 && grid.x > 2u * (unsigned)device_cus()) grid = dim3(2u * (unsigned)device_cus());
#endif
        ProfScope prof(PROF_GCONV_RES, 2.0 * (double)B * (double)T * C * cg * KS, s);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, reinterpret_cast<const _Float16*>(w_frag), wshift, bias, alpha,
                           reinterpret_cast<_Float16*>(y_split), T, C, range_flag, n_tt, n_gb, nb_total);
        TAL_CHECK_LAUNCH("gconv (fp16x3, shifted tile)");
        return TAL_OK;
    }
    if (gconv_short_tiles(T, 256, groups / (cg == 18 ? 2 : 4), B)) {
        if (cg == 10) return launch_mfma_spec<10, 10, 1, true, 4, 64>(x, w_frag, bias, alpha, y, y_split, B, T, T, C, C, groups, s, range_flag, x_split);
        if (cg == 14) return launch_mfma_spec<14, 14, 1, true, 4, 64>(x, w_frag, bias, alpha, y, y_split, B, T, T, C, C, groups, s, range_flag, x_split);
        return launch_mfma_spec<18, 18, 1, true, 2, 64>(x, w_frag, bias, alpha, y, y_split, B, T, T, C, C, groups, s, range_flag, x_split);
    }
    // Long inputs, 14 channels per group: 128-step tiles (38 KB of LDS: four workgroups per CU instead of two at 71 KB) -- the halo grows
    // from 8 to 16 % of the rows filled, but twice the workgroups overlap their dependent phases: 0.317 -> 0.298 ms on the 1-hour shape,
    // 0.220 -> 0.200 on eight 5-minute segments (profiles/r6_gconv_tile_lengths.txt).  10 channels per group (53 KB: three per CU
    // already) lose 9 % with 128 steps and stay at 256; 64-step tiles lose everywhere on long inputs.  Same chain of MFMAs per output:
    // results do not depend on the tile length.  Option gconv_long_tt: 256 / 128 force one length for both widths.
    const int long_tt = opt(OPT_GCONV_LONG_TT);
    if (long_tt == 128 && cg == 10) return launch_mfma_spec<10, 10, 1, true, 4, 128>(x, w_frag, bias, alpha, y, y_split, B, T, T, C, C, groups, s, range_flag, x_split);
    if (long_tt != 256 && cg == 14) return launch_mfma_spec<14, 14, 1, true, 4, 128>(x, w_frag, bias, alpha, y, y_split, B, T, T, C, C, groups, s, range_flag, x_split);
    if (cg == 10) return launch_mfma_spec<10, 10, 1, true, 4, 256>(x, w_frag, bias, alpha, y, y_split, B, T, T, C, C, groups, s, range_flag, x_split);
    if (cg == 14) return launch_mfma_spec<14, 14, 1, true, 4, 256>(x, w_frag, bias, alpha, y, y_split, B, T, T, C, C, groups, s, range_flag, x_split);
    return launch_mfma_spec<18, 18, 1, true, 2, 256>(x, w_frag, bias, alpha, y, y_split, B, T, T, C, C, groups, s, range_flag, x_split);
}

int launch_gconv_s2_f16x3(const float* x, const void* w_frag, const float* bias, int B, int64_t T_in, int C_in, int C_out, int groups,
                          float* y, hipStream_t s, int* range_flag, bool x_split, void* y_split) {
    TAL_CHECK_ARG(x && w_frag && bias && (y || y_split), "tal_gconv_s2_f16x3_fwd: null pointer");
    TAL_CHECK_ARG(gconv_f16x3_weight_bytes(C_in, C_out, groups, 2) > 0, "tal_gconv_s2_f16x3_fwd: no fp16x3 kernel for %d -> %d channels, groups=%d",
                  C_in, C_out, groups);
    TAL_CHECK_ARG(B > 0 && T_in >= KS, "tal_gconv_s2_f16x3_fwd: T_in=%lld shorter than the kernel", (long long)T_in);
    TAL_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && C_in % 4 == 0, "tal_gconv_s2_f16x3_fwd: x must be 16-byte aligned");
    TAL_CHECK_ARG(gconv_f16x3_fits(T_in, C_in), "tal_gconv_s2_f16x3_fwd: one batch item must stay below 2 GiB (T=%lld, C=%d)", (long long)T_in, C_in);
    const int64_t T_out = (T_in - KS) / 2 + 1;
    if (gconv_short_tiles(T_out, 128, groups / (C_in / groups == 10 ? 4 : 2), B)) {
        if (C_in / groups == 10) return launch_mfma_spec<10, 14, 2, false, 4, 64>(x, w_frag, bias, 0.f, y, y_split, B, T_in, T_out, C_in, C_out, groups, s, range_flag, x_split);
        return launch_mfma_spec<14, 18, 2, false, 2, 64>(x, w_frag, bias, 0.f, y, y_split, B, T_in, T_out, C_in, C_out, groups, s, range_flag, x_split);
    }
    if (C_in / groups == 10) return launch_mfma_spec<10, 14, 2, false, 4, 128>(x, w_frag, bias, 0.f, y, y_split, B, T_in, T_out, C_in, C_out, groups, s, range_flag, x_split);
    return launch_mfma_spec<14, 18, 2, false, 2, 128>(x, w_frag, bias, 0.f, y, y_split, B, T_in, T_out, C_in, C_out, groups, s, range_flag, x_split);
}

// reference Conv1d weight [C_out, CIG, K] -> packed [G][CIG][K][COG]
__global__ void pack_gconv_kernel(const float* __restrict__ src, float* __restrict__ dst, int c_out, int cig, int ks,
                                  int groups) {
    const int total = c_out * cig * ks;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cog = c_out / groups;
    // i indexes dst: (((g*cig + ci)*ks + k)*cog + co)
    const int co = i % cog;
    int t = i / cog;
    const int k = t % ks;
    t /= ks;
    const int ci = t % cig;
    const int g = t / cig;
    dst[i] = src[((g * cog + co) * cig + ci) * ks + k];
}

}  // namespace tal

#ifdef GC_TIMELINE
extern "C" int tal_debug_gconv_timeline(void* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tal::g_gc_timeline), sizeof(tal::g_gc_timeline)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int tal_pack_gconv_weight(const float* w_ref, float* w_packed, int c_out, int c_in_per_group, int ksize,
                                     int groups, void* stream) {
    TAL_CHECK_ARG(w_ref && w_packed, "tal_pack_gconv_weight: null pointer");
    TAL_CHECK_ARG(ksize == tal::KS, "tal_pack_gconv_weight: kernel size %d (only 21 is built)", ksize);
    TAL_CHECK_ARG(groups > 0 && c_out % groups == 0 && c_in_per_group > 0, "tal_pack_gconv_weight: bad shape");
    const int total = c_out * c_in_per_group * ksize;
    hipLaunchKernelGGL(tal::pack_gconv_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_ref,
                       w_packed, c_out, c_in_per_group, ksize, groups);
    TAL_CHECK_LAUNCH("tal_pack_gconv_weight");
    return TAL_OK;
}
