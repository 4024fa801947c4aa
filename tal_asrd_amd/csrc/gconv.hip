// Grouped temporal convolutions of the TDS encoder (k = 21, groups = n_mels = 80).
//
//   stride-2 "resize" conv, padding 0 ........ tal/asr/models.py:363-364
//   TDSBlock grouped conv + ReLU + ReZero ..... tal/asr/models.py:304-308,329
//
// Activations are time-major [B, T, C].  A workgroup stages a [GB groups x C/G channels] x
// [time tile + halo] slab of x into LDS, channel-major with an odd row pitch (conflict-free
// both for the transposing store and for lanes that walk consecutive time steps).  Each
// wave owns one group at a time: a lane accumulates all C_out/G output channels of R time
// steps, so the weights of a group are wave-uniform and travel through SGPRs (s_load +
// v_fmac with a scalar operand) while the x taps come from LDS: no per-FMA LDS weight
// traffic.  The fp32 VALU rate equals the fp32 MFMA rate on gfx950 and a group's
// 10/14/18-wide block does not fill a 16- or 32-wide MFMA tile, so this stays on the VALU.
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

int launch_gconv_s2(const float* x, const float* wp, const float* bias, int B, int64_t T_in, int C_in, int C_out,
                    int groups, float* y, hipStream_t s) {
    TAL_CHECK_ARG(x && wp && bias && y, "tal_gconv_s2_fwd: null pointer");
    TAL_CHECK_ARG(groups > 0 && C_in % groups == 0 && C_out % groups == 0, "tal_gconv_s2_fwd: channels %d->%d not divisible by groups %d", C_in, C_out, groups);
    TAL_CHECK_ARG(B > 0 && T_in >= KS, "tal_gconv_s2_fwd: T_in=%lld shorter than the kernel", (long long)T_in);
    const int64_t T_out = (T_in - KS) / 2 + 1;
    const int cig = C_in / groups, cog = C_out / groups;
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
