// Log-mel front-end: LogMelSpec.forward (tal/asr/models.py:35-53) with the torchaudio
// 0.4.0 MelSpectrogram semantics restated in SURVEY.md section 8c:
//   reflect-pad 200 | frames of 400 @ hop 160 | periodic Hann | 400-point one-sided DFT |
//   re^2+im^2 | 201x80 HTK mel filterbank | log(. + eps) | minus one global scalar mean.
// Window product, DFT and power in float64 (near-silent frames have mel power within a few 1e-6 of eps, where an fp32 DFT is
// off by up to 1e-3 in the log domain -- measured; the reference's own fp32 FFT is off by 2e-4 from the float64 truth).
//
// Two forms (tal_logmel_fwd picks; include/tal_asrd.h):
//   logmel_fft_kernel  (default, round 5) a fast transform on the float64 VECTOR ALU, further down: 0.33 ms per hour of audio
//   logmel_kernel      (rounds 1-4, option logmel_mfma) the matrix-core form below: 0.81 ms per hour of audio
//
// The matrix form: one workgroup = 32 consecutive frames of one batch item.  The 5360 samples the frames
// span are read once, coalesced, into LDS (reflect indexing resolved at load time; one pad
// word per hop keeps the frame-strided MFMA operand reads conflict-free).  The windowed DFT
// is a [32 x 400] . [400 x (re|im) x 208] contraction on the FP64 matrix cores
// (v_mfma_f64_16x16x4_f64) against a Hann-folded float64 twiddle table that stays L2-resident
// (1.3 MB): no per-sample window multiply, no bit-reversal traffic.  re/im of a bin land
// in the same lane of two accumulators, so the power spectrum is formed in registers.  The (sparse, triangular) mel
// projection, the log and the partial sum for the global mean are done from an LDS copy of
// the power tile; the output is written once, coalesced, as [B, T, 80].
#include <math.h>

#include <cstddef>
#include <vector>

#include "common.h"
#include "dft20.h"

namespace tal {

constexpr int NFFT = 400;
constexpr int HOP = 160;
constexpr int NBIN = 201;
constexpr int NMEL = 80;
constexpr int FB = 32;                         // frames per workgroup (long inputs); short inputs: FB_SHORT
constexpr int FB_SHORT = 16;                   // (half the matrix work per wave, twice the workgroups: a 30-second clip is 94 workgroups of 32 frames)
constexpr int NTILE = 13;                      // 13 x 16 = 208 >= 201 bins
// samples per workgroup: (FB - 1) * HOP + NFFT + 1 (+1: the folded DFT touches x[400]), + one pad word per hop (in the kernel)
constexpr int NFOLD = 204;                     // folded DFT length: n = 0..200, padded to a multiple of 4
constexpr int PLD = NTILE * 16 + 1;            // power tile pitch (209)
constexpr int HTILE = 7;                       // 7 x 16 = 112 >= 101 bins (k = 0..100) of the even / odd half transforms
constexpr int NHALF = 104;                     // folded half-transform length: n' = 0..100, padded to a multiple of 4
constexpr int MAXW = 48;                       // max mel filter support in bins

struct LogmelPlan {
    double basis[NTILE * NFFT * 2 * 16];   // [tile][n][re|im][16 bins], Hann folded in
    double fbasis[NTILE * NFOLD * 16 * 2]; // symmetric-window form, [tile][n][16 bins][re|im] (one 16-byte load per lane and step):
                                           // re acts on x[n]+x[400-n], im on x[n]-x[400-n]
    double hbasis[HTILE * 2 * NHALF * 16 * 2]; // fbasis split by the parity of n, bins 0..100 only: [tile][n & 1][n >> 1][16 bins][re|im]
    int folded;                            // 1 if the window is symmetric (w[n] == w[400-n], w[0] == 0): use fbasis
    int mel_lo[NMEL];
    int mel_cnt[NMEL];
    alignas(16) float mel_w[NMEL * MAXW];      // read as 16-byte vectors by the short-input kernel
    // the 400 = 20 x 20 transform (logmel_fft_kernel): twiddles exp(-2 pi i n2 k1 / 400) as [n2][k1][re|im], and the window as given
    alignas(16) double tw[20 * 20 * 2];
    float win[NFFT];
    alignas(16) float mel_wc[768];             // the filters' weights back to back, each filter zero-padded to a multiple of 4 (mel_off[m] ..); mel_compact = 0 when they do not fit
    int mel_off[NMEL];
    int mel_compact;
};
static_assert(offsetof(LogmelPlan, mel_w) % 16 == 0, "LogmelPlan::mel_w must be 16-byte aligned");
static_assert(offsetof(LogmelPlan, tw) % 16 == 0 && offsetof(LogmelPlan, mel_wc) % 16 == 0, "LogmelPlan::tw / mel_wc must be 16-byte aligned");

// AT: element type of the waveform -- float, or _Float16 for callers that hand over `.half()` audio as the reference's
// GPU-era call sites do (tal/asr/system.py:92,285, tal/baseline/reconcile.py:78); the samples are widened while they are
// staged into LDS (exact), everything after that is the same arithmetic.
template <int FBT, typename AT>       // frames per workgroup: 32 (two 16-frame MFMA row blocks per wave and bin tile) or 16 (one)
__global__ __launch_bounds__(256, 3) void logmel_kernel(const LogmelPlan* __restrict__ plan,
                                                    const AT* __restrict__ audio, int64_t L, int64_t T, float eps,
                                                    float* __restrict__ out, double* __restrict__ partial) {
    constexpr int FB = FBT, NS = (FB - 1) * HOP + NFFT + 1, NSP = NS + NS / HOP + 2;
    constexpr bool TWO = FBT == 32;
    __shared__ float samp[NSP];
    __shared__ float P[FB * PLD];
    __shared__ double red[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = wave_id();
    const int b = blockIdx.y;
    const int64_t f0 = (int64_t)blockIdx.x * FB;
    const AT* ab = audio + (int64_t)b * L;

    // every sample of this thread is requested before the first one is stored (a load per loop iteration is a serial chain of
    // 11 / 21 round trips in front of the DFT when the waveform is not cache-resident; -2 us per call on a resident 1-minute clip)
    const int64_t p0 = f0 * HOP - NFFT / 2;
    constexpr int NIT = (NS + 255) / 256;
    AT sv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int64_t p = p0 + tid + it * 256;
        if (p < 0) p = -p;                       // reflect (no edge repeat), as torch.stft pad_mode='reflect'
        if (p >= L) p = 2 * (L - 1) - p;
        p = p < 0 ? 0 : (p >= L ? L - 1 : p);    // frames past T (tail block) only
        sv[it] = ab[p];
    }
    // short inputs (one workgroup per CU at most): the mel filter table comes into LDS under the DFT instead of two dependent
    // L2 round trips (support, then weights) per output inside the projection loop
    __shared__ __attribute__((aligned(16))) float melw[TWO ? 4 : NMEL * MAXW];
    __shared__ int mellc[TWO ? 2 : 2 * NMEL];
    if constexpr (!TWO) {
        constexpr int NV = NMEL * MAXW / 4;
        f32x4 mv[(NV + 255) / 256];
#pragma unroll
        for (int it = 0; it < (NV + 255) / 256; ++it) {
            const int i = tid + it * 256;
            mv[it] = reinterpret_cast<const f32x4*>(plan->mel_w)[i < NV ? i : NV - 1];
        }
        const int ml = tid < NMEL ? plan->mel_lo[tid] : (tid < 2 * NMEL ? plan->mel_cnt[tid - NMEL] : 0);
#pragma unroll
        for (int it = 0; it < (NV + 255) / 256; ++it) {
            const int i = tid + it * 256;
            if (i < NV) reinterpret_cast<f32x4*>(melw)[i] = mv[it];
        }
        if (tid < 2 * NMEL) mellc[tid] = ml;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256;
        if (i < NS) samp[i + i / HOP] = (float)sv[it];
    }
    __syncthreads();

    typedef double f64x4 __attribute__((ext_vector_type(4)));
    const int fi = lane & 15;  // frame (A row) / bin (B col) index inside a 16x16 tile
    const int kq = lane >> 4;  // which of the 4 k's of a 16x16x4 step
    for (int j = w; j < NTILE; j += 4) {
        f64x4 re0 = {0., 0., 0., 0.}, im0 = re0, re1 = re0, im1 = re0;
        if (plan->folded) {
            // Symmetric window: cos(2 pi k (400-n)/400) = cos(2 pi k n/400), sin flips sign, so
            //   Re X[k] =  sum_{n=0}^{200} c_n w[n] cos(.) (x[n] + x[400-n])
            //   Im X[k] = -sum_{n=1}^{199}     w[n] sin(.) (x[n] - x[400-n])
            // (c_200 = 1/2; the n = 0 row is w[0] = 0): half the fp64 MFMAs of the direct form.
            // One radix-2 step on top: with E[k] / O[k] the partial sums over even / odd n,
            //   X[k] = E[k] + O[k],   X[200-k] = conj(E[k] - O[k])     (cos / sin(2 pi (200-k) n / 400) = +-(-1)^n cos / sin(2 pi k n / 400))
            // so bins 0..100 of the two half sums give all 201 bins: half the MFMAs again (7 bin tiles instead of 13).
            if (j >= HTILE) continue;
            typedef double f64x2 __attribute__((ext_vector_type(2)));
            const f64x4 zero = {0., 0., 0., 0.};
            f64x4 ore0 = zero, oim0 = zero, ore1 = zero, oim1 = zero;
            const float* s0 = samp + fi * (HOP + 1);
            const float* s1 = s0 + 16 * (HOP + 1);
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const f64x2* fq = reinterpret_cast<const f64x2*>(plan->hbasis) + ((int64_t)(j * 2 + par) * NHALF + kq) * 16 + fi;
                f64x4 r0 = zero, i0 = zero, r1 = zero, i1 = zero;
                if constexpr (!TWO) {
                    // short inputs (one 16-frame workgroup per CU at most: nothing else hides a round trip): the 26 basis pairs of
                    // this (bin tile, parity) are requested up front -- 104 registers -- instead of two per loop iteration, each
                    // behind the previous one's MFMAs (35 -> ~15 us on a 30-second clip)
                    constexpr int NQ = NHALF / 4;
                    f64x2 bq[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) bq[q] = fq[q * 64];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        if (par && q == NQ - 1) break;
                        const int k = 4 * q;
                        const int n = 2 * (k + kq) + par, m = NFFT - n;
                        const int on = n + (n >= HOP ? 1 : 0);
                        const int om = m + (m >= 2 * HOP ? 2 : 1);
                        const double x0n = (double)s0[on], x0m = (double)s0[om];
                        r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0n + x0m, bq[q].x, r0, 0, 0, 0);
                        i0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0n - x0m, bq[q].y, i0, 0, 0, 0);
                    }
                } else
#pragma unroll 2      // (3 needs more than the 168 registers that keep three workgroups on a CU)
                for (int k = 0; k < (par ? NHALF - 4 : NHALF); k += 4) {
                    const int n = 2 * (k + kq) + par, m = NFFT - n;   // n <= 207, 193 <= m <= 400
                    const int on = n + (n >= HOP ? 1 : 0);            // + pad words crossed
                    const int om = m + (m >= 2 * HOP ? 2 : 1);
                    const double x0n = (double)s0[on], x0m = (double)s0[om];
                    const f64x2 bb = fq[k * 16];
                    const double br = bb.x, bi = bb.y;
                    r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0n + x0m, br, r0, 0, 0, 0);
                    i0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0n - x0m, bi, i0, 0, 0, 0);
                    if constexpr (TWO) {
                        const double x1n = (double)s1[on], x1m = (double)s1[om];
                        r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1n + x1m, br, r1, 0, 0, 0);
                        i1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1n - x1m, bi, i1, 0, 0, 0);
                    }
                }
                if (par == 0) { re0 = r0; im0 = i0; re1 = r1; im1 = i1; }
                else { ore0 = r0; oim0 = i0; ore1 = r1; oim1 = i1; }
            }
            // f64 16x16 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
            const int bin = j * 16 + fi;
            if (bin <= NFFT / 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int frame = kq + 4 * e;
                    const double a0 = re0[e] + ore0[e], b0 = im0[e] + oim0[e], c0 = re0[e] - ore0[e], d0 = im0[e] - oim0[e];
                    const double a1 = re1[e] + ore1[e], b1 = im1[e] + oim1[e], c1 = re1[e] - ore1[e], d1 = im1[e] - oim1[e];
                    P[frame * PLD + bin] = (float)(a0 * a0 + b0 * b0);
                    P[frame * PLD + NFFT / 2 - bin] = (float)(c0 * c0 + d0 * d0);
                    if constexpr (TWO) {
                        P[(frame + 16) * PLD + bin] = (float)(a1 * a1 + b1 * b1);
                        P[(frame + 16) * PLD + NFFT / 2 - bin] = (float)(c1 * c1 + d1 * d1);
                    }
                }
            }
            continue;
        } else {
            const double* bp = plan->basis + ((int64_t)j * NFFT + kq) * 32 + fi;
            const float* sp0 = samp + fi * (HOP + 1) + kq;
            const float* sp1 = sp0 + 16 * (HOP + 1);
#pragma unroll 1
            for (int k0 = 0; k0 < NFFT; k0 += 40) {       // 40 | HOP: a chunk never straddles a pad word
                const int base = k0 + k0 / HOP;           // (k + kq) / HOP == k0 / HOP inside the chunk
                const double* bq = bp + k0 * 32;
#pragma unroll
                for (int kk = 0; kk < 40; kk += 4) {
                    const double a0 = (double)sp0[base + kk];
                    const double br = bq[kk * 32];
                    const double bi = bq[kk * 32 + 16];
                    re0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, br, re0, 0, 0, 0);
                    im0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bi, im0, 0, 0, 0);
                    if constexpr (TWO) {
                        const double a1 = (double)sp1[base + kk];
                        re1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, br, re1, 0, 0, 0);
                        im1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bi, im1, 0, 0, 0);
                    }
                }
            }
        }
        // f64 16x16 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int frame = kq + 4 * e;
            P[frame * PLD + j * 16 + fi] = (float)(re0[e] * re0[e] + im0[e] * im0[e]);
            if constexpr (TWO) P[(frame + 16) * PLD + j * 16 + fi] = (float)(re1[e] * re1[e] + im1[e] * im1[e]);
        }
    }
    __syncthreads();

    double local = 0.0;
    for (int idx = tid; idx < FB * NMEL; idx += 256) {
        const int frame = idx / NMEL;
        const int m = idx - frame * NMEL;
        const int lo = TWO ? plan->mel_lo[m] : mellc[m];
        const int cnt = TWO ? plan->mel_cnt[m] : mellc[NMEL + m];
        const float* wm = TWO ? plan->mel_w + m * MAXW : melw + m * MAXW;
        const float* pr = P + frame * PLD + lo;
        float sum = 0.f;
        for (int i = 0; i < cnt; ++i) sum = fmaf(pr[i], wm[i], sum);
        const float v = logf(sum + eps);
        if (f0 + frame < T) {
            out[((int64_t)b * T + f0 + frame) * NMEL + m] = v;
            local += (double)v;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if (lane == 0) red[w] = local;
    __syncthreads();
    if (tid == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same front-end as a FAST transform on the float64 vector ALU (round 5).  The matrix form above spends 46.6 k float64
// multiply-adds per frame (728 MFMAs per 16 frames) and is bound by the fp64 matrix rate -- which on this chip is the fp64 VECTOR
// rate (78.6 TFLOP/s both), so an O(N log N) transform on the vector ALU wins by its operation count alone: ~7 k operations per frame.
//   * two real frames a, b ride one complex transform: z = w (a + i b), A[k] = (Z[k] + conj Z[400-k]) / 2, B[k] = (Z[k] - conj Z[400-k]) / 2i
//   * 400 = 20 x 20 (n = 20 n1 + n2, k = k1 + 20 k2): a lane holds one 20-point transform in registers (dft20.h: prime-factor
//     4 x 5, no inner twiddles); pass 1 = 20 lanes per frame pair (one per n2) transform over n1 and multiply by W400^(n2 k1),
//     the [k1][n2] hand-over goes through LDS (pitch 21: the strided reads of pass 2 hit 16 different 16-byte bank groups),
//     pass 2 = 20 lanes per pair (one per k1) transform over n2 and write Z[k1 + 20 k2] back over the same buffer
//   * window applied in the time domain (any window: no symmetry needed), every product and sum in float64 as before
//   * mel projection: a wave takes 64 / frames mel filters at a time, lane = (filter, frame), so the lanes of a filter run the same
//     number of steps (lane = (frame, filter) made every wave wait for its widest filter: 48 steps for 5 on average); the filter
//     weights come from a compact copy in LDS; results are staged in LDS and written out coalesced.
// One workgroup = NP frame pairs on 128 threads (120 of them transforming); 53 KB of LDS: three workgroups per CU.
constexpr int YP = 21;
// (ablation build -DLM_TIMELINE: workgroup 100 stamps the 100 MHz wall clock after every stage of its first 8 blocks;
//  scripts/r5_logmel_timeline.py reads them through tal_debug_logmel_timeline.  Not part of the product library.)
#ifdef LM_TIMELINE
__device__ unsigned long long g_lm_tl[8 * 8];
#define LM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 100 && lm_it < 8) g_lm_tl[lm_it * 8 + (i)] = wall_clock64(); } while (0)
#else
#define LM_STAMP(i) do { } while (0)
#endif
constexpr int FFT_FB = 12;                        // frames per workgroup
constexpr int MELC = 768;                         // capacity of the compact filter table (the filters' supports, each padded to a multiple of 4 with zeros; 402 + padding for the HTK bank)

template <int NP, typename AT>
__global__ __launch_bounds__(128, 2) void logmel_fft_kernel(const LogmelPlan* __restrict__ plan, const AT* __restrict__ audio, int64_t L,
                                                         int64_t T, int64_t nblk, int64_t total, float eps, float* __restrict__ out,
                                                         double* __restrict__ partial) {
    constexpr int NTH = 128;
    constexpr int FBK = 2 * NP, NS = (FBK - 1) * HOP + NFFT;
    constexpr int SP = NS > FBK * PLD ? NS : FBK * PLD;
    static_assert(NP * 20 <= NTH && FBK * NMEL * 4 <= NP * 20 * YP * 16, "lane roles / staging buffer");
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    __shared__ float sp[SP];                          // the samples, then (from the power stage on) the power tile
    __shared__ f64x2 Yb[NP * 20 * YP];                // [pair][k1][n2] hand-over, then Z[pair][k], then the staged outputs
    __shared__ __attribute__((aligned(16))) float melw[MELC];
    __shared__ int mellc[NMEL];                       // lo | cnt << 8 | off << 16
    __shared__ double red[NTH / 64];
    float* samp = sp;
    float* P = sp;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int pr = tid / 20, q = tid - pr * 20;       // frame pair; n2 in pass 1, k1 in pass 2
    const bool act = pr < NP;
    constexpr int NIT = (NS + NTH - 1) / NTH;
    AT sv[NIT];
    // the samples of block `blk` (flattened over the batch), reflect indexing resolved here
    auto request = [&](int64_t blk) {
        const int64_t b = blk / nblk;
        const int64_t p0 = (blk - b * nblk) * FBK * HOP - NFFT / 2;
        const AT* ab = audio + b * L;
        if (p0 >= 0 && p0 + NIT * NTH <= L) {        // interior block (all but the first and the last few of an item): no index arithmetic
            const AT* src = ab + p0 + tid;
#pragma unroll
            for (int it = 0; it < NIT; ++it) sv[it] = src[it * NTH];
            return;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int64_t p = p0 + tid + it * NTH;
            if (p < 0) p = -p;                       // reflect (no edge repeat), as torch.stft pad_mode='reflect'
            if (p >= L) p = 2 * (L - 1) - p;
            p = p < 0 ? 0 : (p >= L ? L - 1 : p);    // frames past T (tail block) only
            sv[it] = ab[p];
        }
    };
    int64_t blk = blockIdx.x;
    request(blk);
    // loop-invariant operands: the lane's twiddles and window taps (registers), the filter table (LDS)
    // (the 19 twiddles W400^(q k1) of a lane are rebuilt per block from W^q and W^4q -- four chains of at most four products --
    //  instead of living in 76 registers across the loop: one wave more per SIMD)
    f64x2 tw1, tw4;
    const float* wp = plan->win + (act ? q : 0);      // the lane's 20 window taps are re-read per block (cache-resident; 20 registers less across the loop)
    {
        const f64x2* twp = reinterpret_cast<const f64x2*>(plan->tw) + (act ? q : 0) * 20;
        tw1 = twp[1];
        tw4 = twp[4];
    }
    const bool compact = plan->mel_compact != 0;
    {
        for (int i = tid; i < MELC / 4; i += NTH) reinterpret_cast<f32x4*>(melw)[i] = reinterpret_cast<const f32x4*>(plan->mel_wc)[i];
        if (tid < NMEL) mellc[tid] = plan->mel_lo[tid] | (plan->mel_cnt[tid] << 8) | (plan->mel_off[tid] << 16);
    }
    // mel stage roles: a wave takes MPW filters at a time, lane = (filter, frame PAIR): the two frames of a lane share the weight
    // reads and ride packed fp32 multiply-adds; the lane's NPASS filters never change
    constexpr int MPW = 64 / NP, NPASS = (NMEL + (NTH / 64) * MPW - 1) / ((NTH / 64) * MPW);
    const int mf = lane / NP, fr = 2 * (lane - mf * NP);
    __syncthreads();                                  // (the filter table is in LDS)
    double local = 0.0;
    int lm_it = 0;
    while (blk < total) {
        const int64_t b = blk / nblk;
        const int64_t f0 = (blk - b * nblk) * FBK;
        LM_STAMP(0);
        // (an opaque zero per block: the window taps, the lane's filter words and their addresses are loop-invariant, and hoisted
        //  out of the block loop -- the taps already widened to float64 -- they are ~70 registers that get spilled around the passes)
        int oz = 0;
        asm volatile("" : "+v"(oz));
        float wn[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) wn[i] = wp[20 * i + oz];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * NTH;
            if (i < NS) samp[i] = (float)sv[it];
        }
        __syncthreads();
        LM_STAMP(1);
        const int64_t nxt = blk + gridDim.x;

        double xr[20], xi[20], yr[20], yi[20];
        if (act) {
            const float* sa = samp + (2 * pr) * HOP + q;
            const float* sb = sa + HOP;
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) {
                const double wv = (double)wn[n1];
                xr[n1] = wv * (double)sa[20 * n1];
                xi[n1] = wv * (double)sb[20 * n1];
            }
            dft20(xr, xi, yr, yi);
            f64x2* yo = Yb + pr * 20 * YP + q;
            f64x2 t[4];                                   // W^(q (4 a + c)), c = 0..3, advanced over a by W^4q
            // (opaque to the optimiser: the products are loop-invariant, and hoisted out of the block loop they are 76 live registers)
            asm volatile("" : "+v"(tw1.x), "+v"(tw1.y), "+v"(tw4.x), "+v"(tw4.y));
            t[0] = f64x2{1.0, 0.0};
            t[1] = tw1;
            t[2] = f64x2{tw1.x * tw1.x - tw1.y * tw1.y, 2.0 * tw1.x * tw1.y};
            t[3] = f64x2{t[2].x * tw1.x - t[2].y * tw1.y, t[2].x * tw1.y + t[2].y * tw1.x};
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k1 = 4 * a + c;
                    yo[k1 * YP] = f64x2{yr[k1] * t[c].x - yi[k1] * t[c].y, yr[k1] * t[c].y + yi[k1] * t[c].x};
                    if (a < 4) t[c] = f64x2{t[c].x * tw4.x - t[c].y * tw4.y, t[c].x * tw4.y + t[c].y * tw4.x};
                }
        }
        __syncthreads();
        LM_STAMP(2);
        if (nxt < total) request(nxt);               // the next block's samples: in flight under the rest of this block (not before pass 1: its registers)
        if (act) {
            const f64x2* yin = Yb + (pr * 20 + q) * YP;
#pragma unroll
            for (int n2 = 0; n2 < 20; ++n2) {
                const f64x2 v = yin[n2];
                xr[n2] = v.x;
                xi[n2] = v.y;
            }
            dft20(xr, xi, yr, yi);
        }
        __syncthreads();                                  // every row is in registers: the buffer becomes Z[k], natural order
        if (act) {
            f64x2* zo = Yb + pr * 20 * YP + q;
#pragma unroll
            for (int k2 = 0; k2 < 20; ++k2) zo[20 * k2] = f64x2{yr[k2], yi[k2]};
        }
        __syncthreads();                                  // (the samples are dead since pass 1: their buffer takes the power tile)
        LM_STAMP(3);
        if (tid < FBK * 8) P[(tid >> 3) * PLD + NBIN + (tid & 7)] = 0.f;       // (columns 201..208: read, times a zero weight, by the 4-wide mel steps)
        {
            // bins tid and tid + 128 of every pair: all the reads of a lane are requested before its first product
            const int ka = tid, kb = tid + NTH;
            const bool hasb = kb < NBIN;
#pragma unroll
            for (int h = 0; h < NP; h += NP / 2) {
                f64x2 za[NP / 2], zam[NP / 2], zb[NP / 2], zbm[NP / 2];
#pragma unroll
                for (int p = 0; p < NP / 2; ++p) {
                    const f64x2* Z = Yb + (h + p) * 20 * YP;
                    za[p] = Z[ka];
                    zam[p] = Z[ka ? NFFT - ka : 0];
                    zb[p] = Z[hasb ? kb : 0];
                    zbm[p] = Z[hasb ? NFFT - kb : 0];
                }
#pragma unroll
                for (int p = 0; p < NP / 2; ++p) {
                    float* Pa = P + (2 * (h + p)) * PLD;
                    {
                        const double ar = za[p].x + zam[p].x, ai = za[p].y - zam[p].y, br = za[p].y + zam[p].y, bi = za[p].x - zam[p].x;
                        Pa[ka] = (float)(0.25 * (ar * ar + ai * ai));
                        Pa[PLD + ka] = (float)(0.25 * (br * br + bi * bi));
                    }
                    if (hasb) {
                        const double ar = zb[p].x + zbm[p].x, ai = zb[p].y - zbm[p].y, br = zb[p].y + zbm[p].y, bi = zb[p].x - zbm[p].x;
                        Pa[kb] = (float)(0.25 * (ar * ar + ai * ai));
                        Pa[PLD + kb] = (float)(0.25 * (br * br + bi * bi));
                    }
                }
            }
        }
        __syncthreads();
        LM_STAMP(4);

        float* stage = reinterpret_cast<float*>(Yb);      // [frame][80] (Z is dead)
        if (mf < MPW) {
            // the first four taps of all of this lane's filters are requested together (NPASS serial read -> multiply -> log chains
            // otherwise, each two LDS latencies long), then taps 4..7; wider filters finish in a short loop
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const float* pa0 = P + fr * PLD + oz;
            f32x2 sum[NPASS];
            int mlc[NPASS];
#pragma unroll
            for (int pi = 0; pi < NPASS; ++pi) {
                const int m = (pi * (NTH / 64) + w) * MPW + mf;
                mlc[pi] = mellc[(m < NMEL ? m : 0) + oz];
            }
            if (compact) {
#pragma unroll
                for (int pi = 0; pi < NPASS; ++pi) {
                    const int lc = mlc[pi];
                    const float* pa = pa0 + (lc & 255);
                    const float* pb = pa + PLD;
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(melw + (lc >> 16));
                    const f32x2 q0 = {pa[0], pb[0]}, q1 = {pa[1], pb[1]}, q2 = {pa[2], pb[2]}, q3 = {pa[3], pb[3]};
                    sum[pi] = q0 * wv.x;
                    sum[pi] = __builtin_elementwise_fma(q1, f32x2{wv.y, wv.y}, sum[pi]);
                    sum[pi] = __builtin_elementwise_fma(q2, f32x2{wv.z, wv.z}, sum[pi]);
                    sum[pi] = __builtin_elementwise_fma(q3, f32x2{wv.w, wv.w}, sum[pi]);
                }
                // (filters of up to 4 taps: zero weights and zero operands; the reads stay inside the lane's own tile rows and the table)
#pragma unroll
                for (int pi = 0; pi < NPASS; ++pi) {
                    const int lc = mlc[pi];
                    const bool more = ((lc >> 8) & 255) > 4;
                    const float* pa = pa0 + (lc & 255);
                    const float* pb = pa + PLD;
                    f32x4 wv = *reinterpret_cast<const f32x4*>(melw + (lc >> 16) + 4);
                    f32x2 q0 = {pa[4], pb[4]}, q1 = {pa[5], pb[5]}, q2 = {pa[6], pb[6]}, q3 = {pa[7], pb[7]};
                    const f32x2 z2 = {0.f, 0.f};
                    if (!more) { wv = f32x4{0.f, 0.f, 0.f, 0.f}; q0 = z2; q1 = z2; q2 = z2; q3 = z2; }
                    sum[pi] = __builtin_elementwise_fma(q0, f32x2{wv.x, wv.x}, sum[pi]);
                    sum[pi] = __builtin_elementwise_fma(q1, f32x2{wv.y, wv.y}, sum[pi]);
                    sum[pi] = __builtin_elementwise_fma(q2, f32x2{wv.z, wv.z}, sum[pi]);
                    sum[pi] = __builtin_elementwise_fma(q3, f32x2{wv.w, wv.w}, sum[pi]);
                }
#pragma unroll
                for (int pi = 0; pi < NPASS; ++pi) {
                    const int lc = mlc[pi];
                    const int cnt = (lc >> 8) & 255;
                    const float* pa = pa0 + (lc & 255);
                    const float* pb = pa + PLD;
                    const float* wm = melw + (lc >> 16);
                    for (int i = 8; i < cnt; i += 4) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wm + i);
                        const f32x2 q0 = {pa[i], pb[i]}, q1 = {pa[i + 1], pb[i + 1]}, q2 = {pa[i + 2], pb[i + 2]}, q3 = {pa[i + 3], pb[i + 3]};
                        sum[pi] = __builtin_elementwise_fma(q0, f32x2{wv.x, wv.x}, sum[pi]);
                        sum[pi] = __builtin_elementwise_fma(q1, f32x2{wv.y, wv.y}, sum[pi]);
                        sum[pi] = __builtin_elementwise_fma(q2, f32x2{wv.z, wv.z}, sum[pi]);
                        sum[pi] = __builtin_elementwise_fma(q3, f32x2{wv.w, wv.w}, sum[pi]);
                    }
                }
#pragma unroll
                for (int pi = 0; pi < NPASS; ++pi) {
                    const int m = (pi * (NTH / 64) + w) * MPW + mf;
                    if (m < NMEL) {
                        stage[fr * NMEL + m] = logf(sum[pi].x + eps);
                        stage[(fr + 1) * NMEL + m] = logf(sum[pi].y + eps);
                    }
                }
            } else {                                  // (a filterbank whose supports do not fit the compact table: weights from memory)
#pragma unroll 1
                for (int pi = 0; pi < NPASS; ++pi) {
                    const int m = (pi * (NTH / 64) + w) * MPW + mf;
                    const int lc = mellc[m < NMEL ? m : 0];
                    const int cnt = (lc >> 8) & 255;
                    const float* pa = pa0 + (lc & 255);
                    const float* pb = pa + PLD;
                    const float* wm = plan->mel_w + (m < NMEL ? m : 0) * MAXW;
                    float acc0 = 0.f, acc1 = 0.f;
                    for (int i = 0; i < cnt; ++i) {
                        acc0 = fmaf(pa[i], wm[i], acc0);
                        acc1 = fmaf(pb[i], wm[i], acc1);
                    }
                    if (m < NMEL) {
                        stage[fr * NMEL + m] = logf(acc0 + eps);
                        stage[(fr + 1) * NMEL + m] = logf(acc1 + eps);
                    }
                }
            }
        }
        __syncthreads();
        LM_STAMP(5);
        {
            const int64_t nf = T - f0 < FBK ? T - f0 : FBK;           // frames of this block that exist
            float* ob = out + (b * T + f0) * NMEL;
            for (int idx = tid; idx < (int)nf * NMEL; idx += NTH) {
                const float v = stage[idx];
                ob[idx] = v;
                local += (double)v;
            }
        }
        __syncthreads();                                  // (the next block's samples and hand-over overwrite P and the staged outputs)
        LM_STAMP(6);
        ++lm_it;
        blk = nxt;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if (lane == 0) red[w] = local;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = red[0] + red[1];
}

// fixed-order reduction of the per-workgroup partial sums -> {sum, count} and the float mean
__global__ __launch_bounds__(256) void logmel_mean_kernel(const double* __restrict__ partial, int64_t n, double count,
                                                         float* __restrict__ mean_out, double* __restrict__ sum_out,
                                                         float* __restrict__ mean_ws) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = (float)(red[0] / count);
        mean_ws[0] = mean;
        if (mean_out) mean_out[0] = mean;
        if (sum_out) {
            sum_out[0] = red[0];
            sum_out[1] = count;
        }
    }
}

__global__ __launch_bounds__(256) void subtract_scalar_kernel(float* __restrict__ x, int64_t n,
                                                             const float* __restrict__ mean) {
    const float m = mean[0];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    float4* x4 = reinterpret_cast<float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = x4[i];
        v.x -= m; v.y -= m; v.z -= m; v.w -= m;
        x4[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] -= m;
}

static int launch_subtract(float* x, int64_t n, const float* mean, hipStream_t s) {
    if (n == 0) return TAL_OK;
    int64_t blocks = cdiv(n / 4 + 1, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(subtract_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, mean);
    TAL_CHECK_LAUNCH("tal_subtract_scalar");
    return TAL_OK;
}

}  // namespace tal

using namespace tal;

#ifdef LM_TIMELINE
extern "C" int tal_debug_logmel_timeline(void* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tal::g_lm_tl), sizeof(tal::g_lm_tl)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int64_t tal_logmel_num_frames(int64_t L) { return 1 + L / HOP; }

extern "C" size_t tal_logmel_plan_bytes(void) { return sizeof(LogmelPlan); }

extern "C" int tal_logmel_plan_init(const float* window, const float* fb, void* plan, void* stream) {
    TAL_CHECK_ARG(window && fb && plan, "tal_logmel_plan_init: null pointer");
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> hwin(NFFT), hfb(NBIN * NMEL);
    if (hipMemcpyAsync(hwin.data(), window, NFFT * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(hfb.data(), fb, NBIN * NMEL * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        set_error("tal_logmel_plan_init: cannot read window/fb from the device");
        return TAL_EHIP;
    }
    std::vector<char> buf(sizeof(LogmelPlan));
    LogmelPlan* hp = reinterpret_cast<LogmelPlan*>(buf.data());
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j < NTILE; ++j)
        for (int n = 0; n < NFFT; ++n)
            for (int c = 0; c < 16; ++c) {
                const int bin = j * 16 + c;
                double re = 0.0, im = 0.0;
                if (bin < NBIN) {
                    const int ph = (int)(((int64_t)bin * n) % NFFT);  // exact phase reduction
                    const double ang = two_pi * (double)ph / (double)NFFT;
                    re = (double)hwin[n] * cos(ang);
                    im = -(double)hwin[n] * sin(ang);
                }
                hp->basis[((j * NFFT + n) * 2 + 0) * 16 + c] = re;
                hp->basis[((j * NFFT + n) * 2 + 1) * 16 + c] = im;
            }
    // Folded form for a symmetric window.  torch.hann_window evaluates w[n] and w[400-n] separately
    // in float32, so they may differ in the last bit; their mean is used and the asymmetric part
    // (|w[n] - w[400-n]| / 2 <= 4e-7 max|w| required, i.e. below fp32 epsilon of the product) is
    // dropped.  Any other window (or w[0] != 0) keeps the direct 400-term form.
    double wmax = 0.0, asym = 0.0;
    for (int n = 0; n < NFFT; ++n) wmax = fabs((double)hwin[n]) > wmax ? fabs((double)hwin[n]) : wmax;
    for (int n = 1; n < NFFT / 2; ++n) {
        const double d = fabs((double)hwin[n] - (double)hwin[NFFT - n]) * 0.5;
        asym = d > asym ? d : asym;
    }
    hp->folded = (hwin[0] == 0.f && asym <= 4e-7 * wmax && !opt(OPT_LOGMEL_NO_FOLD)) ? 1 : 0;
    for (int j = 0; j < NTILE; ++j)
        for (int n = 0; n < NFOLD; ++n)
            for (int c = 0; c < 16; ++c) {
                const int bin = j * 16 + c;
                double re = 0.0, im = 0.0;
                if (bin < NBIN && n >= 1 && n <= NFFT / 2) {
                    const int ph = (int)(((int64_t)bin * n) % NFFT);
                    const double ang = two_pi * (double)ph / (double)NFFT;
                    const double ws = n == NFFT / 2 ? 0.5 * (double)hwin[n]
                                                    : 0.5 * ((double)hwin[n] + (double)hwin[NFFT - n]);
                    re = ws * cos(ang);
                    im = n == NFFT / 2 ? 0.0 : -ws * sin(ang);
                }
                hp->fbasis[((j * NFOLD + n) * 16 + c) * 2 + 0] = re;
                hp->fbasis[((j * NFOLD + n) * 16 + c) * 2 + 1] = im;
            }
    for (int j = 0; j < HTILE; ++j)
        for (int par = 0; par < 2; ++par)
            for (int h = 0; h < NHALF; ++h)
                for (int c = 0; c < 16; ++c) {
                    const int bin = j * 16 + c, n = 2 * h + par;
                    double re = 0.0, im = 0.0;
                    if (bin <= NFFT / 4 && n >= 1 && n <= NFFT / 2) {
                        const int ph = (int)(((int64_t)bin * n) % NFFT);
                        const double ang = two_pi * (double)ph / (double)NFFT;
                        const double ws = n == NFFT / 2 ? 0.5 * (double)hwin[n] : 0.5 * ((double)hwin[n] + (double)hwin[NFFT - n]);
                        re = ws * cos(ang);
                        im = n == NFFT / 2 ? 0.0 : -ws * sin(ang);
                    }
                    hp->hbasis[((((j * 2 + par) * NHALF + h) * 16) + c) * 2 + 0] = re;
                    hp->hbasis[((((j * 2 + par) * NHALF + h) * 16) + c) * 2 + 1] = im;
                }
    for (int n2 = 0; n2 < 20; ++n2)
        for (int k1 = 0; k1 < 20; ++k1) {
            const double ang = two_pi * (double)(n2 * k1) / (double)NFFT;     // n2 k1 <= 361: no reduction needed
            hp->tw[(n2 * 20 + k1) * 2 + 0] = cos(ang);
            hp->tw[(n2 * 20 + k1) * 2 + 1] = -sin(ang);
        }
    for (int n = 0; n < NFFT; ++n) hp->win[n] = hwin[n];
    for (int m = 0; m < NMEL; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < NBIN; ++k)
            if (hfb[k * NMEL + m] != 0.f) {
                if (lo < 0) lo = k;
                hi = k;
            }
        const int cnt = lo < 0 ? 0 : hi - lo + 1;
        TAL_CHECK_ARG(cnt <= MAXW, "tal_logmel_plan_init: mel filter %d spans %d bins (max %d)", m, cnt, MAXW);
        hp->mel_lo[m] = lo < 0 ? 0 : lo;
        hp->mel_cnt[m] = cnt;
        for (int i = 0; i < MAXW; ++i) hp->mel_w[m * MAXW + i] = i < cnt ? hfb[(lo + i) * NMEL + m] : 0.f;
    }
    {
        int total = 0;
        for (int m = 0; m < NMEL; ++m) total += (hp->mel_cnt[m] + 3) / 4 * 4;
        hp->mel_compact = total + 4 <= MELC ? 1 : 0;          // (a filterbank with wider supports: weights from the [80][48] table)
        int off = 0;
        for (int i = 0; i < MELC; ++i) hp->mel_wc[i] = 0.f;
        for (int m = 0; m < NMEL; ++m) {
            hp->mel_off[m] = hp->mel_compact ? off : 0;
            if (hp->mel_compact) {
                for (int i = 0; i < hp->mel_cnt[m]; ++i) hp->mel_wc[off + i] = hp->mel_w[m * MAXW + i];
                off += (hp->mel_cnt[m] + 3) / 4 * 4;
            }
        }
    }
    if (hipMemcpyAsync(plan, hp, sizeof(LogmelPlan), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        set_error("tal_logmel_plan_init: cannot upload the plan");
        return TAL_EHIP;
    }
    return TAL_OK;
}

extern "C" size_t tal_logmel_workspace_bytes(int B, int64_t L) {
    const int64_t T = 1 + L / HOP;
    return (size_t)(B * cdiv(T, FFT_FB) + 4) * sizeof(double);      // (12-frame workgroups: the most partial sums any form writes)
}

template <typename AT>
static int logmel_fwd_impl(const void* plan, const AT* audio, int B, int64_t L, float eps, int subtract_mean,
                           float* out, float* mean_out, double* sum_out, void* workspace, size_t workspace_bytes,
                           void* stream, const char* what) {
    TAL_CHECK_ARG(plan && audio && out && workspace, "%s: null pointer", what);
    TAL_CHECK_ARG(B > 0 && L > NFFT / 2, "%s: need B>0 and L>%d for reflect padding (L=%lld)", what, NFFT / 2, (long long)L);
    hipStream_t s = (hipStream_t)stream;
    const int64_t T = 1 + L / HOP;
    // (a workspace sized for 32-frame workgroups only -- the figure before the short-input form existed -- is accepted)
    const size_t need_long = (size_t)(B * cdiv(T, FB) + 4) * sizeof(double);
    if (workspace_bytes < need_long) {
        set_error("%s: workspace %zu < %zu bytes", what, workspace_bytes, tal_logmel_workspace_bytes(B, L));
        return TAL_ENOMEM;
    }
    // the fast-transform form (default) or the matrix form (`logmel_mfma`; a workspace sized before the fast form existed)
    const bool fft = !opt(OPT_LOGMEL_MFMA) && workspace_bytes >= tal_logmel_workspace_bytes(B, L);
    // matrix form, short inputs (fewer than two 32-frame workgroups per CU): 16 frames per workgroup
    const bool short_in = !fft && (int64_t)B * cdiv(T, FB) < 2 * (int64_t)device_cus() &&
                          workspace_bytes >= (size_t)(B * cdiv(T, FB_SHORT) + 4) * sizeof(double);
    const int64_t nblk = cdiv(T, fft ? FFT_FB : (short_in ? FB_SHORT : FB));
    // fast form: persistent workgroups (three per CU) walk the blocks of the whole batch, one partial sum each
    const int64_t fft_wgs = B * nblk < 3 * (int64_t)device_cus() ? B * nblk : 3 * (int64_t)device_cus();
    const int64_t nparts = fft ? fft_wgs : B * nblk;
    double* partial = reinterpret_cast<double*>(workspace);
    float* mean_ws = reinterpret_cast<float*>(partial + nparts + 2);
    {
        // algorithmic HBM bytes: read L samples, write T*80 floats per item
        ProfScope prof(PROF_LOGMEL, (double)B * ((double)L * sizeof(AT) + (double)T * NMEL * 4.0), s);
        const LogmelPlan* pl = reinterpret_cast<const LogmelPlan*>(plan);
        const dim3 grid((unsigned)nblk, (unsigned)B);
        if (fft)
            hipLaunchKernelGGL((logmel_fft_kernel<FFT_FB / 2, AT>), dim3((unsigned)fft_wgs), dim3(128), 0, s, pl, audio, L, T, nblk,
                               (int64_t)B * nblk, eps, out, partial);
        else if (short_in)
            hipLaunchKernelGGL((logmel_kernel<FB_SHORT, AT>), grid, dim3(256), 0, s, pl, audio, L, T, eps, out, partial);
        else
            hipLaunchKernelGGL((logmel_kernel<FB, AT>), grid, dim3(256), 0, s, pl, audio, L, T, eps, out, partial);
    }
    TAL_CHECK_LAUNCH(what);
    hipLaunchKernelGGL(logmel_mean_kernel, dim3(1), dim3(256), 0, s, partial, nparts,
                       (double)B * (double)T * (double)NMEL, mean_out, sum_out, mean_ws);
    TAL_CHECK_LAUNCH(what);
    if (subtract_mean) return launch_subtract(out, (int64_t)B * T * NMEL, mean_ws, s);
    return TAL_OK;
}

extern "C" int tal_logmel_fwd(const void* plan, const float* audio, int B, int64_t L, float eps, int subtract_mean,
                              float* out, float* mean_out, double* sum_out, void* workspace, size_t workspace_bytes,
                              void* stream) {
    return logmel_fwd_impl<float>(plan, audio, B, L, eps, subtract_mean, out, mean_out, sum_out, workspace, workspace_bytes,
                                  stream, "tal_logmel_fwd");
}

extern "C" int tal_logmel_f16_fwd(const void* plan, const void* audio_f16, int B, int64_t L, float eps, int subtract_mean,
                                  float* out, float* mean_out, double* sum_out, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    TAL_CHECK_ARG((reinterpret_cast<uintptr_t>(audio_f16) & 1) == 0, "tal_logmel_f16_fwd: audio must be 2-byte aligned");
    return logmel_fwd_impl<_Float16>(plan, reinterpret_cast<const _Float16*>(audio_f16), B, L, eps, subtract_mean, out,
                                     mean_out, sum_out, workspace, workspace_bytes, stream, "tal_logmel_f16_fwd");
}

extern "C" int tal_subtract_scalar(float* x, int64_t n, const float* mean, void* stream) {
    TAL_CHECK_ARG(x && mean && n >= 0, "tal_subtract_scalar: bad argument");
    TAL_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0, "tal_subtract_scalar: x must be 16-byte aligned");
    return launch_subtract(x, n, mean, (hipStream_t)stream);
}
