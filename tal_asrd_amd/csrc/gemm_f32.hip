// fp32 dense layer on the CDNA4 matrix cores: Y = epilogue(X . W^T + b).
//
// Covers nn.Linear and the 1x1 Conv1d pair of TDSBlock (tal/asr/models.py:312-318,
// 86.9 % of the encoder's MACs), the encoder projections (:100,131), the SD / speaker
// heads (:421-422,143-146) and the decoder's projections / FFN (:493-499).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bit-for-bit an fmaf chain),
// so results stay inside the 1e-3 fp32 logit tolerance of BASELINE.json without any
// reduced-precision trick.  Roofline: 157.3 TFLOP/s (fp32 matrix peak of MI355X).
//
// Tiling: 128(M) x 160(N) x 32(K) per 256-thread workgroup; wave w owns rows
// [32w, 32w+32) x all 160 columns (5 accumulators of 32x32).  160 divides every TDS width
// (800 = 5*160, 1120 = 7*160, 1440 = 9*160) so the hot GEMMs have no N-tail waste.
// LDS rows are padded to 36 floats: ds_read_b128 of 16 rows x 4 floats is conflict-free.
#include "common.h"

namespace tal {

constexpr int BM = 128;
constexpr int BN = 160;
constexpr int BK = 32;
constexpr int LDS_LD = 36;
constexpr int NSUB = BN / 32;
constexpr int A_LOADS = BM * BK / 4 / 256;  // 4 float4 per thread
constexpr int B_LOADS = BN * BK / 4 / 256;  // 5 float4 per thread

template <int MODE>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ res, float alpha, int64_t M, int N,
                                                         int K, float* __restrict__ Y, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_LD];
    float* As = lds;
    float* Bs = lds + BM * LDS_LD;

    // XCD-aware bijective remap: XCD x (= blockIdx % 8) walks a contiguous range of logical
    // tiles, N-tiles of one M-tile first, so the A panel is re-read from that XCD's L2.
    const unsigned nb = gridDim.x;
    const unsigned bid = blockIdx.x;
    const unsigned xcd = bid & 7u, q = nb >> 3, r = nb & 7u;
    const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int64_t m0 = (int64_t)(logical / (unsigned)tiles_n) * BM;
    const int n0 = (int)(logical % (unsigned)tiles_n) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = wave_id();
    const int lrow = tid >> 3;  // 0..31
    const int lc4 = tid & 7;    // float4 column inside the 32-wide K slab

    // tile-relative 32-bit offsets (rows past the end are clamped; their results are never stored)
    const float* At = A + m0 * (int64_t)K + lc4 * 4;
    const float* Wt = W + (int64_t)n0 * K + lc4 * 4;
    const int a_rows = (int)((M - m0) < BM ? (M - m0) : BM) - 1;
    const int b_rows = ((N - n0) < BN ? (N - n0) : BN) - 1;
    int a_off[A_LOADS], b_off[B_LOADS];
#pragma unroll
    for (int p = 0; p < A_LOADS; ++p) a_off[p] = min(lrow + 32 * p, a_rows) * K;
#pragma unroll
    for (int p = 0; p < B_LOADS; ++p) b_off[p] = min(lrow + 32 * p, b_rows) * K;

    f32x16 acc[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // K tail (K % 32 != 0, only in small test models): out-of-range float4 columns read as zero
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[A_LOADS], rb[B_LOADS];
    {
        // branch-free: load from a clamped (always valid) K offset, then select
        const bool in = lc4 * 4 < K;
        const int ko = in ? 0 : -lc4 * 4;
#pragma unroll
        for (int p = 0; p < A_LOADS; ++p) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(At + ko + a_off[p]);
            ra[p] = in ? v : zero4;
        }
#pragma unroll
        for (int p = 0; p < B_LOADS; ++p) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Wt + ko + b_off[p]);
            rb[p] = in ? v : zero4;
        }
    }

    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
    const int nk = (K + BK - 1) / BK;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int p = 0; p < A_LOADS; ++p)
            *reinterpret_cast<f32x4*>(&As[(lrow + 32 * p) * LDS_LD + lc4 * 4]) = ra[p];
#pragma unroll
        for (int p = 0; p < B_LOADS; ++p)
            *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * p) * LDS_LD + lc4 * 4]) = rb[p];
        __syncthreads();
        if (kt + 1 < nk) {
            const bool in = (kt + 1) * BK + lc4 * 4 < K;
            const int ko = in ? (kt + 1) * BK : -lc4 * 4;
#pragma unroll
            for (int p = 0; p < A_LOADS; ++p) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(At + ko + a_off[p]);
                ra[p] = in ? v : zero4;
            }
#pragma unroll
            for (int p = 0; p < B_LOADS; ++p) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(Wt + ko + b_off[p]);
                rb[p] = in ? v : zero4;
            }
        }
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&As[w * 32 * LDS_LD + frag_off + kk * 8]);
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[j * 32 * LDS_LD + frag_off + kk * 8]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int colb = lane & 31;
    const int rowb = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        const int col = n0 + j * 32 + colb;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = m0 + w * 32 + rowb + (e & 3) + 8 * (e >> 2);
            if (row < M) {
                float v = acc[j][e] + bv;
                if (MODE == 1) v = fmaxf(v, 0.f);
                if (MODE == 2) v = res[row * N + col] + alpha * v;
                Y[row * N + col] = v;
            }
        }
    }
}

int launch_linear(const float* x, const float* w, const float* b, const float* res, float alpha, int mode, int64_t M,
                  int N, int K, float* y, hipStream_t s) {
    TAL_CHECK_ARG(x && w && y, "tal_linear_fwd: null pointer");
    TAL_CHECK_ARG(M >= 0 && N > 0 && K > 0, "tal_linear_fwd: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    TAL_CHECK_ARG(K % 4 == 0, "tal_linear_fwd: K=%d must be a multiple of 4", K);
    TAL_CHECK_ARG(mode >= 0 && mode <= 2, "tal_linear_fwd: mode %d", mode);
    TAL_CHECK_ARG(mode != 2 || res, "tal_linear_fwd: mode 2 needs a residual");
    if (M == 0) return TAL_OK;
    const int tiles_n = (int)cdiv(N, BN);
    const int64_t tiles_m = cdiv(M, BM);
    const int64_t nb = tiles_m * tiles_n;
    TAL_CHECK_ARG(nb < (1ll << 31), "tal_linear_fwd: grid too large");
    dim3 grid((unsigned)nb), block(256);
    ProfScope prof(PROF_GEMM, 2.0 * (double)M * (double)N * (double)K, s);
    switch (mode) {
        case 0: hipLaunchKernelGGL(gemm_nt_f32_kernel<0>, grid, block, 0, s, x, w, b, res, alpha, M, N, K, y, tiles_n); break;
        case 1: hipLaunchKernelGGL(gemm_nt_f32_kernel<1>, grid, block, 0, s, x, w, b, res, alpha, M, N, K, y, tiles_n); break;
        default: hipLaunchKernelGGL(gemm_nt_f32_kernel<2>, grid, block, 0, s, x, w, b, res, alpha, M, N, K, y, tiles_n); break;
    }
    TAL_CHECK_LAUNCH("tal_linear_fwd");
    return TAL_OK;
}

}  // namespace tal
