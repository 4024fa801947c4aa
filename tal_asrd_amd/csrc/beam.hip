// Beam-search candidate selection: System.generate, tal/asr/system.py:141-160.
//   total = logprobs + scores[row]; rows of finished beams -> -inf; per batch item flatten the
//   cur_beam x V candidates and take the top-k (values descending, lowest flat index first on ties).
// One workgroup per batch item; k (the beam width) is small, so the top-k is k rounds of a
// block-wide arg-max over the candidates that have not been taken yet.
#include "common.h"

namespace tal {

constexpr int TOPK_MAX = 64;

__global__ __launch_bounds__(256) void beam_topk_kernel(const float* __restrict__ logprobs,
                                                       const float* __restrict__ row_score,
                                                       const uint8_t* __restrict__ row_done, int cur_beam, int V,
                                                       int k, float* __restrict__ out_val,
                                                       int64_t* __restrict__ out_idx) {
    __shared__ float s_val[4];
    __shared__ int64_t s_idx[4];
    __shared__ int64_t taken[TOPK_MAX];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t n = (int64_t)cur_beam * V;
    const float* lp = logprobs + (int64_t)b * n;
    for (int j = 0; j < k; ++j) {
        float best = -INFINITY;
        int64_t bi = INT64_MAX;
        for (int64_t i = tid; i < n; i += 256) {
            const int r = (int)(i / V);
            const int row = b * cur_beam + r;
            float v = lp[i] + (row_score ? row_score[row] : 0.f);
            if (row_done && row_done[row]) v = -INFINITY;
            bool skip = false;
            for (int t = 0; t < j; ++t) skip |= (taken[t] == i);
            if (!skip && (v > best || (v == best && i < bi))) {
                best = v;
                bi = i;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int64_t oi = __shfl_xor(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_val[w] = best;
            s_idx[w] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int q = 1; q < 4; ++q)
                if (s_val[q] > best || (s_val[q] == best && s_idx[q] < bi)) {
                    best = s_val[q];
                    bi = s_idx[q];
                }
            // every candidate already taken or -inf everywhere: fall back to the lowest untaken index
            if (bi == INT64_MAX) {
                bi = 0;
                for (bool clash = true; clash;) {
                    clash = false;
                    for (int t = 0; t < j; ++t)
                        if (taken[t] == bi) {
                            ++bi;
                            clash = true;
                        }
                }
                best = -INFINITY;
            }
            taken[j] = bi;
            out_val[(int64_t)b * k + j] = best;
            out_idx[(int64_t)b * k + j] = bi;
        }
        __syncthreads();
    }
}

}  // namespace tal

using namespace tal;

extern "C" int tal_beam_topk(const float* logprobs, const float* row_score, const uint8_t* row_done, int B,
                             int cur_beam, int V, int k, float* out_val, int64_t* out_idx, void* stream) {
    TAL_CHECK_ARG(logprobs && out_val && out_idx, "tal_beam_topk: null pointer");
    TAL_CHECK_ARG(B > 0 && cur_beam > 0 && V > 0 && k > 0 && k <= TOPK_MAX && (int64_t)k <= (int64_t)cur_beam * V,
                  "tal_beam_topk: bad shape B=%d beams=%d V=%d k=%d", B, cur_beam, V, k);
    hipLaunchKernelGGL(beam_topk_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logprobs, row_score,
                       row_done, cur_beam, V, k, out_val, out_idx);
    TAL_CHECK_LAUNCH("tal_beam_topk");
    return TAL_OK;
}
