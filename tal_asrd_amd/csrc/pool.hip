// Attention-weighted pooling of diarization features per generated token:
// tal/utils/aligned_to_wder_format.py:150-178,203-214 (the consumer of the `attention` /
// `chunkStart` alignments that System.generate_unaligned returns).
//   emb[n]  = sum_s attn[n, s] * feat[cs[n] + s]             (s < min(S, T - cs[n]))
//   vote[n] = arg max over speaker ids of sum_s attn[n, s] * [ids[cs[n] + s] == id]
#include "common.h"

namespace tal {

__global__ __launch_bounds__(128) void attn_pool_kernel(const float* __restrict__ attn, const int64_t* __restrict__ cs,
                                                       const float* __restrict__ feat, int64_t T, int E, int S,
                                                       float* __restrict__ out) {
    const int n = blockIdx.x;
    const int64_t c0 = cs[n];
    const int64_t avail = T - c0;
    const int len = (int)(avail < S ? (avail < 0 ? 0 : avail) : S);
    const float* a = attn + (int64_t)n * S;
    for (int e = threadIdx.x; e < E; e += 128) {
        float acc = 0.f;
        for (int s = 0; s < len; ++s) acc = fmaf(a[s], feat[(c0 + s) * E + e], acc);
        out[(int64_t)n * E + e] = acc;
    }
}

// one workgroup per token; O(S^2) comparisons (S = 357) replace a hash map of speaker ids
__global__ __launch_bounds__(256) void attn_vote_kernel(const float* __restrict__ attn, const int64_t* __restrict__ cs,
                                                       const int32_t* __restrict__ ids, int64_t T, int S,
                                                       int32_t* __restrict__ out_id, float* __restrict__ out_w) {
    extern __shared__ float sm[];  // [S] weights, [S] ids (as int)
    float* w = sm;
    int* id = reinterpret_cast<int*>(sm + S);
    __shared__ float bw[4];
    __shared__ int bid[4], bpos[4];
    const int n = blockIdx.x;
    const int64_t c0 = cs[n];
    const int64_t avail = T - c0;
    const int len = (int)(avail < S ? (avail < 0 ? 0 : avail) : S);
    for (int s = threadIdx.x; s < len; s += 256) {
        w[s] = attn[(int64_t)n * S + s];
        id[s] = ids[c0 + s];
    }
    __syncthreads();
    float best = -INFINITY;
    int bi = -1, bp = 0x7fffffff;
    for (int s = threadIdx.x; s < len; s += 256) {
        float tot = 0.f;
        int first = s;
        for (int q = 0; q < len; ++q)
            if (id[q] == id[s]) {
                tot += w[q];
                if (q < first) first = q;
            }
        if (first == s && (tot > best || (tot == best && s < bp))) {  // one candidate per distinct id
            best = tot;
            bi = id[s];
            bp = s;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64), op = __shfl_xor(bp, off, 64);
        if (ov > best || (ov == best && op < bp)) { best = ov; bi = oi; bp = op; }
    }
    if (lane == 0) { bw[wv] = best; bid[wv] = bi; bpos[wv] = bp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q)
            if (bw[q] > best || (bw[q] == best && bpos[q] < bp)) { best = bw[q]; bi = bid[q]; bp = bpos[q]; }
        out_id[n] = bi;
        if (out_w) out_w[n] = best;
    }
}

}  // namespace tal

using namespace tal;

extern "C" int tal_attn_pool_fwd(const float* attn, const int64_t* chunk_start, const float* feat, int64_t T, int E,
                                 int N, int S, float* out, void* stream) {
    TAL_CHECK_ARG(attn && chunk_start && feat && out, "tal_attn_pool_fwd: null pointer");
    TAL_CHECK_ARG(T > 0 && E > 0 && N >= 0 && S > 0, "tal_attn_pool_fwd: bad shape");
    if (N == 0) return TAL_OK;
    hipLaunchKernelGGL(attn_pool_kernel, dim3((unsigned)N), dim3(128), 0, (hipStream_t)stream, attn, chunk_start, feat, T,
                       E, S, out);
    TAL_CHECK_LAUNCH("tal_attn_pool_fwd");
    return TAL_OK;
}

extern "C" int tal_attn_vote_fwd(const float* attn, const int64_t* chunk_start, const int32_t* ids, int64_t T, int N,
                                 int S, int32_t* out_id, float* out_weight, void* stream) {
    TAL_CHECK_ARG(attn && chunk_start && ids && out_id, "tal_attn_vote_fwd: null pointer");
    TAL_CHECK_ARG(T > 0 && N >= 0 && S > 0 && S <= 8192, "tal_attn_vote_fwd: bad shape");
    if (N == 0) return TAL_OK;
    hipLaunchKernelGGL(attn_vote_kernel, dim3((unsigned)N), dim3(256), (size_t)S * 8, (hipStream_t)stream, attn,
                       chunk_start, ids, T, S, out_id, out_weight);
    TAL_CHECK_LAUNCH("tal_attn_vote_fwd");
    return TAL_OK;
}
