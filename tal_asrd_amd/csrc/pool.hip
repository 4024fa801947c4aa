// Attention-weighted pooling of diarization features and speaker votes per generated token / word / utterance:
// tal/utils/aligned_to_wder_format.py:150-214 (unaligned: per token and per word), :321-353 (aligned: majority vote),
// the consumer of the `attention` / `chunkStart` alignments that System.generate_unaligned returns.
//   emb[n]   = sum_s attn[n, s] * feat[win(n)][s]
//   vote[g]  = the speaker id with the largest summed attention over the tokens of group g
//   major[g] = the most frequent speaker id in ids[start_g : end_g]
// win(n) is the python slice x[cs : cs + S] the reference takes (negative starts wrap from the end, ends are
// clamped, an empty slice contributes nothing), attention is truncated to the slice length (aw[:len(chunk)]).
// `half_mode` reproduces the reference's arithmetic, which rounds attention and features to fp16
// (`.half()`, :74,159,206) before a matmul with fp32 accumulation and an fp16 result.
#include "common.h"

namespace tal {

// python slice [c0 : c0 + S] of a length-T sequence -> (start, len)
__device__ __forceinline__ void py_window(int64_t c0, int S, int64_t T, int64_t& start, int& len) {
    int64_t a = c0, b = c0 + S;
    if (a < 0) { a += T; if (a < 0) a = 0; } else if (a > T) a = T;
    if (b < 0) { b += T; if (b < 0) b = 0; } else if (b > T) b = T;
    start = a;
    len = b > a ? (int)(b - a) : 0;
}

__device__ __forceinline__ float round_f16(float x) { return (float)(_Float16)x; }

template <bool HALF>
__global__ __launch_bounds__(128) void attn_pool_kernel(const float* __restrict__ attn, const int64_t* __restrict__ cs,
                                                       const float* __restrict__ feat, int64_t T, int E, int S,
                                                       float* __restrict__ out) {
    const int n = blockIdx.x;
    int64_t c0;
    int len;
    py_window(cs[n], S, T, c0, len);
    const float* a = attn + (int64_t)n * S;
    for (int e = threadIdx.x; e < E; e += 128) {
        float acc = 0.f;
        for (int s = 0; s < len; ++s) {
            const float av = HALF ? round_f16(a[s]) : a[s];
            const float fv = HALF ? round_f16(feat[(c0 + s) * E + e]) : feat[(c0 + s) * E + e];
            acc = fmaf(av, fv, acc);
        }
        out[(int64_t)n * E + e] = HALF ? round_f16(acc) : acc;
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
    int64_t c0;
    int len;
    py_window(cs[n], S, T, c0, len);
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

// Group votes with a dense per-speaker table in LDS (num_ids doubles + num_ids ints; 6008 speakers = 72 KB).
//   MODE 0: attention-weighted vote over the tokens [seg[g], seg[g+1]) of group g (aligned_to_wder_format.py:150-196).
//           Sums are float64 like the reference's python floats; in half_mode every addend is an fp16 value, the
//           partial sums stay below 2^11 with a 2^-34 granularity, so float64 addition is exact and the atomic
//           order cannot matter.  Ties go to the id that appeared LAST for the first time
//           (`sorted(items, key=weight)[-1]` is a stable sort over a dict in insertion order).
//   MODE 1: majority vote over ids[seg[2g] : seg[2g+1]] (python slice, :330-333); ties go to the id that appeared
//           first (Counter.most_common(1) = max() over insertion order).
template <int MODE>
__global__ __launch_bounds__(256) void group_vote_kernel(const float* __restrict__ attn, const int64_t* __restrict__ cs,
                                                        const int32_t* __restrict__ ids, int64_t T, int S,
                                                        const int64_t* __restrict__ seg, int num_ids, int half_mode,
                                                        int32_t* __restrict__ out_id, double* __restrict__ out_w) {
    extern __shared__ double tab[];                       // [num_ids] sums (counts for MODE 1)
    int* first = reinterpret_cast<int*>(tab + num_ids);   // [num_ids] first-appearance position
    __shared__ double bw[4];
    __shared__ int bid[4], bpos[4];
    const int g = blockIdx.x;
    for (int i = threadIdx.x; i < num_ids; i += 256) {
        tab[i] = 0.0;
        first[i] = 0x7fffffff;
    }
    __syncthreads();
    if (MODE == 0) {
        const int64_t t0 = seg[g], t1 = seg[g + 1];
        for (int64_t t = t0; t < t1; ++t) {
            int64_t c0;
            int len;
            py_window(cs[t], S, T, c0, len);
            for (int s = threadIdx.x; s < len; s += 256) {
                const int id = ids[c0 + s];
                if (id < 0 || id >= num_ids) continue;
                const float a = attn[t * S + s];
                atomicAdd(&tab[id], (double)(half_mode ? round_f16(a) : a));
                atomicMin(&first[id], (int)((t - t0) * S + s));
            }
        }
    } else {
        int64_t a = seg[2 * g], b = seg[2 * g + 1];
        if (a < 0) { a += T; if (a < 0) a = 0; } else if (a > T) a = T;
        if (b < 0) { b += T; if (b < 0) b = 0; } else if (b > T) b = T;
        for (int64_t p = a + threadIdx.x; p < b; p += 256) {
            const int id = ids[p];
            if (id < 0 || id >= num_ids) continue;
            atomicAdd(&tab[id], 1.0);
            atomicMin(&first[id], (int)(p - a));
        }
    }
    __syncthreads();
    double best = -1.0;
    int bi = -1, bp = MODE == 0 ? -1 : 0x7fffffff;
    auto better = [](double v, int p, double bv, int bpv) {
        return v > bv || (v == bv && (MODE == 0 ? p > bpv : p < bpv));
    };
    for (int i = threadIdx.x; i < num_ids; i += 256)
        if (first[i] != 0x7fffffff && better(tab[i], first[i], best, bp)) {
            best = tab[i];
            bi = i;
            bp = first[i];
        }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64), op = __shfl_xor(bp, off, 64);
        if (oi >= 0 && (bi < 0 || better(ov, op, best, bp))) { best = ov; bi = oi; bp = op; }
    }
    if (lane == 0) { bw[wv] = best; bid[wv] = bi; bpos[wv] = bp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q)
            if (bid[q] >= 0 && (bi < 0 || better(bw[q], bpos[q], best, bp))) { best = bw[q]; bi = bid[q]; bp = bpos[q]; }
        out_id[g] = bi;                 // -1: empty group
        if (out_w) out_w[g] = bi < 0 ? 0.0 : best;
    }
}

template <int MODE>
static int launch_group_vote(const char* what, const float* attn, const int64_t* cs, const int32_t* ids, int64_t T, int S,
                             const int64_t* seg, int G, int num_ids, int half_mode, int32_t* out_id, double* out_w,
                             hipStream_t s) {
    const size_t lds = (size_t)num_ids * 12 + 8;
    TAL_CHECK_ARG(lds <= 160 * 1024 - 256, "%s: %d speaker ids need %zu bytes of LDS (max 160 KB)", what, num_ids, lds);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&group_vote_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
        set_error("%s: cannot reserve %zu bytes of LDS", what, lds);
        return TAL_EHIP;
    }
    hipLaunchKernelGGL((group_vote_kernel<MODE>), dim3((unsigned)G), dim3(256), lds, s, attn, cs, ids, T, S, seg, num_ids,
                       half_mode, out_id, out_w);
    TAL_CHECK_LAUNCH(what);
    return TAL_OK;
}

}  // namespace tal

using namespace tal;

extern "C" int tal_attn_pool_fwd(const float* attn, const int64_t* chunk_start, const float* feat, int64_t T, int E,
                                 int N, int S, int half_mode, float* out, void* stream) {
    TAL_CHECK_ARG(attn && chunk_start && feat && out, "tal_attn_pool_fwd: null pointer");
    TAL_CHECK_ARG(T > 0 && E > 0 && N >= 0 && S > 0, "tal_attn_pool_fwd: bad shape");
    if (N == 0) return TAL_OK;
    if (half_mode)
        hipLaunchKernelGGL(attn_pool_kernel<true>, dim3((unsigned)N), dim3(128), 0, (hipStream_t)stream, attn, chunk_start,
                           feat, T, E, S, out);
    else
        hipLaunchKernelGGL(attn_pool_kernel<false>, dim3((unsigned)N), dim3(128), 0, (hipStream_t)stream, attn, chunk_start,
                           feat, T, E, S, out);
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

extern "C" int tal_attn_vote_groups_fwd(const float* attn, const int64_t* chunk_start, const int32_t* ids, int64_t T,
                                        int S, const int64_t* group_offsets, int G, int num_ids, int half_mode,
                                        int32_t* out_id, double* out_weight, void* stream) {
    TAL_CHECK_ARG(attn && chunk_start && ids && group_offsets && out_id, "tal_attn_vote_groups_fwd: null pointer");
    TAL_CHECK_ARG(T > 0 && S > 0 && G >= 0 && num_ids > 0, "tal_attn_vote_groups_fwd: bad shape");
    if (G == 0) return TAL_OK;
    return launch_group_vote<0>("tal_attn_vote_groups_fwd", attn, chunk_start, ids, T, S, group_offsets, G, num_ids,
                                half_mode, out_id, out_weight, (hipStream_t)stream);
}

extern "C" int tal_majority_vote_fwd(const int32_t* ids, int64_t T, const int64_t* ranges, int G, int num_ids,
                                     int32_t* out_id, double* out_count, void* stream) {
    TAL_CHECK_ARG(ids && ranges && out_id, "tal_majority_vote_fwd: null pointer");
    TAL_CHECK_ARG(T > 0 && G >= 0 && num_ids > 0, "tal_majority_vote_fwd: bad shape");
    if (G == 0) return TAL_OK;
    return launch_group_vote<1>("tal_majority_vote_fwd", nullptr, nullptr, ids, T, 0, ranges, G, num_ids, 0, out_id,
                                out_count, (hipStream_t)stream);
}
