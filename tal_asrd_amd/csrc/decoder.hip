// Transformer decoder side of ASRModel (tal/asr/models.py:203-289, ModRZTXDecoderLayer
// :488-528, torch.nn.MultiheadAttention, tal/modules.py:41-64).
//
// Every contraction (in/out projections, QK^T, PV, FFN, LM head) runs through the fp32-MFMA
// dense-layer kernel (gemm_f32.hip), batched over (batch item, head) with strides so no head
// split / merge copies exist.  V is produced already transposed (V^T = W_v . x^T is just
// another NT product) so PV is NT as well; its bias is added after PV, which is exact
// because softmax rows sum to one.  The only non-GEMM kernels are the token embedding, the
// masked row softmax (which also emits the head-averaged probabilities the decode loop
// steers by, system.py:392-408) and small row utilities.
#include <atomic>
#include <chrono>

#include "decode_bodies.h"

namespace tal {

static inline int64_t pad4(int64_t n) { return (n + 3) & ~(int64_t)3; }

__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ emb,
                                                   const float* __restrict__ proj, const float* __restrict__ pe,
                                                   float* __restrict__ out, int U, int V, int E0, int D) {
    embed_body(tokens, emb, proj, pe, out, U, V, E0, D, blockIdx.x);
}

// several prefixes in one launch (the merged decode step): blockIdx.y picks the prefix, rows past its length return
struct EmbedMulti {
    const int64_t* tokens[TAL_GROUP_MAX];
    float* out[TAL_GROUP_MAX];
    int U[TAL_GROUP_MAX];
};
__global__ __launch_bounds__(256) void embed_multi_kernel(const EmbedMulti m, const float* __restrict__ emb, const float* __restrict__ proj,
                                                         const float* __restrict__ pe, int V, int E0, int D) {
    const int i = blockIdx.y;
    if ((int)blockIdx.x >= m.U[i]) return;
    embed_body(m.tokens[i], emb, proj, pe, m.out[i], m.U[i], V, E0, D, blockIdx.x);
}

__global__ __launch_bounds__(256) void add_positional_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                            float* __restrict__ out, int U, int D, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % D);
    const int u = (int)((i / D) % U);
    out[i] = x[i] + pe[(int64_t)u * D + d];
}

// ---- masked softmax over the key axis, one wave per (b, u), heads looped ----------------------
// scores [B, H, U, S4] in place -> probabilities (pad columns zeroed); optional additive float
// mask [U, S], optional key-padding mask [B, S] (non-zero -> -inf), optional head average
// avg [B, U, S] = mean_h P (MultiheadAttention's returned weights, models.py:517-519).
__global__ __launch_bounds__(256) void attn_softmax_kernel(float* __restrict__ scores, const float* __restrict__ mask,
                                                          const uint8_t* __restrict__ kpm, float* __restrict__ avg,
                                                          int B, int H, int U, int S, int S4, float* __restrict__ vt_pad,
                                                          int E) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * U) return;
    const int b = (int)(row / U), u = (int)(row % U);
    if (vt_pad && u == 0)   // pad columns of this batch item's V^T (consumed by the P.V product that follows)
        for (int i = lane; i < E * (S4 - S); i += 64) vt_pad[((int64_t)b * E + i / (S4 - S)) * S4 + S + i % (S4 - S)] = 0.f;
    const float* mrow = mask ? mask + (int64_t)u * S : nullptr;
    const uint8_t* krow = kpm ? kpm + (int64_t)b * S : nullptr;
    float* arow = avg ? avg + row * S : nullptr;
    const float inv_h = 1.0f / (float)H;
    for (int h = 0; h < H; ++h) {
        float* p = scores + (((int64_t)b * H + h) * U + u) * S4;
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) {
            float v = p[s];
            if (mrow) v += mrow[s];
            if (krow && krow[s]) v = -INFINITY;
            m = fmaxf(m, v);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        float sum = 0.f;
        for (int s = lane; s < S; s += 64) {
            float v = p[s];
            if (mrow) v += mrow[s];
            if (krow && krow[s]) v = -INFINITY;
            sum += expf(v - m);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        for (int s = lane; s < S; s += 64) {
            float v = p[s];
            if (mrow) v += mrow[s];
            if (krow && krow[s]) v = -INFINITY;
            const float pr = expf(v - m) / sum;
            p[s] = pr;
            if (arow) {
                float a = (h == 0 ? 0.f : arow[s]) + pr;
                if (h == H - 1) a *= inv_h;
                arow[s] = a;
            }
        }
        if (lane < S4 - S) p[S + lane] = 0.f;
    }
}

// Latency-oriented variant for the decode loops (S <= 512, H <= 8): one workgroup per (b, u), wave h
// owns head h and keeps its row in registers (one global read, one write); the head average goes
// through LDS and is summed in head order, exactly like the kernel above.
constexpr int SM_NR = 8;
__global__ __launch_bounds__(512) void attn_softmax_small_kernel(float* __restrict__ scores,
                                                                const float* __restrict__ mask,
                                                                const uint8_t* __restrict__ kpm,
                                                                float* __restrict__ avg, int H, int U, int S, int S4,
                                                                float* __restrict__ vt_pad, int E) {
    extern __shared__ float ph[];  // [H][S]
    const int lane = threadIdx.x & 63;
    const int h = threadIdx.x >> 6;
    const int64_t row = blockIdx.x;
    const int b = (int)(row / U), u = (int)(row % U);
    if (vt_pad && u == 0)   // pad columns of this batch item's V^T (consumed by the P.V product that follows)
        for (int i = threadIdx.x; i < E * (S4 - S); i += blockDim.x)
            vt_pad[((int64_t)b * E + i / (S4 - S)) * S4 + S + i % (S4 - S)] = 0.f;
    float* p = scores + (((int64_t)b * H + h) * U + u) * S4;
    float v[SM_NR];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < SM_NR; ++r) {
        const int s = lane + 64 * r;
        float x = -INFINITY;
        if (s < S) {
            x = p[s];
            if (mask) x += mask[(int64_t)u * S + s];
            if (kpm && kpm[(int64_t)b * S + s]) x = -INFINITY;
        }
        v[r] = x;
        m = fmaxf(m, x);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < SM_NR; ++r) {
        const int s = lane + 64 * r;
        v[r] = s < S ? expf(v[r] - m) : 0.f;
        sum += v[r];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
#pragma unroll
    for (int r = 0; r < SM_NR; ++r) {
        const int s = lane + 64 * r;
        if (s < S) {
            const float pr = v[r] / sum;
            p[s] = pr;
            if (avg) ph[h * S + s] = pr;
        }
    }
    if (lane < S4 - S) p[S + lane] = 0.f;
    if (avg) {
        __syncthreads();
        const float inv_h = 1.0f / (float)H;
        for (int s = threadIdx.x; s < S; s += blockDim.x) {
            float a = 0.f;
            for (int q = 0; q < H; ++q) a += ph[q * S + s];
            avg[row * S + s] = a * inv_h;
        }
    }
}

// one workgroup per row: the M = 1 rows of the greedy / beam loops would otherwise run on one wave
__global__ __launch_bounds__(256) void log_softmax_row_block_kernel(const float* __restrict__ x, int N,
                                                                   float* __restrict__ out) {
    __shared__ float red[4];
    const float* xr = x + (int64_t)blockIdx.x * N;
    float* orow = out + (int64_t)blockIdx.x * N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < N; i += 256) m = fmaxf(m, xr[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) sum += expf(xr[i] - m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) red[w] = sum;
    __syncthreads();
    const float lse = logf((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = threadIdx.x; i < N; i += 256) orow[i] = (xr[i] - m) - lse;
}

// Greedy step epilogue in one launch and one result buffer: token = argmax(log_softmax(logits)) with exactly the
// arithmetic (and summation order) of log_softmax_row_block_kernel + argmax_row_block_kernel, and the new token's
// cross-attention row averaged over the layers in numpy's order, ((l0 + l1) + l2) + ... then / n_layers.
// out[0] = token (int32 bits), out[1 .. S] = attention.
// The rows may come per head (H > 1, rows head_stride apart): a layer's row is then (sum over heads in head order) * (1 / H),
// the arithmetic of the softmax kernels' head average.
__global__ __launch_bounds__(256) void greedy_pick_kernel(const float* __restrict__ x, int N, const float* __restrict__ attn,
                                                         int n_layers, int64_t layer_stride, int H, int64_t head_stride, int S,
                                                         float* __restrict__ out, int64_t* __restrict__ token_out,
                                                         const float* __restrict__ bias = nullptr) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inv_h = 1.0f / (float)H;
    for (int i = threadIdx.x; i < S; i += 256) {
        float a = 0.f;
        for (int l = 0; l < n_layers; ++l) {
            const float* r = attn + l * layer_stride + i;
            float al = r[0];
            if (H > 1) {
                for (int h = 1; h < H; ++h) al += r[h * head_stride];
                al *= inv_h;
            }
            a = l == 0 ? al : a + al;
        }
        out[1 + i] = a / (float)n_layers;
    }
    float m = -INFINITY;
    for (int i = threadIdx.x; i < N; i += 256) m = fmaxf(m, x[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) sum += expf(x[i] - m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) red[w] = sum;
    __syncthreads();
    const float lse = logf((red[0] + red[1]) + (red[2] + red[3]));
    __syncthreads();
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < N; i += 256) {
        float v = (x[i] - m) - lse;
        if (bias) v += bias[i];          // (system.py:383-384: the LM's weighted log-probabilities on top of the decoder's)
        if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        red[w] = best;
        redi[w] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q)
            if (red[q] > best || (red[q] == best && redi[q] < bi)) {
                best = red[q];
                bi = redi[q];
            }
        bi = bi == 0x7fffffff ? 0 : bi;
        out[0] = __int_as_float(bi);
        if (token_out) *token_out = bi;          // appended to the device-side prefix: the next step needs no upload
    }
}

__global__ __launch_bounds__(256) void lm_pick_kernel(const LmPickArgs q, const float* __restrict__ proj_t, int E, int K0,
                                                     const float* __restrict__ emb, int V, int n_layers, int H) {
    lm_pick_body(q, proj_t, E, K0, emb, V, n_layers, H, blockIdx.x, gridDim.x);
}
// the picks of several sessions in one launch: blockIdx.y = session (own hidden row, attention rows, partials, ticket, result)
__global__ __launch_bounds__(256) void lm_pick_multi_kernel(const ArgPack<LmPickArgs> p, const float* __restrict__ proj_t, int E, int K0,
                                                           const float* __restrict__ emb, int V, int n_layers, int H) {
    lm_pick_body(p.a[blockIdx.y], proj_t, E, K0, emb, V, n_layers, H, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, int64_t M, int N,
                                                              float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * N;
    float* orow = out + row * N;
    float m = -INFINITY;
    for (int i = lane; i < N; i += 64) m = fmaxf(m, xr[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float sum = 0.f;
    for (int i = lane; i < N; i += 64) sum += expf(xr[i] - m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float lse = logf(sum);
    for (int i = lane; i < N; i += 64) orow[i] = (xr[i] - m) - lse;
}

__global__ void transpose_kernel(const float* __restrict__ x, int R, int Cc, float* __restrict__ y) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        if (r < R && c < Cc) t[i][threadIdx.x] = x[(int64_t)r * Cc + c];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < R && c < Cc) y[(int64_t)c * R + r] = t[threadIdx.x][i];
    }
}

// ---- multi-head attention built from batched GEMMs ------------------------------------------
struct MhaBufs {
    float* q;       // [B, U, E]
    float* k;       // [B, S, E]      (unused when a cache is given)
    float* vt;      // [B, E, S4]     (unused when a cache is given)
    float* scores;  // [B, H, U, S4]
    float* ctx;     // [B, U, E]
};

// K = x W_k^T + b_k and V^T = W_v . x^T (bias of V added after P.V) of `src` [B, S, E]
static int project_kv(const float* in_w, const float* in_b, const float* src, int B, int S, int E, float* k, float* vt,
                      bool zero_pad, hipStream_t s) {
    const int64_t S4 = pad4(S);
    int rc = 0;
    if (k) {
        rc = launch_linear(src, in_w + (int64_t)E * E, in_b + E, nullptr, 0.f, 0, (int64_t)B * S, E, E, k, s);
        if (rc) return rc;
    }
    if (zero_pad && S4 != S && hipMemsetAsync(vt, 0, (size_t)B * E * S4 * sizeof(float), s) != hipSuccess) {
        set_error("mha: memset failed");
        return TAL_EHIP;
    }
    GemmArgs g = {};
    g.A = in_w + 2 * (int64_t)E * E; g.W = src; g.Y = vt;   // V^T[b] = W_v . src_b^T
    g.M = E; g.N = S; g.K = E;
    g.lda = E; g.ldw = E; g.ldy = S4;
    g.nb2 = 1; g.w_s1 = (int64_t)S * E; g.y_s1 = (int64_t)E * S4;
    return launch_gemm(g, 0, B, s);
}

// out = res + alpha * out_proj(softmax(q k^T + masks) v) given projected q [B,U,*] (row pitch ldq, already
// scaled), k [B,S,*] (pitch ldk) and V^T [B,E,S4].  vt_pad != NULL: the softmax kernel also zeroes the pad
// columns of V^T (saves a memset launch per layer and step).
static int mha_core(const float* in_b, const float* out_w, const float* out_b, const float* q, int64_t ldq,
                    const float* k, int64_t ldk, const float* vt, float* vt_pad, int B, int U, int S, int E, int H,
                    const float* attn_mask, const uint8_t* kpm, const float* res, float alpha, float* out, float* avg,
                    const MhaBufs& bf, hipStream_t s) {
    const int hd = E / H;
    const int64_t S4 = pad4(S);
    GemmArgs g = {};
    g.A = q; g.W = k; g.Y = bf.scores;
    g.M = U; g.N = S; g.K = hd;
    g.lda = ldq; g.ldw = ldk; g.ldy = S4;
    g.nb2 = H;
    g.a_s1 = (int64_t)U * ldq; g.a_s2 = hd;
    g.w_s1 = (int64_t)S * ldk; g.w_s2 = hd;
    g.y_s1 = (int64_t)H * U * S4; g.y_s2 = (int64_t)U * S4;
    int rc = launch_gemm(g, 0, B * H, s);
    if (rc) return rc;
    if (S4 == S) vt_pad = nullptr;
    {
        ProfScope prof(PROF_OTHER, (double)B * H * U * S * 8.0, s);
        if (S <= 64 * SM_NR && H <= 8)
            hipLaunchKernelGGL(attn_softmax_small_kernel, dim3((unsigned)(B * U)), dim3(64 * H),
                               avg ? (size_t)H * S * sizeof(float) : 0, s, bf.scores, attn_mask, kpm, avg, H, U, S,
                               (int)S4, vt_pad, E);
        else
            hipLaunchKernelGGL(attn_softmax_kernel, dim3((unsigned)cdiv((int64_t)B * U, 4)), dim3(256), 0, s, bf.scores,
                               attn_mask, kpm, avg, B, H, U, S, (int)S4, vt_pad, E);
    }
    TAL_CHECK_LAUNCH("attn_softmax");
    GemmArgs p = {};
    p.A = bf.scores; p.W = vt; p.Y = bf.ctx; p.bias = in_b + 2 * E;
    p.M = U; p.N = hd; p.K = (int)S4;
    p.lda = S4; p.ldw = S4; p.ldy = E;
    p.nb2 = H;
    p.a_s1 = (int64_t)H * U * S4; p.a_s2 = (int64_t)U * S4;
    p.w_s1 = (int64_t)E * S4; p.w_s2 = (int64_t)hd * S4;
    p.y_s1 = (int64_t)U * E; p.y_s2 = hd;
    p.bias_s2 = hd;
    rc = launch_gemm(p, 0, B * H, s);
    if (rc) return rc;
    return launch_linear(bf.ctx, out_w, out_b, res, alpha, 2, (int64_t)B * U, E, E, out, s);
}

// q (scaled by hd^-0.5, as torch scales q after the in-projection) [and k] in ONE launch: rows 0..n*E of the
// packed in-projection; alpha applies to the first E columns only.  dst [M, n*E].
static int project_q_or_qk(const float* in_w, const float* in_b, const float* x, int64_t M, int E, int H, int n,
                           float* dst, hipStream_t s) {
    GemmArgs g = {};
    g.A = x; g.W = in_w; g.bias = in_b; g.Y = dst;
    g.M = M; g.N = n * E; g.K = E;
    g.lda = E; g.ldw = E; g.ldy = n * E;
    g.nb2 = 1;
    g.alpha = 1.0f / sqrtf((float)(E / H));
    g.scale_cols = n > 1 ? E : 0;
    return launch_gemm(g, 3, 1, s);
}

struct LayerWs {
    MhaBufs mha;
    float* x1;   // [B, U, E]
    float* x2;   // [B, U, E]
    float* ff;   // [B, U, FF]
    float* y3;   // [B, U, 2E]      folded layer: x1 | q_c
    float* y5;   // [B, U, E + FF]  folded layer: x2 | ff
    size_t total_floats;
};

static LayerWs carve(float* base, int B, int U, int S, int E, int H, int FF) {
    const int64_t L = U > S ? U : S;
    const int64_t L4 = pad4(L);
    auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
    LayerWs w;
    size_t o = 0;
    auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += up(n); return p; };
    w.mha.q = take((size_t)B * U * 2 * E);   // q | k of the self-attention in one buffer
    w.mha.k = take((size_t)B * L * E);
    w.mha.vt = take((size_t)B * E * L4);
    w.mha.scores = take((size_t)B * H * U * L4);
    w.mha.ctx = take((size_t)B * U * E);
    w.x1 = take((size_t)B * U * E);
    w.x2 = take((size_t)B * U * E);
    w.ff = take((size_t)B * U * FF);
    w.y3 = take((size_t)B * U * 2 * E);
    w.y5 = take((size_t)B * U * (E + FF));
    w.total_floats = o;
    return w;
}

// ---- the same layer on the latency-oriented kernels (csrc/decode_small.hip): 8 launches instead of 13 ----------------
static SkinnyArgs skinny(const float* A, int64_t lda, const float* W, const float* bias, const float* res, float* Y, int64_t ldy,
                         int M, int N, int K, float alpha) {
    SkinnyArgs g = {};
    g.A = A; g.W = W; g.bias = bias; g.res = res; g.Y = Y;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldw = K; g.ldy = ldy; g.ldres = ldy;
    g.alpha = alpha;
    return g;
}

// Does a problem of `rows` rows run this layer in the folded form (6 launches)?  A function of the problem's OWN row count, so that a
// session takes the same form -- the same bits -- alone and inside a merged step.  The fold trades two launches for a K axis twice
// as long in two dense layers: it wins while the step is latency-bound (measured: -8 % per step at <= 32 prefix tokens, even at
// 64, +6 % at 128, +20 % at 256: profiles/r6_decode_folded_layer.txt).
static bool layer_folded(const tal_decoder_layer_w* w, int E, int rows) {
    return w->fold_sa_w && w->fold_sa_b && w->fold_ca_w && w->fold_ca_b && !opt(OPT_DECODE_NO_FOLD) && E % 256 == 0 &&
           rows <= opt(OPT_DECODE_FOLD_ROWS);
}

static bool small_layer_applicable(int B, int U, int S, int E, int H, int FF, bool have_kv_cache) {
    const bool off = opt(OPT_DECODE_NO_SMALL) != 0;
    const int max_rows = opt(OPT_DECODE_SMALL_ROWS);   // default 256; measured: 0.36 vs 0.41 ms per step at 128 rows, even at 256, slower at 512
    return !off && have_kv_cache && (int64_t)B * U <= max_rows && E % 64 == 0 && FF % 64 == 0 && E % H == 0 &&
           attn_small_applicable(U, S, E / H) && attn_small_applicable(U, U, E / H);
}

// probs_last != NULL: the cross-attention writes the per-head probabilities of the LAST prefix row only, [B][H][1][S]
// (all the greedy loop reads); xattn_avg != NULL: the head-averaged weights of every row, [B][U][S].
// scratch of the kernels that merge partial results in-launch (key-split attention, LM head + pick): the ticket words
// must be zero before the first launch; every kernel leaves them zero
struct DecodeScratch {
    float* part;         // attn_split_scratch_floats(...) floats
    unsigned* tickets;   // TAL_GREEDY_TICKETS words; the last one belongs to the pick
};
// floats of DecodeScratch::part for a prefix of U rows (batch 1): the key-split attention runs while its (row block, head)
// groups fit 64 tickets, the FFN's split-K while its tiles fit the tickets 64 .. 254; the two uses never overlap in time
static size_t decode_scratch_floats(int U, int S, int E, int H) {
    int ua = U;
    while (ua > 16 && attn_split_tickets(1, ua, H) > 64) ua -= 16;
    const size_t a = attn_split_scratch_floats(1, ua, S, H, E / H);
    int rb = (U + 31) / 32;
    const int rb_max = (TAL_GREEDY_TICKETS - 64) / (E / 16 > 0 ? E / 16 : 1);
    if (rb > rb_max) rb = rb_max;
    const size_t b = (size_t)4 * (E / 16) * (rb > 0 ? rb : 1) * 512;
    return a > b ? a : b;
}

static int decoder_layer_small(const tal_decoder_layer_w* w, const float* tgt, int B, int U, int S, int E, int H, int FF,
                               const float* tgt_mask, const uint8_t* mem_kpm, const float* ck, const float* cvt, float* out,
                               float* xattn_avg, float* probs_last, const LayerWs& ws, hipStream_t s,
                               const DecodeScratch* sk = nullptr, int64_t k_pitch = 0, bool allow_fold = true) {
    const int M = B * U, hd = E / H;
    const int64_t U4 = pad4(U), S4 = pad4(S);
    const float qscale = 1.0f / sqrtf((float)hd);
    // self attention: q | k | v^T in one launch (q scaled; v goes out transposed per batch item, bias folded after P.V)
    float* qkv = ws.mha.q;                       // [M][3E]; the v columns of it are never written
    SkinnyArgs g = skinny(tgt, E, w->sa_in_w, w->sa_in_b, nullptr, qkv, 3 * E, M, 3 * E, E, qscale);
    g.scale_cols = E;
    g.Yt = ws.mha.vt; g.vt_begin = 2 * E; g.U = U; g.ldt = U4; g.vt_bs = (int64_t)E * U4;
    // (the v bias is part of the packed bias and so goes into V^T here; P . (V + b) = P . V + b because rows of P sum to 1)
    int rc = launch_skinny_gemm(g, 3, s);
    if (rc) return rc;
    AttnArgs a = {};
    a.q = qkv; a.ldq = 3 * E; a.q_bs = (int64_t)U * 3 * E;
    a.k = qkv + E; a.ldk = 3 * E; a.k_bs = (int64_t)U * 3 * E;
    a.vt = ws.mha.vt; a.ldvt = U4; a.vt_bs = (int64_t)E * U4;
    a.vbias = nullptr;                            // already inside V^T
    a.mask = tgt_mask; a.kpm = nullptr;
    a.ctx = ws.mha.ctx; a.ldc = E; a.c_bs = (int64_t)U * E;
    a.U = U; a.S = U; a.H = H;
    rc = launch_attn_small(a, B, hd, s);
    if (rc) return rc;
    // Folded form (tal_decoder_layer_w.fold_*): the self-attention's out-projection + ReZero and the cross-attention's q projection are
    // ONE dense layer over [ctx | tgt] (x1 | q_c side by side in y3), and so are the cross-attention's out-projection + ReZero and
    // FFN-1 over [ctx2 | x1] (x2 | ff in y5): 6 dependent launches per layer instead of 8.
    const bool fold = allow_fold && layer_folded(w, E, M);
    const float* qc;
    int64_t ldq;
    if (fold) {
        SkinnyArgs g3 = skinny(ws.mha.ctx, E, w->fold_sa_w, w->fold_sa_b, nullptr, ws.y3, 2 * E, M, 2 * E, 2 * E, 0.f);
        g3.A2 = tgt; g3.lda2 = E; g3.K1 = E; g3.k1_cols = E;
        rc = launch_skinny_gemm(g3, 0, s);
        if (rc) return rc;
        qc = ws.y3 + E;
        ldq = 2 * E;
    } else {
        rc = launch_skinny_gemm(skinny(ws.mha.ctx, E, w->sa_out_w, w->sa_out_b, tgt, ws.x1, E, M, E, E, w->resweight), 2, s);
        if (rc) return rc;
        // cross attention over the cached K / V^T of the encoder window
        rc = launch_skinny_gemm(skinny(ws.x1, E, w->ca_in_w, w->ca_in_b, nullptr, ws.mha.q, E, M, E, E, qscale), 3, s);
        if (rc) return rc;
        qc = ws.mha.q;
        ldq = E;
    }
    AttnArgs c = {};
    c.q = qc; c.ldq = ldq; c.q_bs = (int64_t)U * ldq;
    c.k = ck; c.ldk = k_pitch ? k_pitch : E; c.k_bs = (int64_t)S * c.ldk;      // (k_pitch: the window is a view of an episode-wide K | V table)
    c.vt = cvt; c.ldvt = S4; c.vt_bs = (int64_t)E * S4;
    c.vbias = w->ca_in_b + 2 * E;
    c.mask = nullptr; c.kpm = mem_kpm;
    c.ctx = ws.mha.ctx; c.ldc = E; c.c_bs = (int64_t)U * E;
    c.U = U; c.S = S; c.H = H;
    if (xattn_avg) { c.probs = ws.mha.scores; c.prob_row0 = 0; }
    else if (probs_last) { c.probs = probs_last; c.prob_row0 = U - 1; }
    // long key axis + scratch available: cut the keys over workgroups (8x the CUs pulling K / V^T)
    if (sk && S > 64 && attn_split_tickets(B, U, H) <= 64)
        rc = launch_attn_split(c, B, hd, sk->part, sk->tickets, s);
    else
        rc = launch_attn_small(c, B, hd, s);
    if (rc) return rc;
    if (xattn_avg) {
        rc = launch_head_average(ws.mha.scores, xattn_avg, B, H, U, S, s);
        if (rc) return rc;
    }
    SkinnyArgs f2;
    if (fold) {
        SkinnyArgs g5 = skinny(ws.mha.ctx, E, w->fold_ca_w, w->fold_ca_b, nullptr, ws.y5, E + FF, M, E + FF, 2 * E, 0.f);
        g5.A2 = ws.y3; g5.lda2 = 2 * E; g5.K1 = E; g5.k1_cols = E;          // x1 = the first E columns of y3
        g5.relu_begin = E;                                   // x2 plain, ff through the relu
        rc = launch_skinny_gemm(g5, 1, s);
        if (rc) return rc;
        f2 = skinny(ws.y5 + E, E + FF, w->lin2_w, w->lin2_b, ws.y5, out, E, M, E, FF, w->resweight);
        f2.ldw = FF;
        f2.ldres = E + FF;
    } else {
        rc = launch_skinny_gemm(skinny(ws.mha.ctx, E, w->ca_out_w, w->ca_out_b, ws.x1, ws.x2, E, M, E, E, w->resweight_src), 2, s);
        if (rc) return rc;
        // feed-forward
        rc = launch_skinny_gemm(skinny(ws.x2, E, w->lin1_w, w->lin1_b, nullptr, ws.ff, FF, M, FF, E, 0.f), 1, s);
        if (rc) return rc;
        f2 = skinny(ws.ff, FF, w->lin2_w, w->lin2_b, ws.x2, out, E, M, E, FF, w->resweight);
    }
    if (sk && FF >= 2048 && FF % 256 == 0 && (E / 16) * ((M + 31) / 32) <= TAL_GREEDY_TICKETS - 64) {
        // K = FF is deep: four workgroups per tile, each pulling a quarter of the operands (tickets 64 .. 254)
        f2.ksplit = 4;
        f2.sk_part = sk->part;
        f2.sk_tickets = sk->tickets + 64;
    }
    return launch_skinny_gemm(f2, 2, s);
}

// The same layer for the decode steps of G sessions at once (batch 1 each, own prefix length / window / buffers): every launch of
// decoder_layer_small becomes ONE launch over all sessions (the multi forms of csrc/decode_small.hip), so a step of G sessions
// costs the 8 dependent launches of one.  A session's rows go through the same kernel bodies with the same arguments as in
// decoder_layer_small: bit-identical.  All sessions must take the SAME kernel forms there (key-split cross-attention, FFN-2 cut
// along K): greedy_group_ok checks it.
struct SessionLayerIo {
    const float* tgt;        // [U, E]
    int U, S;
    const uint8_t* mem_kpm;
    const float* ck;
    int64_t k_pitch;         // floats between the window's K rows (E, or the pitch of the episode-wide K | V table)
    const float* cvt;
    float* out;
    float* probs_last;       // [H][S]
    LayerWs ws;
    DecodeScratch sk;
    bool no_fold;            // the session's tal_greedy_ctx.no_fold
};
static int decoder_layer_small_multi_form(const tal_decoder_layer_w* w, const SessionLayerIo* io, int G, int E, int H, int FF, bool fold, hipStream_t s);
// (sessions on either side of the fold's row limit go through the layer as two groups of launches: each in the form its solo step takes)
static int decoder_layer_small_multi(const tal_decoder_layer_w* w, const SessionLayerIo* io, int G, int E, int H, int FF, hipStream_t s) {
    SessionLayerIo part[2][TAL_GROUP_MAX];
    int n[2] = {0, 0};
    for (int i = 0; i < G; ++i) {
        const int f = (!io[i].no_fold && layer_folded(w, E, io[i].U)) ? 1 : 0;
        part[f][n[f]++] = io[i];
    }
    for (int f = 1; f >= 0; --f)
        if (n[f] > 0) {
            const int rc = decoder_layer_small_multi_form(w, part[f], n[f], E, H, FF, f == 1, s);
            if (rc) return rc;
        }
    return TAL_OK;
}
static int decoder_layer_small_multi_form(const tal_decoder_layer_w* w, const SessionLayerIo* io, int G, int E, int H, int FF, bool fold, hipStream_t s) {
    const int hd = E / H;
    const float qscale = 1.0f / sqrtf((float)hd);
    SkinnyArgs g[TAL_GROUP_MAX];
    AttnArgs a[TAL_GROUP_MAX];
    float* scr[TAL_GROUP_MAX];
    unsigned* tik[TAL_GROUP_MAX];
    for (int i = 0; i < G; ++i) {
        const SessionLayerIo& x = io[i];
        const int64_t U4 = pad4(x.U);
        g[i] = skinny(x.tgt, E, w->sa_in_w, w->sa_in_b, nullptr, x.ws.mha.q, 3 * E, x.U, 3 * E, E, qscale);
        g[i].scale_cols = E;
        g[i].Yt = x.ws.mha.vt; g[i].vt_begin = 2 * E; g[i].U = x.U; g[i].ldt = U4; g[i].vt_bs = (int64_t)E * U4;
    }
    int rc = launch_skinny_gemm_multi(g, G, 3, s);
    if (rc) return rc;
    for (int i = 0; i < G; ++i) {
        const SessionLayerIo& x = io[i];
        const int64_t U4 = pad4(x.U);
        float* qkv = x.ws.mha.q;
        a[i] = AttnArgs{};
        a[i].q = qkv; a[i].ldq = 3 * E; a[i].q_bs = (int64_t)x.U * 3 * E;
        a[i].k = qkv + E; a[i].ldk = 3 * E; a[i].k_bs = (int64_t)x.U * 3 * E;
        a[i].vt = x.ws.mha.vt; a[i].ldvt = U4; a[i].vt_bs = (int64_t)E * U4;
        a[i].ctx = x.ws.mha.ctx; a[i].ldc = E; a[i].c_bs = (int64_t)x.U * E;
        a[i].U = x.U; a[i].S = x.U; a[i].H = H;
    }
    rc = launch_attn_small_multi(a, G, hd, s);
    if (rc) return rc;
    // (fold: the same arguments per session as decoder_layer_small builds: bit-identical)
    if (fold) {
        for (int i = 0; i < G; ++i) {
            g[i] = skinny(io[i].ws.mha.ctx, E, w->fold_sa_w, w->fold_sa_b, nullptr, io[i].ws.y3, 2 * E, io[i].U, 2 * E, 2 * E, 0.f);
            g[i].A2 = io[i].tgt; g[i].lda2 = E; g[i].K1 = E; g[i].k1_cols = E;
        }
        rc = launch_skinny_gemm_multi(g, G, 0, s);
        if (rc) return rc;
    } else {
        for (int i = 0; i < G; ++i) g[i] = skinny(io[i].ws.mha.ctx, E, w->sa_out_w, w->sa_out_b, io[i].tgt, io[i].ws.x1, E, io[i].U, E, E, w->resweight);
        rc = launch_skinny_gemm_multi(g, G, 2, s);
        if (rc) return rc;
        for (int i = 0; i < G; ++i) g[i] = skinny(io[i].ws.x1, E, w->ca_in_w, w->ca_in_b, nullptr, io[i].ws.mha.q, E, io[i].U, E, E, qscale);
        rc = launch_skinny_gemm_multi(g, G, 3, s);
        if (rc) return rc;
    }
    const int64_t ldq = fold ? 2 * E : E;
    for (int i = 0; i < G; ++i) {
        const SessionLayerIo& x = io[i];
        const int64_t S4 = pad4(x.S);
        a[i] = AttnArgs{};
        a[i].q = fold ? x.ws.y3 + E : x.ws.mha.q; a[i].ldq = ldq; a[i].q_bs = (int64_t)x.U * ldq;
        a[i].k = x.ck; a[i].ldk = x.k_pitch; a[i].k_bs = (int64_t)x.S * x.k_pitch;
        a[i].vt = x.cvt; a[i].ldvt = S4; a[i].vt_bs = (int64_t)E * S4;
        a[i].vbias = w->ca_in_b + 2 * E;
        a[i].kpm = x.mem_kpm;
        a[i].ctx = x.ws.mha.ctx; a[i].ldc = E; a[i].c_bs = (int64_t)x.U * E;
        a[i].U = x.U; a[i].S = x.S; a[i].H = H;
        a[i].probs = x.probs_last; a[i].prob_row0 = x.U - 1;
        scr[i] = x.sk.part;
        tik[i] = x.sk.tickets;
    }
    rc = launch_attn_split_multi(a, scr, tik, G, hd, s);
    if (rc) return rc;
    if (fold) {
        for (int i = 0; i < G; ++i) {
            g[i] = skinny(io[i].ws.mha.ctx, E, w->fold_ca_w, w->fold_ca_b, nullptr, io[i].ws.y5, E + FF, io[i].U, E + FF, 2 * E, 0.f);
            g[i].A2 = io[i].ws.y3; g[i].lda2 = 2 * E; g[i].K1 = E; g[i].k1_cols = E;
            g[i].relu_begin = E;
        }
        rc = launch_skinny_gemm_multi(g, G, 1, s);
        if (rc) return rc;
    } else {
        for (int i = 0; i < G; ++i) g[i] = skinny(io[i].ws.mha.ctx, E, w->ca_out_w, w->ca_out_b, io[i].ws.x1, io[i].ws.x2, E, io[i].U, E, E, w->resweight_src);
        rc = launch_skinny_gemm_multi(g, G, 2, s);
        if (rc) return rc;
        for (int i = 0; i < G; ++i) g[i] = skinny(io[i].ws.x2, E, w->lin1_w, w->lin1_b, nullptr, io[i].ws.ff, FF, io[i].U, FF, E, 0.f);
        rc = launch_skinny_gemm_multi(g, G, 1, s);
        if (rc) return rc;
    }
    for (int i = 0; i < G; ++i) {
        if (fold) {
            g[i] = skinny(io[i].ws.y5 + E, E + FF, w->lin2_w, w->lin2_b, io[i].ws.y5, io[i].out, E, io[i].U, E, FF, w->resweight);
            g[i].ldres = E + FF;
        } else {
            g[i] = skinny(io[i].ws.ff, FF, w->lin2_w, w->lin2_b, io[i].ws.x2, io[i].out, E, io[i].U, E, FF, w->resweight);
        }
        g[i].ksplit = 4;
        g[i].sk_part = io[i].sk.part;
        g[i].sk_tickets = io[i].sk.tickets + 64;
    }
    return launch_skinny_gemm_multi(g, G, 2, s);
}

// may a step of (U, S) run as ONE launch (csrc/decode_persist.hip)?  The shapes the merged launches take (so that every phase is the
// kernel form the launch chain uses for this step), dense layers no deeper than 512 per K slice (four waves per workgroup), and the
// FFN's split-K tickets ending below the words the one-launch step keeps its counters in.
static bool greedy_persist_ok(const tal_greedy_ctx* c, int U, int S) {
    const int E = c->E, H = c->H, FF = c->FF, K0 = c->E0 > 0 ? c->E0 : c->E, hd = E / H;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // (the one-launch form walks the UNFOLDED layer's phases: it is a measurement switch of round 5, valid beside option decode_no_fold)
    return c->tickets && !c->pick_bias && (c->no_fold || !layer_folded(&c->layers[0], c->E, U)) && c->n_layers <= TAL_PS_MAX_LAYERS && (hd == 128 || hd == 64) && E <= 512 && FF / 4 <= 512 && E % 64 == 0 &&
           small_layer_applicable(1, U, S, E, H, FF, true) && S > 64 && attn_split_tickets(1, U, H) <= 64 && FF >= 2048 && FF % 256 == 0 &&
           64 + (E / 16) * ((U + 31) / 32) <= PS_BAR && E % 16 == 0 && K0 % 8 == 0 && al16(c->emb) && (!c->proj_t || al16(c->proj_t));
}
static void ps_fill_model(const tal_greedy_ctx* c, PsModel& m) {
    m = PsModel{};
    for (int l = 0; l < c->n_layers; ++l) m.layer[l] = c->layers[l];
    m.n_layers = c->n_layers; m.E = c->E; m.H = c->H; m.FF = c->FF; m.V = c->V;
    m.K0 = c->E0 > 0 ? c->E0 : c->E;
    m.qscale = 1.0f / sqrtf((float)(c->E / c->H));
    m.emb = c->emb; m.proj = c->E0 > 0 ? c->proj : nullptr; m.proj_t = c->E0 > 0 ? c->proj_t : nullptr; m.pe = c->pe;
}

// does a session's step take, in decoder_layer_small, exactly the kernel forms decoder_layer_small_multi launches?
static bool greedy_group_ok(int U, int S, int E, int H, int FF) {
    return small_layer_applicable(1, U, S, E, H, FF, true) && S > 64 && attn_split_tickets(1, U, H) <= 64 &&
           FF >= 2048 && FF % 256 == 0 && (E / 16) * ((U + 31) / 32) <= TAL_GREEDY_TICKETS - 64;
}

}  // namespace tal

using namespace tal;

extern "C" int64_t tal_pad4(int64_t n) { return pad4(n); }

extern "C" int tal_embed_tokens_fwd(const int64_t* tokens, int B, int U, const float* emb, int V, int E0,
                                    const float* proj, int D, const float* pe, int max_len, float* out,
                                    void* stream) {
    TAL_CHECK_ARG(tokens && emb && pe && out, "tal_embed_tokens_fwd: null pointer");
    TAL_CHECK_ARG(B > 0 && U > 0 && V > 0 && E0 > 0 && D > 0, "tal_embed_tokens_fwd: bad shape");
    TAL_CHECK_ARG(U <= max_len, "tal_embed_tokens_fwd: sequence length %d exceeds max_positions %d", U, max_len);
    TAL_CHECK_ARG(proj || D == E0, "tal_embed_tokens_fwd: no projection needs D == E0");
    TAL_CHECK_ARG(E0 <= 8192, "tal_embed_tokens_fwd: embedding width %d too large", E0);
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)(B * U)), dim3(256), (size_t)E0 * sizeof(float),
                       (hipStream_t)stream, tokens, emb, proj, pe, out, U, V, E0, D);
    TAL_CHECK_LAUNCH("tal_embed_tokens_fwd");
    return TAL_OK;
}

extern "C" int tal_add_positional_fwd(const float* x, int B, int U, int D, const float* pe, int max_len, float* out,
                                      void* stream) {
    TAL_CHECK_ARG(x && pe && out && B > 0 && U > 0 && D > 0, "tal_add_positional_fwd: bad argument");
    TAL_CHECK_ARG(U <= max_len, "tal_add_positional_fwd: sequence length %d exceeds max_len %d", U, max_len);
    const int64_t total = (int64_t)B * U * D;
    hipLaunchKernelGGL(add_positional_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       pe, out, U, D, total);
    TAL_CHECK_LAUNCH("tal_add_positional_fwd");
    return TAL_OK;
}

extern "C" int tal_cross_kv_fwd(const tal_decoder_layer_w* w, const float* mem, int B, int S, int E, float* k,
                                float* vt, void* stream) {
    TAL_CHECK_ARG(w && mem && k && vt && w->ca_in_w && w->ca_in_b, "tal_cross_kv_fwd: null pointer");
    TAL_CHECK_ARG(B > 0 && S > 0 && E > 0 && E % 4 == 0, "tal_cross_kv_fwd: bad shape");
    return project_kv(w->ca_in_w, w->ca_in_b, mem, B, S, E, k, vt, true, (hipStream_t)stream);
}

extern "C" size_t tal_decoder_layer_workspace_bytes(int B, int U, int S, int E, int H, int FF) {
    if (B <= 0 || U <= 0 || S <= 0 || E <= 0 || H <= 0 || FF <= 0) return 0;
    return carve(nullptr, B, U, S, E, H, FF).total_floats * sizeof(float);
}

static int decoder_layer_pitched(const tal_decoder_layer_w* w, const float* tgt, int B, int U, const float* mem,
                                 int S, int E, int H, int FF, const float* tgt_mask, const uint8_t* mem_kpm,
                                 const float* k_cache, const float* vt_cache, float* out, float* xattn_avg,
                                 void* workspace, size_t workspace_bytes, void* stream, int64_t k_pitch);

extern "C" int tal_decoder_layer_fwd(const tal_decoder_layer_w* w, const float* tgt, int B, int U, const float* mem,
                                     int S, int E, int H, int FF, const float* tgt_mask, const uint8_t* mem_kpm,
                                     const float* k_cache, const float* vt_cache, float* out, float* xattn_avg,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return decoder_layer_pitched(w, tgt, B, U, mem, S, E, H, FF, tgt_mask, mem_kpm, k_cache, vt_cache, out, xattn_avg, workspace,
                                 workspace_bytes, stream, 0);
}

// k_pitch != 0: k_cache is a window of an episode-wide K | V table (rows k_pitch floats apart; batch 1)
static int decoder_layer_pitched(const tal_decoder_layer_w* w, const float* tgt, int B, int U, const float* mem,
                                 int S, int E, int H, int FF, const float* tgt_mask, const uint8_t* mem_kpm,
                                 const float* k_cache, const float* vt_cache, float* out, float* xattn_avg,
                                 void* workspace, size_t workspace_bytes, void* stream, int64_t k_pitch) {
    TAL_CHECK_ARG(w && tgt && out && workspace, "tal_decoder_layer_fwd: null pointer");
    TAL_CHECK_ARG(w->sa_in_w && w->sa_in_b && w->sa_out_w && w->sa_out_b && w->ca_in_w && w->ca_in_b && w->ca_out_w &&
                      w->ca_out_b && w->lin1_w && w->lin1_b && w->lin2_w && w->lin2_b,
                  "tal_decoder_layer_fwd: null weight");
    TAL_CHECK_ARG(B > 0 && U > 0 && S > 0 && E > 0 && H > 0 && FF > 0 && E % H == 0 && (E / H) % 4 == 0 && FF % 4 == 0,
                  "tal_decoder_layer_fwd: bad shape B=%d U=%d S=%d E=%d H=%d FF=%d", B, U, S, E, H, FF);
    TAL_CHECK_ARG(mem || (k_cache && vt_cache), "tal_decoder_layer_fwd: need the memory or its cached K / V^T");
    TAL_CHECK_ARG(!k_pitch || (B == 1 && k_cache && k_pitch >= E && k_pitch % 4 == 0), "tal_decoder_layer_fwd: a pitched K window needs batch 1");
    if (workspace_bytes < tal_decoder_layer_workspace_bytes(B, U, S, E, H, FF)) {
        set_error("tal_decoder_layer_fwd: workspace %zu < %zu bytes", workspace_bytes,
                  tal_decoder_layer_workspace_bytes(B, U, S, E, H, FF));
        return TAL_ENOMEM;
    }
    hipStream_t s = (hipStream_t)stream;
    LayerWs ws = carve(reinterpret_cast<float*>(workspace), B, U, S, E, H, FF);
    if (small_layer_applicable(B, U, S, E, H, FF, k_cache && vt_cache))
        return decoder_layer_small(w, tgt, B, U, S, E, H, FF, tgt_mask, mem_kpm, k_cache, vt_cache, out, xattn_avg, nullptr, ws, s, nullptr, k_pitch);
    // self attention over the prefix: q|k in one launch, V^T, scores/softmax/PV, out-proj + ReZero
    int rc = project_q_or_qk(w->sa_in_w, w->sa_in_b, tgt, (int64_t)B * U, E, H, 2, ws.mha.q, s);
    if (rc) return rc;
    rc = project_kv(w->sa_in_w, w->sa_in_b, tgt, B, U, E, nullptr, ws.mha.vt, false, s);
    if (rc) return rc;
    rc = mha_core(w->sa_in_b, w->sa_out_w, w->sa_out_b, ws.mha.q, 2 * E, ws.mha.q + E, 2 * E, ws.mha.vt, ws.mha.vt, B,
                  U, U, E, H, tgt_mask, nullptr, tgt, w->resweight, ws.x1, nullptr, ws.mha, s);
    if (rc) return rc;
    // cross attention over the encoder window
    const float* ck = k_cache;
    const float* cvt = vt_cache;
    float* pad = nullptr;
    if (!ck || !cvt) {
        rc = project_kv(w->ca_in_w, w->ca_in_b, mem, B, S, E, ws.mha.k, ws.mha.vt, false, s);
        if (rc) return rc;
        ck = ws.mha.k;
        cvt = ws.mha.vt;
        pad = ws.mha.vt;
    }
    rc = project_q_or_qk(w->ca_in_w, w->ca_in_b, ws.x1, (int64_t)B * U, E, H, 1, ws.mha.q, s);
    if (rc) return rc;
    rc = mha_core(w->ca_in_b, w->ca_out_w, w->ca_out_b, ws.mha.q, E, ck, (k_pitch && ck == k_cache) ? k_pitch : E, cvt, pad, B, U, S, E, H, nullptr, mem_kpm,
                  ws.x1, w->resweight_src, ws.x2, xattn_avg, ws.mha, s);
    if (rc) return rc;
    // feed-forward
    rc = launch_linear(ws.x2, w->lin1_w, w->lin1_b, nullptr, 0.f, 1, (int64_t)B * U, FF, E, ws.ff, s);
    if (rc) return rc;
    return launch_linear(ws.ff, w->lin2_w, w->lin2_b, ws.x2, w->resweight, 2, (int64_t)B * U, E, FF, out, s);
}

// Whole decoder stack in one call (nn.TransformerDecoder.forward of torch 1.4 = a loop over the
// layers, norm=None): saves the per-layer host round trips of the decode loops, which launch a
// few hundred microseconds of GPU work per generated token.
extern "C" int tal_decoder_stack_fwd(const tal_decoder_layer_w* layers, int n_layers, const float* tgt, int B, int U,
                                     const float* mem, int S, int E, int H, int FF, const float* tgt_mask,
                                     const uint8_t* mem_kpm, const float* const* k_cache,
                                     const float* const* vt_cache, float* out, float* xattn_avg, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    TAL_CHECK_ARG(layers && n_layers >= 1 && tgt && out, "tal_decoder_stack_fwd: bad argument");
    const float* cur = tgt;
    for (int l = 0; l < n_layers; ++l) {
        float* avg = xattn_avg ? xattn_avg + (size_t)l * B * U * S : nullptr;
        const int rc = tal_decoder_layer_fwd(&layers[l], cur, B, U, mem, S, E, H, FF, tgt_mask, mem_kpm,
                                             k_cache ? k_cache[l] : nullptr, vt_cache ? vt_cache[l] : nullptr, out,
                                             avg, workspace, workspace_bytes, stream);
        if (rc) return rc;
        cur = out;  // layers l >= 1 run in place (tal_decoder_layer_fwd allows out == tgt)
    }
    return TAL_OK;
}

extern "C" int tal_lm_head_fwd(const float* h, int64_t M, int64_t ldh, int D, const float* proj_t, int E0,
                               const float* emb, int V, float* logits, void* workspace, size_t workspace_bytes,
                               void* stream) {
    TAL_CHECK_ARG(h && emb && logits, "tal_lm_head_fwd: null pointer");
    TAL_CHECK_ARG(M >= 0 && D > 0 && E0 > 0 && V > 0 && ldh >= D && ldh % 4 == 0, "tal_lm_head_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g = {};
    g.nb2 = 1;
    if (!proj_t) {
        TAL_CHECK_ARG(D == E0, "tal_lm_head_fwd: no projection needs D == E0");
        g.A = h; g.W = emb; g.Y = logits; g.M = M; g.N = V; g.K = D; g.lda = ldh; g.ldw = D; g.ldy = V;
        return launch_gemm(g, 0, 1, s);
    }
    TAL_CHECK_ARG(workspace, "tal_lm_head_fwd: needs a workspace of M*E0 floats");
    if (workspace_bytes < (size_t)M * E0 * sizeof(float)) {
        set_error("tal_lm_head_fwd: workspace %zu < %zu bytes", workspace_bytes, (size_t)M * E0 * sizeof(float));
        return TAL_ENOMEM;
    }
    float* t = reinterpret_cast<float*>(workspace);
    g.A = h; g.W = proj_t; g.Y = t; g.M = M; g.N = E0; g.K = D; g.lda = ldh; g.ldw = D; g.ldy = E0;
    int rc = launch_gemm(g, 0, 1, s);
    if (rc) return rc;
    return launch_linear(t, emb, nullptr, nullptr, 0.f, 0, M, V, E0, logits, s);
}

extern "C" int tal_transpose_fwd(const float* x, int R, int Cc, float* y, void* stream) {
    TAL_CHECK_ARG(x && y && R > 0 && Cc > 0, "tal_transpose_fwd: bad argument");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)cdiv(Cc, 32), (unsigned)cdiv(R, 32)), dim3(32, 8), 0,
                       (hipStream_t)stream, x, R, Cc, y);
    TAL_CHECK_LAUNCH("tal_transpose_fwd");
    return TAL_OK;
}

extern "C" int tal_greedy_pick_fwd(const float* logits, int V, const float* attn_rows, int n_layers, int64_t layer_stride,
                                   int S, float* out, int64_t* token_out, void* stream) {
    TAL_CHECK_ARG(logits && attn_rows && out && V > 0 && S > 0 && n_layers > 0, "tal_greedy_pick_fwd: bad argument");
    hipLaunchKernelGGL(greedy_pick_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, V, attn_rows, n_layers,
                       layer_stride, 1, (int64_t)0, S, out, token_out);
    TAL_CHECK_LAUNCH("tal_greedy_pick_fwd");
    return TAL_OK;
}

extern "C" int tal_log_softmax_rows(const float* x, int64_t M, int N, float* out, void* stream) {
    TAL_CHECK_ARG(x && out && M >= 0 && N > 0, "tal_log_softmax_rows: bad argument");
    if (M == 0) return TAL_OK;
    if (M <= 512)
        hipLaunchKernelGGL(log_softmax_row_block_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, x, N, out);
    else
        hipLaunchKernelGGL(log_softmax_rows_kernel, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, M,
                           N, out);
    TAL_CHECK_LAUNCH("tal_log_softmax_rows");
    return TAL_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// One step of the sliding-window greedy decode (System.generate_unaligned, tal/asr/system.py:332-411) as ONE call:
// embed the live prefix -> decoder stack against the cached K / V^T of the encoder window -> tied LM head on the last
// position -> token = argmax(log_softmax) + the new token's layer- and head-averaged cross-attention row -> append the
// token to the device-resident prefix -> (sync != 0) copy {token, row} to pinned host memory and wait.
// ---------------------------------------------------------------------------------------------------------------
static size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }

extern "C" size_t tal_greedy_step_workspace_bytes(int U_max, int S, int E, int H, int FF, int V, int E0, int n_layers) {
    if (U_max <= 0 || S <= 0 || E <= 0 || H <= 0 || FF <= 0 || V <= 0 || n_layers <= 0) return 0;
    size_t f = carve(nullptr, 1, U_max, S, E, H, FF).total_floats;
    f += 2 * up64((size_t)U_max * E);                        // embedded prefix, stack output
    f += up64((size_t)n_layers * (size_t)U_max * S);         // attention rows: per-head last rows [L][H][S] or averaged [L][U][S]
    f += up64((size_t)n_layers * H * S);
    f += up64((size_t)(E0 > 0 ? E0 : E)) + up64((size_t)V);   // LM head intermediate, logits
    f += up64(decode_scratch_floats(U_max, S, E, H));         // key-split attention records / split-K partial tiles
    f += up64((size_t)2 * cdiv(V, LMP_ROWS));                 // pick partials
    return f * sizeof(float);
}

// 1: the result of the context's latest host-direct step (sync 2 / 3) is in picked_host; 0: not yet, after waiting up to
// wait_ms milliseconds (0: one look); < 0: bad argument.  Host-only: reads the sequence word the pick kernel writes last.
// A context whose last step failed part-way (needs_reset) starts its next step from a zeroed ticket block, behind everything the
// failed step may still have in flight.
static int greedy_reset_if_needed(tal_greedy_ctx* c, hipStream_t s) {
    if (!c->needs_reset) return TAL_OK;
    if (hipStreamSynchronize(s) != hipSuccess ||
        (c->tickets && hipMemsetAsync(c->tickets, 0, (size_t)TAL_GREEDY_TICKETS * sizeof(unsigned), s) != hipSuccess)) {
        set_error("greedy step: could not reset the context's ticket block after a failed step: %s", hipGetErrorString(hipGetLastError()));
        return TAL_EHIP;
    }
    c->needs_reset = 0;
    return TAL_OK;
}
// the failure marker of the one-launch step (csrc/decode_persist.hip, ps_raise): token -1 in the result buffer
static int greedy_failed_marker(tal_greedy_ctx* c, const char* who) {
    if (reinterpret_cast<const volatile int*>(c->picked_host)[0] != -1) return TAL_OK;
    c->needs_reset = 1;
    set_error("%s: the one-launch decode step gave up at a phase barrier (its workgroups were not resident together, or one of them "
              "died); no token was produced, the context's ticket block is reset before its next step", who);
    return TAL_EHIP;
}

extern "C" int tal_greedy_step_poll(tal_greedy_ctx* c, int wait_ms) {
    TAL_CHECK_ARG(c && c->picked_host && c->S > 0 && wait_ms >= 0, "tal_greedy_step_poll: bad argument");
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(c->picked_host + 1 + c->S);
    const unsigned seq = c->seq;
    if (seq == 0) return 0;          // no host-direct step has been issued from this context yet (0 is never a sequence value)
    if (*flag != seq && wait_ms > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; *flag != seq; ++spins)
            if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(wait_ms)) break;
    }
    if (*flag != seq) return 0;
    std::atomic_thread_fence(std::memory_order_acquire);
    if (greedy_failed_marker(c, "tal_greedy_step_poll")) return TAL_EHIP;
    return 1;
}

extern "C" int tal_greedy_step_fwd(tal_greedy_ctx* c, int64_t history_start, int64_t n_gen, int sync, void* stream) {
    TAL_CHECK_ARG(c && c->layers && c->emb && c->pe && c->k_cache && c->vt_cache && c->tokens && c->workspace && c->picked_dev,
                  "tal_greedy_step_fwd: null pointer");
    const int E = c->E, H = c->H, FF = c->FF, S = c->S, V = c->V, E0 = c->E0, L = c->n_layers;
    const int64_t U64 = n_gen - history_start;
    TAL_CHECK_ARG(history_start >= 0 && U64 >= 1 && U64 <= c->max_len, "tal_greedy_step_fwd: prefix [%lld, %lld) must hold 1..%d tokens",
                  (long long)history_start, (long long)n_gen, c->max_len);
    TAL_CHECK_ARG(sync >= 0 && sync <= 3, "tal_greedy_step_fwd: sync=%d", sync);
    TAL_CHECK_ARG(!sync || c->picked_host, "tal_greedy_step_fwd: sync needs the pinned host buffer");
    TAL_CHECK_ARG(sync != 3 || (c->tickets && c->E % 16 == 0 && (c->E0 > 0 ? c->E0 : c->E) % 8 == 0 && (reinterpret_cast<uintptr_t>(c->emb) & 15) == 0 &&
                                (!c->proj_t || (reinterpret_cast<uintptr_t>(c->proj_t) & 15) == 0)),
                  "tal_greedy_step_fwd: sync 3 (result written to pinned memory, polled by the caller) needs the merged LM-head + pick kernel "
                  "(tickets, E %% 16 == 0, embedding width %% 8 == 0, 16-byte aligned emb / proj_t): nothing else writes the sequence word");
    const int U = (int)U64;
    if (c->workspace_bytes < tal_greedy_step_workspace_bytes(U, S, E, H, FF, V, E0, L)) {
        set_error("tal_greedy_step_fwd: workspace %zu < %zu bytes", c->workspace_bytes, tal_greedy_step_workspace_bytes(U, S, E, H, FF, V, E0, L));
        return TAL_ENOMEM;
    }
    hipStream_t s = (hipStream_t)stream;
    if (int rr = greedy_reset_if_needed(c, s)) return rr;
    float* base = reinterpret_cast<float*>(c->workspace);
    LayerWs ws = carve(base, 1, U, S, E, H, FF);
    float* p = base + ws.total_floats;
    float* h0 = p; p += up64((size_t)U * E);
    float* h1 = p; p += up64((size_t)U * E);
    float* avg = p; p += up64((size_t)L * U * S);
    float* probs = p; p += up64((size_t)L * H * S);
    float* lm_t = p; p += up64((size_t)(E0 > 0 ? E0 : E));
    float* logits = p; p += up64((size_t)V);
    DecodeScratch sk = {p, c->tickets};
    p += up64(decode_scratch_floats(U, S, E, H));
    float* pick_part = p;
    if (opt(OPT_DECODE_PERSIST) != 0 && greedy_persist_ok(c, U, S)) {
        // ---- the whole step as ONE launch (csrc/decode_persist.hip): the same kernel bodies on the same arguments behind phase barriers
        for (int l = 0; l < L; ++l) TAL_CHECK_ARG(c->k_cache[l] && c->vt_cache[l], "tal_greedy_step_fwd: layer %d has no cached K / V^T", l);
        const bool host_direct = sync >= 2;
        if (host_direct && !c->picked_host_dev) {
            void* alias = nullptr;
            if (hipHostGetDevicePointer(&alias, c->picked_host, 0) != hipSuccess || !alias) {
                set_error("tal_greedy_step_fwd: picked_host is not mapped pinned host memory (%s)", hipGetErrorString(hipGetLastError()));
                return TAL_EINVAL;
            }
            c->picked_host_dev = reinterpret_cast<float*>(alias);
            reinterpret_cast<volatile unsigned*>(c->picked_host)[1 + S] = 0u;
        }
        if (host_direct && ++c->seq == 0) ++c->seq;
        PsArgs a;
        ps_fill_model(c, a.m);
        a.n = 1;
        a.G = opt(OPT_DECODE_PERSIST_WGS) > 0 ? opt(OPT_DECODE_PERSIST_WGS) : 32;
        PsSession& q = a.s[0];
        q = PsSession{};
        q.tokens = c->tokens + history_start; q.token_out = c->tokens + n_gen;
        q.U = U; q.S = S;
        q.h0 = h0; q.h1 = h1; q.qkv = ws.mha.q; q.vt = ws.mha.vt; q.ctx = ws.mha.ctx; q.x1 = ws.x1; q.x2 = ws.x2; q.ff = ws.ff;
        q.probs = probs; q.sk_part = sk.part; q.pick_part = pick_part;
        q.out = host_direct ? c->picked_host_dev : c->picked_dev;
        q.tickets = c->tickets;
        for (int l = 0; l < L; ++l) { q.k_cache[l] = c->k_cache[l]; q.vt_cache[l] = c->vt_cache[l]; }
        q.kpm = c->mem_kpm; q.k_pitch = c->k_pitch;
        q.host_seq = host_direct ? c->seq : 0u;
        int rc = launch_greedy_persist(a, s);
        if (rc) return rc;
        // sync 0: the caller reads picked_dev itself -- a token of -1 there is the failure marker (include/tal_asrd.h); sync 3: the
        // caller's tal_greedy_step_poll finds it
        if (sync == 3 || sync == 0) return TAL_OK;
        if (host_direct) {
            const int got = tal_greedy_step_poll(c, 20000);
            if (got == 1) return TAL_OK;
            if (got == 0) {
                const hipError_t e = hipStreamSynchronize(s);
                c->needs_reset = 1;          // (whatever it was: the counters of the phases that did run are still in the block)
                set_error("tal_greedy_step_fwd: no result of the one-launch step after 20 s (stream after the wait: %s)", hipGetErrorString(e));
            }
            return TAL_EHIP;
        }
        if (hipMemcpyAsync(c->picked_host, c->picked_dev, (size_t)(1 + S) * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            set_error("tal_greedy_step_fwd: device-to-host copy failed: %s", hipGetErrorString(hipGetLastError()));
            return TAL_EHIP;
        }
        return greedy_failed_marker(c, "tal_greedy_step_fwd");
    }
    int rc = tal_embed_tokens_fwd(c->tokens + history_start, 1, U, c->emb, V, E0 > 0 ? E0 : E, E0 > 0 ? c->proj : nullptr, E, c->pe,
                                  c->max_len, h0, stream);
    if (rc) return rc;
    const bool small = small_layer_applicable(1, U, S, E, H, FF, true);
    const float* cur = h0;
    for (int l = 0; l < L; ++l) {
        TAL_CHECK_ARG(c->k_cache[l] && c->vt_cache[l], "tal_greedy_step_fwd: layer %d has no cached K / V^T", l);
        if (small)
            rc = decoder_layer_small(&c->layers[l], cur, 1, U, S, E, H, FF, nullptr, c->mem_kpm, c->k_cache[l], c->vt_cache[l], h1,
                                     nullptr, probs + (size_t)l * H * S, ws, s, c->tickets ? &sk : nullptr, c->k_pitch, !c->no_fold);
        else
            rc = decoder_layer_pitched(&c->layers[l], cur, 1, U, nullptr, S, E, H, FF, nullptr, c->mem_kpm, c->k_cache[l],
                                       c->vt_cache[l], h1, avg + (size_t)l * U * S, base, ws.total_floats * sizeof(float), stream, c->k_pitch);
        if (rc) return rc;
        cur = h1;
    }
    // tied factorised LM head on the last position (models.py:243-246; system.py:355-361 reads only that row)
    const float* hl = cur + (size_t)(U - 1) * E;
    const int K0 = E0 > 0 ? E0 : E;
    // sync == 2 / 3: the pick kernel writes {token, row, sequence word} straight into the pinned host buffer (1 + S + 1 words)
    // through its device alias; sync 2 polls the word here, sync 3 leaves the polling to the caller (tal_greedy_step_poll:
    // several sessions in flight on several streams).  The sequence value is per context (one context per stream).
    const bool host_direct = sync >= 2 && c->tickets;
    if (host_direct && !c->picked_host_dev) {
        void* alias = nullptr;
        if (hipHostGetDevicePointer(&alias, c->picked_host, 0) != hipSuccess || !alias) {
            set_error("tal_greedy_step_fwd: picked_host is not mapped pinned host memory (%s)", hipGetErrorString(hipGetLastError()));
            return TAL_EINVAL;
        }
        c->picked_host_dev = reinterpret_cast<float*>(alias);
        reinterpret_cast<volatile unsigned*>(c->picked_host)[1 + S] = 0u;      // (a recycled buffer may hold an old sequence value)
    }
    if (host_direct && ++c->seq == 0) ++c->seq;         // (0 is the value of a buffer nobody has written yet)
    const unsigned seq = host_direct ? c->seq : 0u;
    if (c->tickets && E % 16 == 0 && K0 % 8 == 0 && (reinterpret_cast<uintptr_t>(c->emb) & 15) == 0 &&
        (!c->proj_t || (reinterpret_cast<uintptr_t>(c->proj_t) & 15) == 0)) {
        const float* rows = small ? probs : avg + (size_t)(U - 1) * S;
        LmPickArgs q = {};
        q.h = hl; q.attn = rows;
        q.layer_stride = small ? (int64_t)H * S : (int64_t)U * S;
        q.head_stride = small ? (int64_t)S : (int64_t)0;
        q.S = S; q.partial = pick_part; q.ticket_word = c->tickets + (TAL_GREEDY_TICKETS - 1);
        q.out = host_direct ? c->picked_host_dev : c->picked_dev;
        q.token_out = c->tokens + n_gen;
        q.host_seq = host_direct ? seq : 0u;
        q.bias = c->pick_bias;
        hipLaunchKernelGGL(lm_pick_kernel, dim3((unsigned)cdiv(V, LMP_ROWS)), dim3(256), (size_t)(E + K0 + LMP_ROWS) * sizeof(float), s, q,
                           E0 > 0 ? c->proj_t : nullptr, E, K0, c->emb, V, L, small ? H : 1);
        TAL_CHECK_LAUNCH("tal_greedy_step_fwd(lm head + pick)");
        if (sync == 3) return TAL_OK;
        if (host_direct) {
            const int got = tal_greedy_step_poll(c, 20000);
            if (got == 1) return TAL_OK;
            if (got == 0) {
                // nothing may stay in flight when the caller is told the step failed: it is free to release the pinned buffer
                const hipError_t e = hipStreamSynchronize(s);
                set_error("tal_greedy_step_fwd: no result after 20 s (stream after the wait: %s)", hipGetErrorString(e));
            }
            return TAL_EHIP;
        }
    } else {
    if (E0 > 0) {
        rc = tal_lm_head_fwd(hl, 1, E, E, c->proj_t, E0, c->emb, V, logits, lm_t, (size_t)E0 * sizeof(float), stream);
    } else {
        rc = tal_lm_head_fwd(hl, 1, E, E, nullptr, E, c->emb, V, logits, nullptr, 0, stream);
    }
    if (rc) return rc;
    if (small)
        hipLaunchKernelGGL(greedy_pick_kernel, dim3(1), dim3(256), 0, s, logits, V, probs, L, (int64_t)H * S, H, (int64_t)S, S,
                           c->picked_dev, c->tokens + n_gen, c->pick_bias);
    else
        hipLaunchKernelGGL(greedy_pick_kernel, dim3(1), dim3(256), 0, s, logits, V, avg + (size_t)(U - 1) * S, L, (int64_t)U * S, 1,
                           (int64_t)0, S, c->picked_dev, c->tokens + n_gen, c->pick_bias);
    TAL_CHECK_LAUNCH("tal_greedy_step_fwd(pick)");
    }
    if (sync) {       // (sync 2 without the merged kernels: the copy form)
        if (hipMemcpyAsync(c->picked_host, c->picked_dev, (size_t)(1 + S) * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            set_error("tal_greedy_step_fwd: device-to-host copy failed: %s", hipGetErrorString(hipGetLastError()));
            return TAL_EHIP;
        }
    }
    return TAL_OK;
}

// ---- several sessions per launch -----------------------------------------------------------------------------------------
extern "C" int tal_greedy_group_ok(const tal_greedy_ctx* c, int64_t history_start, int64_t n_gen) {
    if (!c || !c->tickets || !c->picked_host) return 0;
    const int64_t U = n_gen - history_start;
    if (history_start < 0 || U < 1 || U > c->max_len) return 0;
    const int K0 = c->E0 > 0 ? c->E0 : c->E;
    if (!(c->E % 16 == 0 && K0 % 8 == 0 && (reinterpret_cast<uintptr_t>(c->emb) & 15) == 0 && (!c->proj_t || (reinterpret_cast<uintptr_t>(c->proj_t) & 15) == 0)))
        return 0;
    return greedy_group_ok((int)U, c->S, c->E, c->H, c->FF) ? 1 : 0;
}

extern "C" int tal_greedy_step_multi_fwd(tal_greedy_ctx* const* ctxs, const int64_t* history_start, const int64_t* n_gen, int G, void* stream) {
    TAL_CHECK_ARG(ctxs && history_start && n_gen && G >= 1 && G <= TAL_GROUP_MAX, "tal_greedy_step_multi_fwd: 1..%d sessions", TAL_GROUP_MAX);
    const tal_greedy_ctx* c0 = ctxs[0];
    TAL_CHECK_ARG(c0, "tal_greedy_step_multi_fwd: null context");
    const int E = c0->E, H = c0->H, FF = c0->FF, V = c0->V, E0 = c0->E0, L = c0->n_layers, K0 = E0 > 0 ? E0 : E;
    hipStream_t s = (hipStream_t)stream;
    SessionLayerIo io[TAL_GROUP_MAX];
    float* h0[TAL_GROUP_MAX];
    float* h1[TAL_GROUP_MAX];
    float* probs[TAL_GROUP_MAX];
    float* pick_part[TAL_GROUP_MAX];
    EmbedMulti em = {};
    int umax = 0;
    for (int i = 0; i < G; ++i) {
        tal_greedy_ctx* c = ctxs[i];
        TAL_CHECK_ARG(c && c->layers && c->emb && c->pe && c->k_cache && c->vt_cache && c->tokens && c->workspace && c->picked_dev && c->picked_host && c->tickets,
                      "tal_greedy_step_multi_fwd: session %d: null pointer", i);
        TAL_CHECK_ARG(c->layers == c0->layers && c->E == E && c->H == H && c->FF == FF && c->V == V && c->E0 == E0 && c->n_layers == L &&
                          c->emb == c0->emb && c->proj == c0->proj && c->proj_t == c0->proj_t && c->pe == c0->pe && c->max_len == c0->max_len,
                      "tal_greedy_step_multi_fwd: session %d decodes with another model", i);
        for (int j = 0; j < i; ++j)
            TAL_CHECK_ARG(ctxs[j] != c && ctxs[j]->workspace != c->workspace && ctxs[j]->tickets != c->tickets,
                          "tal_greedy_step_multi_fwd: sessions %d and %d share a context, workspace or tickets", j, i);
        TAL_CHECK_ARG(tal_greedy_group_ok(c, history_start[i], n_gen[i]),
                      "tal_greedy_step_multi_fwd: session %d (prefix [%lld, %lld), window %d) does not take the merged kernels' forms: step it alone",
                      i, (long long)history_start[i], (long long)n_gen[i], c->S);
        if (int rr = greedy_reset_if_needed(c, s)) return rr;
        const int U = (int)(n_gen[i] - history_start[i]), S = c->S;
        if (c->workspace_bytes < tal_greedy_step_workspace_bytes(U, S, E, H, FF, V, E0, L)) {
            set_error("tal_greedy_step_multi_fwd: session %d: workspace %zu < %zu bytes", i, c->workspace_bytes, tal_greedy_step_workspace_bytes(U, S, E, H, FF, V, E0, L));
            return TAL_ENOMEM;
        }
        // the same carving of the session's workspace as tal_greedy_step_fwd
        float* base = reinterpret_cast<float*>(c->workspace);
        LayerWs ws = carve(base, 1, U, S, E, H, FF);
        float* p = base + ws.total_floats;
        h0[i] = p; p += up64((size_t)U * E);
        h1[i] = p; p += up64((size_t)U * E);
        p += up64((size_t)L * U * S);                      // (avg: the batched-GEMM layer's rows, unused here)
        probs[i] = p; p += up64((size_t)L * H * S);
        p += up64((size_t)(E0 > 0 ? E0 : E));
        p += up64((size_t)V);
        DecodeScratch sk = {p, c->tickets};
        p += up64(decode_scratch_floats(U, S, E, H));
        pick_part[i] = p;
        io[i].U = U; io[i].S = S; io[i].mem_kpm = c->mem_kpm; io[i].ws = ws; io[i].sk = sk; io[i].no_fold = c->no_fold != 0;
        em.tokens[i] = c->tokens + history_start[i];
        em.out[i] = h0[i];
        em.U[i] = U;
        umax = U > umax ? U : umax;
        if (!c->picked_host_dev) {
            void* alias = nullptr;
            if (hipHostGetDevicePointer(&alias, c->picked_host, 0) != hipSuccess || !alias) {
                set_error("tal_greedy_step_multi_fwd: session %d: picked_host is not mapped pinned host memory (%s)", i, hipGetErrorString(hipGetLastError()));
                return TAL_EINVAL;
            }
            c->picked_host_dev = reinterpret_cast<float*>(alias);
            reinterpret_cast<volatile unsigned*>(c->picked_host)[1 + S] = 0u;
        }
    }
    TAL_CHECK_ARG(K0 <= 8192, "tal_greedy_step_multi_fwd: embedding width %d too large", K0);
    hipLaunchKernelGGL(embed_multi_kernel, dim3((unsigned)umax, (unsigned)G), dim3(256), (size_t)K0 * sizeof(float), s, em, c0->emb,
                       E0 > 0 ? c0->proj : nullptr, c0->pe, V, K0, E);
    TAL_CHECK_LAUNCH("tal_greedy_step_multi_fwd(embed)");
    for (int l = 0; l < L; ++l) {
        for (int i = 0; i < G; ++i) {
            const tal_greedy_ctx* c = ctxs[i];
            TAL_CHECK_ARG(c->k_cache[l] && c->vt_cache[l], "tal_greedy_step_multi_fwd: session %d, layer %d has no cached K / V^T", i, l);
            io[i].tgt = l == 0 ? h0[i] : h1[i];
            io[i].out = h1[i];
            io[i].ck = c->k_cache[l];
            io[i].k_pitch = c->k_pitch ? c->k_pitch : E;
            io[i].cvt = c->vt_cache[l];
            io[i].probs_last = probs[i] + (size_t)l * H * io[i].S;
        }
        const int rc = decoder_layer_small_multi(&c0->layers[l], io, G, E, H, FF, s);
        if (rc) return rc;
    }
    ArgPack<LmPickArgs> pk;
    pk.n = G;
    for (int i = 0; i < G; ++i) {
        tal_greedy_ctx* c = ctxs[i];
        if (++c->seq == 0) ++c->seq;
        LmPickArgs& q = pk.a[i];
        q = LmPickArgs{};
        q.h = h1[i] + (size_t)(io[i].U - 1) * E;
        q.attn = probs[i];
        q.layer_stride = (int64_t)H * io[i].S;
        q.head_stride = (int64_t)io[i].S;
        q.S = io[i].S;
        q.partial = pick_part[i];
        q.ticket_word = c->tickets + (TAL_GREEDY_TICKETS - 1);
        q.out = c->picked_host_dev;
        q.token_out = c->tokens + n_gen[i];
        q.host_seq = c->seq;
        q.bias = c->pick_bias;
    }
    hipLaunchKernelGGL(lm_pick_multi_kernel, dim3((unsigned)cdiv(V, LMP_ROWS), (unsigned)G), dim3(256), (size_t)(E + K0 + LMP_ROWS) * sizeof(float), s, pk,
                       E0 > 0 ? c0->proj_t : nullptr, E, K0, c0->emb, V, L, H);
    TAL_CHECK_LAUNCH("tal_greedy_step_multi_fwd(lm head + pick)");
    return TAL_OK;
}

// ---- windows of an episode-wide K | V table -----------------------------------------------------------------------------------
// The cross-attention K and V of an encoder frame do not depend on the window it is seen through, so a session projects the WHOLE
// episode once (one dense layer per decoder layer: enc [T', E] x in_proj_weight[E : 3E]^T -> table [T', 2E] = K | V per frame;
// 737 MB per hour of audio for the reference's 4 x 512 decoder) and a window is a VIEW: K rows are read in place (pitch 2E), the
// key-padding bytes too; only V^T -- the P.V kernels want V transposed, 16-byte aligned at an arbitrary first frame -- is
// materialised, by one small transposing launch for all layers.  A window move costs that one launch and no host work beyond
// rewriting the context's pointers, instead of 12 launches (K, memset, V^T per layer) and fresh buffers.
struct WindowVt {
    const float* kv[8];
    float* vt[8];
};
__global__ __launch_bounds__(256) void window_vt_kernel(const WindowVt p, int64_t frame0, int S, int S4, int E, int64_t kv_pitch) {
    __shared__ float t[32][33];
    const int l = blockIdx.z, s0 = blockIdx.x * 32, e0 = blockIdx.y * 32;
    const float* src = p.kv[l] + E;                   // the V half of a table row
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int sidx = s0 + i, e = e0 + threadIdx.x;
        t[i][threadIdx.x] = (sidx < S && e < E) ? src[(frame0 + sidx) * kv_pitch + e] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int e = e0 + i, sidx = s0 + threadIdx.x;
        if (e < E && sidx < S4) p.vt[l][(int64_t)e * S4 + sidx] = t[threadIdx.x][i];     // (pad columns S .. S4: zeros)
    }
}

extern "C" int tal_window_vt_fwd(const float* const* kv_all, int n_layers, int64_t frame0, int S, int E, int64_t kv_pitch,
                                 float* const* vt, void* stream) {
    TAL_CHECK_ARG(kv_all && vt && n_layers >= 1 && n_layers <= 8 && frame0 >= 0 && S >= 1 && E >= 1 && kv_pitch >= 2 * (int64_t)E,
                  "tal_window_vt_fwd: bad argument");
    WindowVt p = {};
    for (int l = 0; l < n_layers; ++l) {
        TAL_CHECK_ARG(kv_all[l] && vt[l], "tal_window_vt_fwd: layer %d: null pointer", l);
        p.kv[l] = kv_all[l];
        p.vt[l] = vt[l];
    }
    const int S4 = (int)pad4(S);
    hipLaunchKernelGGL(window_vt_kernel, dim3((unsigned)cdiv(S4, 32), (unsigned)cdiv(E, 32), (unsigned)n_layers), dim3(32, 8), 0,
                       (hipStream_t)stream, p, frame0, S, S4, E, kv_pitch);
    TAL_CHECK_LAUNCH("tal_window_vt_fwd");
    return TAL_OK;
}

// point a context at the window [frame0, frame0 + c->S) of its episode-wide table (c->kv_all): K / key-padding views + V^T launch
extern "C" int tal_greedy_set_window(tal_greedy_ctx* c, int64_t frame0, void* stream) {
    TAL_CHECK_ARG(c && c->kv_all && c->k_cache && c->vt_cache && c->kv_pitch >= 2 * (int64_t)c->E && c->k_pitch == c->kv_pitch,
                  "tal_greedy_set_window: the context has no episode-wide K | V table");
    TAL_CHECK_ARG(frame0 >= 0 && frame0 + c->S <= c->enc_frames, "tal_greedy_set_window: window [%lld, %lld) outside the episode's %lld frames",
                  (long long)frame0, (long long)(frame0 + c->S), (long long)c->enc_frames);
    const float** kc = const_cast<const float**>(c->k_cache);             // (the caller's array: n_layers mutable slots)
    for (int l = 0; l < c->n_layers; ++l) kc[l] = c->kv_all[l] + frame0 * c->kv_pitch;
    c->mem_kpm = c->kpm_all ? c->kpm_all + frame0 : nullptr;
    return tal_window_vt_fwd(c->kv_all, c->n_layers, frame0, c->S, c->E, c->kv_pitch, const_cast<float* const*>(c->vt_cache), stream);
}

// ---- host side of the sliding-window loop (System.generate_unaligned, tal/asr/system.py:389-521) ----------------------------
extern "C" int tal_unaligned_consume(tal_unaligned_state* st, int64_t token, const float* attn, int S) {
    TAL_CHECK_ARG(st && st->gen && st->rec_chunk_start && st->rec_attn && st->rec_len && attn, "tal_unaligned_consume: null pointer");
    TAL_CHECK_ARG(S >= 1 && S <= st->rec_stride && st->n >= 1 && st->n < st->gen_cap && st->n_rec < st->rec_cap && st->n_rec == st->n - 1,
                  "tal_unaligned_consume: buffers full or inconsistent (n=%lld of %lld, records %lld of %lld, S=%d of %d)", (long long)st->n,
                  (long long)st->gen_cap, (long long)st->n_rec, (long long)st->rec_cap, S, st->rec_stride);
    int64_t n = st->n, history_start = st->history_start, chunk_start = st->chunk_start;
    const int64_t hist_len = n - history_start, chunk_size = st->chunk_size;
    const int64_t chunk_before = chunk_start;
    int flags = 0;
    st->gen[n++] = token;
    // record: window start (rewritten below, as the reference's aliased tensor is) + the attention row
    int64_t rec = st->n_rec++;
    st->rec_chunk_start[rec] = chunk_start;
    st->rec_len[rec] = S;
    float* row = st->rec_attn + rec * (int64_t)st->rec_stride;
    // progress = sum_i attn[i] * (i / S): float32 products (system.py:405-408), summed in double and rounded ONCE to float32.  The
    // reference leaves the order of its float32 sum to torch (a vectorised reduction on the CPU, a tree on CUDA: its own devices differ
    // in the last place); the exactly-rounded sum is the one value every such order is within an ulp of, and the decisions below on
    // rows a few ulps from the thresholds are pinned by tests/test_host_logic.py::test_unaligned_consume_progress_next_to_the_threshold
    double acc = 0.0;
    const float fS = (float)S;
    for (int i = 0; i < S; ++i) {
        row[i] = attn[i];
        acc += (double)(attn[i] * ((float)i / fS));
    }
    const double progress = (double)(float)acc;
    if (progress > st->highest_progress) {
        st->num_no_improve = 0;
        if (st->window_time > 5) st->highest_progress = progress;
    } else {
        st->num_no_improve += 1;
    }
    const bool stalling = st->num_no_improve >= st->stall_patience;
    const bool repeating = tal_ngram_repeat_count(st->gen + history_start, hist_len, st->rep_n) > (int64_t)st->rep_n * 2;
    const bool last_chunk = st->encoder_len - chunk_start <= chunk_size;
    const bool reset = stalling || repeating;
    bool kept = true;
    if (!last_chunk) {
        if (reset) {
            chunk_start += st->skip_frames;
            if (repeating) {
                const int64_t back = 2 * (int64_t)st->rep_n - 1;
                n -= back;
                st->n_rec -= back;
                kept = false;                              // this step's record is among the ones rolled back
            }
            st->gen[n - 1] = st->eos;
            flags |= TAL_UNALIGNED_PREFIX_REWRITTEN;
            history_start = n - 1;
            st->highest_progress = 0.0;
            st->window_time = 0;
        } else if (progress > st->thresh_prct) {
            const int64_t history_size = n - history_start;
            chunk_start += st->shift_frames;
            history_start += (int64_t)floorf(st->del_prct * (float)(history_size - 1));
            st->highest_progress = 0.0;
            st->window_time = 0;
        }
    }
    if (kept) st->rec_chunk_start[rec] = chunk_start;     // the post-advance, pre-clamp value (system.py:400,441,468)
    if (chunk_start > st->encoder_len - chunk_size) chunk_start = st->encoder_len - chunk_size;
    const int64_t floor_hist = n - st->max_positions > 0 ? n - st->max_positions : 0;
    if (history_start < floor_hist) history_start = floor_hist;
    TAL_CHECK_ARG(history_start < n && n - history_start <= st->max_positions && st->n_rec == n - 1,
                  "tal_unaligned_consume: invalid history start %lld of %lld tokens", (long long)history_start, (long long)n);
    st->window_time += 1;
    st->n = n;
    st->history_start = history_start;
    st->chunk_start = chunk_start;
    st->it += 1;
    if (chunk_start != chunk_before) flags |= TAL_UNALIGNED_WINDOW_MOVED;
    if ((reset && last_chunk) || st->it >= st->max_iters) flags |= TAL_UNALIGNED_DONE;
    if (n + 1 >= st->gen_cap || st->n_rec + 1 >= st->rec_cap) flags |= TAL_UNALIGNED_GROW;
    st->flags |= flags;
    return flags;
}

// what the control flow asked for and the library can do itself: the window as a view of the episode-wide K | V table, the
// rewritten prefix from the (pinned) host token stream.  Clears the flags it served.
static int unaligned_serve(tal_unaligned_state* st, tal_greedy_ctx* c, int64_t dev_cap, void* stream) {
    if ((st->flags & TAL_UNALIGNED_WINDOW_MOVED) && !(st->flags & TAL_UNALIGNED_DONE) && c->kv_all && st->chunk_start >= 0 &&
        st->chunk_start + c->S <= c->enc_frames) {
        const int rc = tal_greedy_set_window(c, st->chunk_start, stream);
        if (rc) return rc;
        st->flags &= ~TAL_UNALIGNED_WINDOW_MOVED;
    }
    if ((st->flags & TAL_UNALIGNED_PREFIX_REWRITTEN) && !(st->flags & (TAL_UNALIGNED_DONE | TAL_UNALIGNED_WINDOW_MOVED)) && st->gen_pinned &&
        st->n <= dev_cap) {
        if (hipMemcpyAsync(c->tokens, st->gen, (size_t)st->n * sizeof(int64_t), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
            set_error("tal_unaligned_group_run: prefix upload failed: %s", hipGetErrorString(hipGetLastError()));
            return TAL_EHIP;
        }
        st->flags &= ~TAL_UNALIGNED_PREFIX_REWRITTEN;
    }
    return TAL_OK;
}

// ONE session on its own launches: step -> poll -> consume without leaving the library (System.generate_unaligned's solo loop between
// the decisions that need Python).  Keeping the NEXT step in flight as well -- enqueued for the prefix one token longer before this
// step's result is back, dropped when the control flow says otherwise -- was built and measured in round 6 and LOSES: the window
// moves in 9.4 % of the steps of an episode (538 of 5,721), every such step wastes a whole speculated step (~230 us), and a correct
// guess saves only the ~13 us between a pick and the next embed: 0.261 -> 0.274 ms per token on the 1-hour episode
// (profiles/r6_decode_two_steps_in_flight.txt).  Removed again.
static int unaligned_run_solo(tal_unaligned_state* st, tal_greedy_ctx* c, int64_t dev_cap, int max_steps, void* stream) {
    TAL_CHECK_ARG(c->picked_host && c->tickets && c->S > 0, "tal_unaligned_group_run: the session has no pinned result buffer / tickets");
    for (int step = 0; step < max_steps; ++step) {
        int rc = unaligned_serve(st, c, dev_cap, stream);
        if (rc) return rc;
        if (st->flags) return step;
        if (st->n + 1 >= st->gen_cap || st->n_rec + 1 >= st->rec_cap || st->n + 1 > dev_cap) {
            st->flags |= TAL_UNALIGNED_GROW;
            return step;
        }
        rc = tal_greedy_step_fwd(c, st->history_start, st->n, 3, stream);
        if (rc) return rc;
        const int got = tal_greedy_step_poll(c, 20000);
        if (got != 1) {
            if (got == 0) {
                const hipError_t e = hipStreamSynchronize((hipStream_t)stream);      // nothing may stay in flight behind an error
                set_error("tal_unaligned_group_run: no result after 20 s (stream after the wait: %s)", hipGetErrorString(e));
            }
            return TAL_EHIP;
        }
        const float* ph = c->picked_host;
        rc = tal_unaligned_consume(st, (int64_t)__builtin_bit_cast(int32_t, ph[0]), ph + 1, c->S);
        if (rc < 0) return rc;
    }
    return max_steps;
}

extern "C" int tal_unaligned_group_run(tal_unaligned_state* const* st, tal_greedy_ctx* const* ctxs, const int64_t* dev_cap, int G,
                                       int max_steps, void* stream) {
    TAL_CHECK_ARG(st && ctxs && dev_cap && G >= 1 && G <= TAL_GROUP_MAX && max_steps >= 1, "tal_unaligned_group_run: bad argument");
    if (G == 1) {
        TAL_CHECK_ARG(st[0] && ctxs[0], "tal_unaligned_group_run: null session");
        return unaligned_run_solo(st[0], ctxs[0], dev_cap[0], max_steps, stream);
    }
    int64_t hs[TAL_GROUP_MAX], ng[TAL_GROUP_MAX];
#ifdef TAL_GROUP_TIMING      // (ablation build: where a merged step's wall time goes, printed every 2000 steps)
    static thread_local double t_launch = 0, t_poll = 0, t_consume = 0, u_max_sum = 0, u_sum = 0;
    static thread_local long n_steps = 0, n_sess = 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
#define GT(var, expr) do { const double t0__ = now(); expr; var += now() - t0__; } while (0)
#else
#define GT(var, expr) do { expr; } while (0)
#endif
    for (int step = 0; step < max_steps; ++step) {
        bool go = true;
        for (int i = 0; i < G; ++i) {
            TAL_CHECK_ARG(st[i] && ctxs[i], "tal_unaligned_group_run: null session %d", i);
            {
                const int rc = unaligned_serve(st[i], ctxs[i], dev_cap[i], stream);
                if (rc) return rc;
            }
            if (st[i]->flags) { go = false; continue; }
            if (st[i]->n + 1 >= st[i]->gen_cap || st[i]->n_rec + 1 >= st[i]->rec_cap || st[i]->n + 1 > dev_cap[i]) {
                st[i]->flags |= TAL_UNALIGNED_GROW;
                go = false;
            } else if (G > 1 && !tal_greedy_group_ok(ctxs[i], st[i]->history_start, st[i]->n)) {
                st[i]->flags |= TAL_UNALIGNED_ALONE;
                go = false;
            }
            hs[i] = st[i]->history_start;
            ng[i] = st[i]->n;
        }
        if (!go) return step;
        int rc;
        GT(t_launch, rc = tal_greedy_step_multi_fwd(ctxs, hs, ng, G, stream));
        if (rc) return rc;
#ifdef TAL_GROUP_TIMING
        n_steps += 1; n_sess += G;
        {
            int64_t um = 0;
            for (int i = 0; i < G; ++i) { const int64_t u = ng[i] - hs[i]; um = u > um ? u : um; u_sum += (double)u; }
            u_max_sum += (double)um;
        }
        if (n_steps % 2000 == 0)
            fprintf(stderr, "[group timing] %ld steps, %.2f sessions per step, prefix mean %.1f / longest of a step %.1f tokens: launch %.1f us, poll %.1f us, consume + serve %.1f us per step\n",
                    n_steps, (double)n_sess / n_steps, u_sum / n_sess, u_max_sum / n_steps, t_launch / n_steps, t_poll / n_steps, t_consume / n_steps);
#endif
        bool flagged = false;
        for (int i = 0; i < G; ++i) {
            int got;
            GT(t_poll, got = tal_greedy_step_poll(ctxs[i], 20000));
            if (got != 1) {
                const hipError_t e = hipStreamSynchronize((hipStream_t)stream);      // nothing may stay in flight behind an error
                set_error("tal_unaligned_group_run: session %d: no result after 20 s (stream after the wait: %s)", i, hipGetErrorString(e));
                return TAL_EHIP;
            }
            const float* ph = ctxs[i]->picked_host;
            GT(t_consume, rc = tal_unaligned_consume(st[i], (int64_t)__builtin_bit_cast(int32_t, ph[0]), ph + 1, ctxs[i]->S));
            if (rc < 0) return rc;
            if (st[i]->flags) {
                GT(t_consume, rc = unaligned_serve(st[i], ctxs[i], dev_cap[i], stream));
                if (rc) return rc;
            }
            flagged = flagged || st[i]->flags != 0;
        }
        if (flagged) return step + 1;
    }
    return max_steps;
}
