// Device-side BODIES of the decode step's kernels (csrc/decode_small.hip, csrc/decoder.hip): one __device__ function per kernel,
// taking the coordinates of a block of the problem's OWN grid (`Blk`).  Three launch forms call them: the ordinary kernels (one
// problem per launch), the MULTI kernels (the same problem of several decode sessions in one launch) and the one-launch decode step
// (csrc/decode_persist.hip: a persistent workgroup walks the step's phases and calls the bodies for the blocks dealt to it).  A
// problem's outputs are the same instructions on the same operands in every form: bit-identical.
//
// COH (one-launch step only): the outputs another workgroup reads LATER IN THE SAME LAUNCH are stored write-through at agent scope
// (`sc1`: visible to every XCD without a cache write-back); the consumer side is an acquire fence behind the phase barrier, after
// which the bodies' plain loads are fresh.  Values are unaffected.
#pragma once
#include "common.h"

namespace tal {

template <bool COH>
__device__ __forceinline__ void store_f32(float* p, float v) {
    if (COH) st_agent(p, v);
    else *p = v;
}
template <bool COH>
__device__ __forceinline__ void store_f32x4(float* p, const f32x4& v) {
    if (COH) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<f32x4*>(p) = v;
}

// mode 0: acc+b | 1: relu(acc+b) | 2: res + alpha*(acc+b) | 3: alpha*(acc+b) on columns < scale_cols
// One workgroup = 16 output columns x up to 32 rows (blockIdx.y picks the 32-row half).  NW waves split K; a wave's whole
// K slice (<= 8 chunks of 16 k when K <= 128 NW) is requested before its first MFMA, so a launch costs ONE operand round
// trip -- a CU pulls only ~25-60 GB/s, and the round trip, not the arithmetic, is what a launch this small pays for.
// The kernels of this file exist in two launch forms that share one body each: the ordinary one (one problem per launch) and
// the MULTI form, which runs the same body for up to TAL_GROUP_MAX independent problems -- the decode steps of several
// sessions (System.transcribe_unaligned_many) -- in ONE launch: the argument structs travel by value in the kernel argument
// segment, a grid dimension picks the problem, a block outside its problem's own grid returns.  A problem's outputs are the
// same instructions on the same operands in both forms, i.e. bit-identical.  `Blk` carries the block coordinates (and the
// grid extents the body derives indices from) of the problem's OWN grid.
struct Blk {
    unsigned x, y, z, gx, gy;
};

template <int MODE, int MT, int NW, bool COH = false>
__device__ __forceinline__ void skinny_gemm_body(const SkinnyArgs& g, const Blk blk) {
    __shared__ __attribute__((aligned(16))) float part[NW * MT * 256];   // [wave][m tile][row 16][col 16]
    const int n0 = blk.x * 16;
    const int m0 = blk.y * 32;
    const int lane = threadIdx.x & 63, w = wave_id();
    const int r16 = lane & 15, kq = lane >> 4;
    const int KS = g.ksplit > 1 ? g.ksplit : 1;
    const int Kw = g.K / (NW * KS);                // this wave's share of K
    const int nchunk = Kw >> 4;                    // 16 k per chunk = 4 MFMAs
    const int kofs = ((int)blk.z * NW + w) * Kw;
    const float* wp = g.W + (int64_t)(n0 + r16) * g.ldw + kofs + 4 * kq;
    const float* ap[MT];
    const bool seg2 = g.A2 && kofs >= g.K1;         // (wave-uniform: this wave's K slice lies in the second column segment of A)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = m0 + mt * 16 + r16;
        const int64_t rc = row < g.M ? row : g.M - 1;
        ap[mt] = seg2 ? g.A2 + rc * g.lda2 + (kofs - g.K1) + 4 * kq : g.A + rc * g.lda + kofs + 4 * kq;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = {0.f, 0.f, 0.f, 0.f};
    const bool idle = seg2 && n0 < g.k1_cols;      // (wave-uniform: this column block stops at K1 -- a zero partial tile)
    // A lane's 16-byte load holds k = 4 kq + {0..3} of a chunk; MFMA j of the chunk contracts the k set {j, 4+j, 8+j, 12+j}
    // on both operands, so the chunk's four MFMAs cover its 16 k exactly once.
    constexpr int UNR = 8;
    constexpr int NBUF = NW >= 16 ? 1 : 2;        // 16 waves = 4 per SIMD = 128 registers each: one batch, no double buffer
    f32x4 bw[NBUF][UNR], av[NBUF][MT][UNR];
    auto load_batch = [&](int c0, int buf) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int c = c0 + u < nchunk ? c0 + u : nchunk - 1;     // (clamped: a tail batch re-reads the last chunk, unused)
            bw[buf][u] = *reinterpret_cast<const f32x4*>(wp + c * 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[buf][mt][u] = *reinterpret_cast<const f32x4*>(ap[mt] + c * 16);
        }
    };
    auto mfma_batch = [&](int c0, int buf) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (c0 + u < nchunk) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][mt][u].x, bw[buf][u].x, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][mt][u].y, bw[buf][u].y, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][mt][u].z, bw[buf][u].z, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][mt][u].w, bw[buf][u].w, acc[mt], 0, 0, 0);
                }
            }
    };
    if (idle) {
    } else {
    load_batch(0, 0);
    if constexpr (NBUF == 1) {
        for (int c0 = 0; c0 < nchunk; c0 += UNR) {
            mfma_batch(c0, 0);
            if (c0 + UNR < nchunk) load_batch(c0 + UNR, 0);
        }
    } else {
        for (int c0 = 0; c0 < nchunk; c0 += 2 * UNR) {
            if (c0 + UNR < nchunk) load_batch(c0 + UNR, NBUF - 1);
            mfma_batch(c0, 0);
            if (c0 + 2 * UNR < nchunk) load_batch(c0 + 2 * UNR, 0);
            if (c0 + UNR < nchunk) mfma_batch(c0 + UNR, NBUF - 1);
        }
    }
    }
    // C layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[((w * MT + mt) * 16 + 4 * kq + i) * 16 + r16] = acc[mt][i];
    __syncthreads();
    const int t = threadIdx.x;
    const int mt = t >> 6, r = (t & 63) >> 2, c4 = t & 3;
    const int m = m0 + mt * 16 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t < MT * 64) {
        v = *reinterpret_cast<const f32x4*>(&part[((0 * MT + mt) * 16 + r) * 16 + 4 * c4]);
#pragma unroll
        for (int q = 1; q < NW; ++q) v += *reinterpret_cast<const f32x4*>(&part[((q * MT + mt) * 16 + r) * 16 + 4 * c4]);   // wave order
    }
    if (KS > 1) {
        __shared__ unsigned ticket;
        const unsigned tile = blk.y * blk.gx + blk.x, ntile = blk.gx * blk.gy;
        float* mine = g.sk_part + ((size_t)blk.z * ntile + tile) * 512 + t * 4;
        if (t < MT * 64) { st_agent(mine, v.x); st_agent(mine + 1, v.y); st_agent(mine + 2, v.z); st_agent(mine + 3, v.w); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) ticket = take_ticket(&g.sk_tickets[tile]);
        __syncthreads();
        if (ticket != (unsigned)(KS - 1)) return;
        if (t == 0) reset_ticket(&g.sk_tickets[tile]);
        if (t < MT * 64) {
            const float* p0 = g.sk_part + (size_t)tile * 512 + t * 4;
            v = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < KS; ++q) {                 // split order
                const float* pq = p0 + (size_t)q * ntile * 512;
                v.x += ld_agent(pq); v.y += ld_agent(pq + 1); v.z += ld_agent(pq + 2); v.w += ld_agent(pq + 3);
            }
        }
    }
    if (t >= MT * 64 || m >= g.M) return;
    const int col = n0 + 4 * c4;
    if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
    if (col < g.k1_cols) v += *reinterpret_cast<const f32x4*>(g.A2 + (int64_t)m * g.lda2 + col);
    if (MODE == 1 && col >= g.relu_begin) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (MODE == 2) v = *reinterpret_cast<const f32x4*>(g.res + (int64_t)m * g.ldres + col) + g.alpha * v;
    if (MODE == 3 && (g.scale_cols == 0 || col < g.scale_cols)) v = g.alpha * v;
    if (g.Yt && col >= g.vt_begin) {
        const int b = m / g.U, u = m - b * g.U;
        float* yt = g.Yt + (int64_t)b * g.vt_bs + (int64_t)(col - g.vt_begin) * g.ldt + u;
        store_f32<COH>(yt, v.x);
        store_f32<COH>(yt + g.ldt, v.y);
        store_f32<COH>(yt + 2 * g.ldt, v.z);
        store_f32<COH>(yt + 3 * g.ldt, v.w);
        return;
    }
    store_f32x4<COH>(g.Y + (int64_t)m * g.ldy + col, v);
}


// ---------------------------------------------------------------------------------------------------------------
// 8 waves per workgroup.  Scores: wave w takes key blocks w, w + 8, ... three at a time (all 3 x HD/16 fragment loads in
// flight before the first MFMA).  P.V: one 16-feature block per wave (HD = 128), the V^T fragments of the next 8 key
// steps in flight while the current 8 are multiplied.
constexpr int ATT_NW = 8;
// NWV: the waves the workgroup really has.  The work is defined for ATT_NW LOGICAL waves; a workgroup of fewer (the one-launch
// decode step runs 4) takes the logical waves w, w + NWV, ... one after the other -- every output element is produced by the same
// instructions on the same operands as with 8 waves.
template <int HD, int NWV = ATT_NW, bool COH = false>
__device__ __forceinline__ void attn_small_body(const AttnArgs& g, const Blk blk) {
    static_assert(ATT_NW % NWV == 0, "logical waves per wave");
    extern __shared__ __attribute__((aligned(16))) float sc[];   // [16][SP] scores -> probabilities
    const int u0 = blk.x * 16, h = blk.y, b = blk.z;
    const int lane = threadIdx.x & 63, wv = wave_id();
    const int r16 = lane & 15, kq = lane >> 4;
    const int U = g.U, S = g.S;
    const int S16 = (S + 15) & ~15, SP = S16 + 4;
    constexpr int NC = HD / 16;
    // The P.V operand (V^T) does not depend on the scores: its first fragments are requested before anything else, so
    // their round trip runs under the score / softmax phases (for the wave's first logical wave).
    constexpr int VU = 8;
    const int nstep = S16 / 16;
    const int S4 = (int)g.ldvt;
    f32x4 vb[3][VU];
    auto load_v = [&](const float* vrow, int t0, int buf) {
#pragma unroll
        for (int t = 0; t < VU; ++t) {
            const int s = (t0 + t) * 16 + 4 * kq;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t0 + t < nstep && s < S4) v = *reinterpret_cast<const f32x4*>(vrow + (t0 + t) * 16);   // ldvt % 4 == 0: all 16 bytes in the row
            if (s + 0 >= S) v.x = 0.f;
            if (s + 1 >= S) v.y = 0.f;
            if (s + 2 >= S) v.z = 0.f;
            if (s + 3 >= S) v.w = 0.f;
            vb[buf][t] = v;
        }
    };
    auto vrow_of = [&](int w) {
        const int cb = w < NC ? w : NC - 1;
        return g.vt + (int64_t)b * g.vt_bs + (int64_t)(h * HD + cb * 16 + r16) * g.ldvt + 4 * kq;
    };
    {
        const float* vrow = vrow_of(wv);
        load_v(vrow, 0, 0);
        load_v(vrow, VU, 1);
        load_v(vrow, 2 * VU, 2);
    }
    // ---- scores
#pragma unroll
    for (int w = wv; w < ATT_NW; w += NWV) {
        const int qr = u0 + r16 < U ? u0 + r16 : U - 1;
        const float* qrow = g.q + (int64_t)b * g.q_bs + (int64_t)qr * g.ldq + h * HD + 4 * kq;
        f32x4 qa[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) qa[c] = *reinterpret_cast<const f32x4*>(qrow + 16 * c);
        constexpr int JU = HD <= 64 ? 4 : 3;
        const int nblk = S16 / 16;
        for (int jb0 = w; jb0 < nblk; jb0 += ATT_NW * JU) {
            f32x4 kb[JU][NC];
#pragma unroll
            for (int ju = 0; ju < JU; ++ju) {
                const int jb = jb0 + ju * ATT_NW;
                int kr = jb * 16 + r16;
                kr = kr < S ? kr : S - 1;
                const float* krow = g.k + (int64_t)b * g.k_bs + (int64_t)kr * g.ldk + h * HD + 4 * kq;
#pragma unroll
                for (int c = 0; c < NC; ++c) kb[ju][c] = *reinterpret_cast<const f32x4*>(krow + 16 * c);
            }
#pragma unroll
            for (int ju = 0; ju < JU; ++ju) {
                const int jb = jb0 + ju * ATT_NW;
                if (jb >= nblk) break;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c].x, kb[ju][c].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c].y, kb[ju][c].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c].z, kb[ju][c].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c].w, kb[ju][c].w, acc, 0, 0, 0);
                }
                const int key = jb * 16 + r16;            // C: row = 4 kq + i (query), col = r16 (key)
                const bool dead = key >= S || (g.kpm && g.kpm[(int64_t)b * S + key]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int u = u0 + 4 * kq + i;
                    float v = acc[i];
                    if (g.mask && u < U && key < S) v += g.mask[(int64_t)u * S + key];
                    sc[(4 * kq + i) * SP + key] = dead ? -INFINITY : v;
                }
            }
        }
    }
    __syncthreads();
    // ---- row softmax (max, exp, sum, divide, as torch): logical wave w owns rows 2 w and 2 w + 1, one per 32-lane half, so the two rows'
    // cross-lane reductions (a shuffle is ~100 cycles of latency) run side by side
#pragma unroll
    for (int w = wv; w < ATT_NW; w += NWV) {
        constexpr int LPR = 64 / (16 / ATT_NW);             // lanes per row
        const int row = (16 / ATT_NW) * w + lane / LPR, ll = lane % LPR, u = u0 + row;
        float* p = sc + row * SP;
        if (u >= U) {
            for (int s = ll; s < S16; s += LPR) p[s] = 0.f;
        } else {
            float m = -INFINITY;
            for (int s = ll; s < S; s += LPR) m = fmaxf(m, p[s]);
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            float sum = 0.f;
            for (int s = ll; s < S; s += LPR) {
                const float e = expf(p[s] - m);
                p[s] = e;
                sum += e;
            }
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            float* pr = (g.probs && u >= g.prob_row0)
                            ? g.probs + (((int64_t)b * g.H + h) * (U - g.prob_row0) + (u - g.prob_row0)) * S : nullptr;
            for (int s = ll; s < S16; s += LPR) {
                const float v = s < S ? p[s] / sum : 0.f;
                p[s] = v;
                if (pr && s < S) store_f32<COH>(pr + s, v);
            }
        }
    }
    __syncthreads();
    // ---- ctx = P . V: feature block cb = logical wave (logical waves >= HD / 16 have none); the first one's batches 0..2 are
    // already in registers
#pragma unroll
    for (int w = wv; w < ATT_NW; w += NWV) {
        if (w >= NC) break;
        const float* vrow = vrow_of(w);
        if (w != wv) {
            load_v(vrow, 0, 0);
            load_v(vrow, VU, 1);
            load_v(vrow, 2 * VU, 2);
        }
        const float* prow = sc + r16 * SP + 4 * kq;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        auto mfma_v = [&](int t0, int buf) {
#pragma unroll
            for (int t = 0; t < VU; ++t)
                if (t0 + t < nstep) {
                    const f32x4 pa = *reinterpret_cast<const f32x4*>(prow + (t0 + t) * 16);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.x, vb[buf][t].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.y, vb[buf][t].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.z, vb[buf][t].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.w, vb[buf][t].w, acc, 0, 0, 0);
                }
        };
        for (int t0 = 0; t0 < nstep; t0 += 3 * VU) {
            mfma_v(t0, 0);
            if (t0 + 3 * VU < nstep) load_v(vrow, t0 + 3 * VU, 0);
            if (t0 + VU < nstep) mfma_v(t0 + VU, 1);
            if (t0 + 4 * VU < nstep) load_v(vrow, t0 + 4 * VU, 1);
            if (t0 + 2 * VU < nstep) mfma_v(t0 + 2 * VU, 2);
            if (t0 + 5 * VU < nstep) load_v(vrow, t0 + 5 * VU, 2);
        }
        const int col = h * HD + w * 16 + r16;
        const float bv = g.vbias ? g.vbias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = u0 + 4 * kq + i;
            if (u < U) store_f32<COH>(&g.ctx[(int64_t)b * g.c_bs + (int64_t)u * g.ldc + col], acc[i] + bv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same attention with the KEY axis cut over workgroups (cross-attention of a decode step: 357 keys x 512 features of
// K and V^T are 1.4 MB, and the 8 workgroups of the kernel above each pull 367 KB through one CU at 25-60 GB/s).
// Workgroup (row block, head, batch item x chunk) handles CB key blocks: scores, chunk-local max m_c, p = exp(s - m_c),
// l_c = sum p, o_c = p . V_c, all written to a scratch record; the last workgroup to arrive at (row block, head, item)
// -- ticket on a zero-initialised, self-resetting counter -- merges the chunks in chunk order,
//     M = max m_c,  w_c = exp(m_c - M),  ctx = sum_c w_c o_c / sum_c w_c l_c (+ v bias),
// and, for the rows >= prob_row0, the per-head probabilities p w_c / L.  Deterministic (fixed merge order).
#ifndef TAL_SPLIT_NCH          // (ablation builds, scripts/build_ablation.sh: key chunks per (row block, head); measured in round 6:
#define TAL_SPLIT_NCH 8        //  profiles/r6_cross_attention_chunks.txt)
#endif
constexpr int SPLIT_NCH = TAL_SPLIT_NCH;
constexpr int SPLIT_MAX_CB = (60 + SPLIT_NCH - 1) / SPLIT_NCH;      // key blocks per chunk (S <= 960 = 60 key blocks over SPLIT_NCH chunks)
__host__ __device__ static inline int split_cb(int S) { const int nblk = (S + 15) / 16; return (nblk + SPLIT_NCH - 1) / SPLIT_NCH; }
__host__ __device__ static inline size_t split_record_floats(int hd, int cb) { return (size_t)16 * hd + 32 + (size_t)16 * 16 * cb; }

template <int HD, bool COH = false>
__device__ __forceinline__ void attn_split_body(const AttnArgs& g, int CB, int NCH, float* __restrict__ scratch,
                                                unsigned* __restrict__ tickets, const Blk blk) {
    extern __shared__ __attribute__((aligned(16))) float sc[];   // [16][SPc] chunk scores -> exp(s - m_c)
    __shared__ float mrow[16], lrow[16], wsh[16 * SPLIT_NCH], Lsh[16];
    __shared__ unsigned ticket;
    const int u0 = blk.x * 16, h = blk.y;
    const int b = blk.z / NCH, ch = blk.z % NCH;
    const int lane = threadIdx.x & 63, w = wave_id();
    const int r16 = lane & 15, kq = lane >> 4;
    const int U = g.U, S = g.S;
    const int nblk = (S + 15) / 16;
    const int jb0 = ch * CB;                         // first key block of this chunk
    const int Sc = CB * 16, SPc = Sc + 4;
    constexpr int NC = HD / 16;
    const size_t rec_f = split_record_floats(HD, CB);
    const size_t group = ((size_t)b * blk.gy + h) * blk.gx + blk.x;      // (item, head, row block)
    float* rec = scratch + (group * NCH + ch) * rec_f;
    // V^T fragments of this chunk for the wave's feature blocks (NC / 4 of them): requested first
    constexpr int FB = (NC + 3) / 4;
    const int S4 = (int)g.ldvt;
    f32x4 vb[FB][SPLIT_MAX_CB];
#pragma unroll
    for (int f = 0; f < FB; ++f) {
        const int cb = w * FB + f < NC ? w * FB + f : NC - 1;
        const float* vrow = g.vt + (int64_t)b * g.vt_bs + (int64_t)(h * HD + cb * 16 + r16) * g.ldvt + 4 * kq;
#pragma unroll
        for (int t = 0; t < SPLIT_MAX_CB; ++t) {
            const int s = (jb0 + t) * 16 + 4 * kq;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < CB && s < S4) v = *reinterpret_cast<const f32x4*>(vrow + (jb0 + t) * 16);
            if (s + 0 >= S) v.x = 0.f;
            if (s + 1 >= S) v.y = 0.f;
            if (s + 2 >= S) v.z = 0.f;
            if (s + 3 >= S) v.w = 0.f;
            vb[f][t] = v;
        }
    }
    // ---- scores of the chunk: wave w takes key blocks w, w + 4
    {
        const int qr = u0 + r16 < U ? u0 + r16 : U - 1;
        const float* qrow = g.q + (int64_t)b * g.q_bs + (int64_t)qr * g.ldq + h * HD + 4 * kq;
        f32x4 qa[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) qa[c] = *reinterpret_cast<const f32x4*>(qrow + 16 * c);
        f32x4 kb[2][NC];
#pragma unroll
        for (int ju = 0; ju < 2; ++ju) {
            int kr = (jb0 + w + 4 * ju) * 16 + r16;
            kr = kr < S ? kr : S - 1;
            const float* krow = g.k + (int64_t)b * g.k_bs + (int64_t)kr * g.ldk + h * HD + 4 * kq;
#pragma unroll
            for (int c = 0; c < NC; ++c) kb[ju][c] = *reinterpret_cast<const f32x4*>(krow + 16 * c);
        }
#pragma unroll
        for (int ju = 0; ju < 2; ++ju) {
            const int jl = w + 4 * ju;                 // key block inside the chunk
            if (jl >= CB) break;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c].x, kb[ju][c].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c].y, kb[ju][c].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c].z, kb[ju][c].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c].w, kb[ju][c].w, acc, 0, 0, 0);
            }
            const int key = (jb0 + jl) * 16 + r16;
            const bool dead = key >= S || (g.kpm && g.kpm[(int64_t)b * S + key]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = u0 + 4 * kq + i;
                float v = acc[i];
                if (g.mask && u < U && key < S) v += g.mask[(int64_t)u * S + key];
                sc[(4 * kq + i) * SPc + jl * 16 + r16] = dead ? -INFINITY : v;
            }
        }
    }
    __syncthreads();
    // ---- chunk-local softmax statistics, wave w owns rows 4 w .. 4 w + 3: lane group (lane >> 4) takes one row, 16 lanes per
    // row, so the four rows' reductions run side by side (a cross-lane shuffle is ~100 cycles of latency; one row at a
    // time, 64 lanes wide, that is 48 dependent shuffles per wave)
    {
        const int row = 4 * w + (lane >> 4), l16 = lane & 15;
        float* p = sc + row * SPc;
        float m = -INFINITY;
        for (int s = l16; s < Sc; s += 16) m = fmaxf(m, p[s]);
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        float sum = 0.f;
        for (int s = l16; s < Sc; s += 16) {
            const float e = m == -INFINITY ? 0.f : expf(p[s] - m);     // a chunk whose keys are all masked contributes nothing
            p[s] = e;
            sum += e;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (l16 == 0) {
            mrow[row] = m;
            lrow[row] = sum;
        }
    }
    __syncthreads();
    // ---- o_c = p . V_c for the wave's feature blocks; record = [16][HD] o | [16] m | [16] l | [16][Sc] p
#pragma unroll
    for (int f = 0; f < FB; ++f) {
        const int cb = w * FB + f;
        if (cb >= NC) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < SPLIT_MAX_CB; ++t)
            if (t < CB) {
                const f32x4 pa = *reinterpret_cast<const f32x4*>(sc + r16 * SPc + t * 16 + 4 * kq);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.x, vb[f][t].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.y, vb[f][t].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.z, vb[f][t].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.w, vb[f][t].w, acc, 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) st_agent(&rec[(4 * kq + i) * HD + cb * 16 + r16], acc[i]);
    }
    if (threadIdx.x < 16) {
        st_agent(&rec[16 * HD + threadIdx.x], mrow[threadIdx.x]);
        st_agent(&rec[16 * HD + 16 + threadIdx.x], lrow[threadIdx.x]);
    }
    if (g.probs && u0 + 15 >= g.prob_row0)
        for (int i = threadIdx.x; i < 16 * Sc; i += 256) st_agent(&rec[16 * HD + 32 + i], sc[(i / Sc) * SPc + i % Sc]);
    // ---- published (write-through stores); take a ticket; the last arriver merges
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) ticket = take_ticket(&tickets[group]);
    __syncthreads();
    if (ticket != (unsigned)(NCH - 1)) return;
    if (threadIdx.x == 0) reset_ticket(&tickets[group]);   // ready for the next launch (stream order)
    const float* grp = scratch + group * NCH * rec_f;
    // Merge.  Every thread reads the (m_c, l_c) of the rows it needs itself and its chunks' o values in the same batch of
    // agent-scope loads: ONE round trip to memory, then arithmetic (fixed-trip loops: a run-time-bounded loop of memory
    // loads is a serial chain of round trips).
    if (threadIdx.x < 16) {
        const int row = threadIdx.x;
        float mc[SPLIT_NCH], lc[SPLIT_NCH];
#pragma unroll
        for (int c = 0; c < SPLIT_NCH; ++c) {
            mc[c] = c < NCH ? ld_agent(grp + c * rec_f + 16 * HD + row) : -INFINITY;
            lc[c] = c < NCH ? ld_agent(grp + c * rec_f + 16 * HD + 16 + row) : 0.f;
        }
        float M = -INFINITY;
#pragma unroll
        for (int c = 0; c < SPLIT_NCH; ++c) M = fmaxf(M, mc[c]);
        float L = 0.f;
#pragma unroll
        for (int c = 0; c < SPLIT_NCH; ++c) {
            const float wc = mc[c] == -INFINITY ? 0.f : expf(mc[c] - M);
            wsh[row * SPLIT_NCH + c] = wc;
            L += wc * lc[c];
        }
        Lsh[row] = L;
    }
    {   // every chunk's o tile for this thread's groups of 4 outputs: 16-byte agent-scope loads, all chunks in flight at once,
        // requested BEFORE the barrier that publishes the weights
        constexpr int NI = (16 * HD + 1023) / 1024;
        f32x4 oc[NI][SPLIT_NCH];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = (threadIdx.x + 256 * j) * 4;
            if (i < 16 * HD) {
                if (NCH == SPLIT_NCH) ld_agent_x4<SPLIT_NCH>(grp + i, rec_f, oc[j]);
                else {
#pragma unroll
                    for (int c = 0; c < SPLIT_NCH; ++c) {
                        const float* pc = grp + (c < NCH ? c : 0) * rec_f + i;      // (chunks past NCH: never used)
                        oc[j][c] = {ld_agent(pc), ld_agent(pc + 1), ld_agent(pc + 2), ld_agent(pc + 3)};
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = (threadIdx.x + 256 * j) * 4;
            if (i >= 16 * HD) continue;
            const int row = i / HD, col = i - row * HD;
            const int u = u0 + row;
            if (u >= U) continue;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < SPLIT_NCH; ++c)
                if (c < NCH) o += wsh[row * SPLIT_NCH + c] * oc[j][c];       // chunk order
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (g.vbias) bv = *reinterpret_cast<const f32x4*>(g.vbias + h * HD + col);
            const float L = Lsh[row];
            f32x4 r;
            r.x = o.x / L + bv.x; r.y = o.y / L + bv.y; r.z = o.z / L + bv.z; r.w = o.w / L + bv.w;
            store_f32x4<COH>(g.ctx + (int64_t)b * g.c_bs + (int64_t)u * g.ldc + h * HD + col, r);
        }
    }
    if (g.probs) {
        const int row_lo = g.prob_row0 > u0 ? g.prob_row0 - u0 : 0;
        const int row_hi = U - u0 < 16 ? U - u0 : 16;
        const int nrow = row_hi - row_lo;
        for (int i = threadIdx.x; i < nrow * NCH * Sc; i += 256) {
            const int row = row_lo + i / (NCH * Sc), rem = i % (NCH * Sc);
            const int c = rem / Sc, sl = rem - c * Sc;
            const int u = u0 + row, key = c * Sc + sl;
            if (key >= S) continue;
            const float pc = ld_agent(grp + c * rec_f + 16 * HD + 32 + row * Sc + sl);
            store_f32<COH>(&g.probs[(((int64_t)b * g.H + h) * (U - g.prob_row0) + (u - g.prob_row0)) * S + key], pc * wsh[row * SPLIT_NCH + c] / Lsh[row]);
        }
    }
}


// ---- token embedding: emb[tok] -> (proj) -> + pe[u] ---------------------------------------
template <bool COH = false>
__device__ __forceinline__ void embed_body(const int64_t* __restrict__ tokens, const float* __restrict__ emb,
                                           const float* __restrict__ proj, const float* __restrict__ pe,
                                           float* __restrict__ out, int U, int V, int E0, int D, int row) {
    extern __shared__ float e[];
    const int u = row % U;
    int64_t tok = tokens[row];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);  // validated on the host side; never trusted for addressing
    for (int k = threadIdx.x; k < E0; k += 256) e[k] = emb[tok * E0 + k];
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc;
        if (proj && E0 == 64 && (reinterpret_cast<uintptr_t>(proj) & 15) == 0) {
            // all 16 loads of the projection row in flight at once (a run-time-bounded scalar loop is 64 serial round trips)
            const float* pr = proj + (int64_t)d * 64;
            f32x4 pv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) pv[k] = *reinterpret_cast<const f32x4*>(pr + 4 * k);
            acc = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {                    // same fmaf chain order as the scalar loop
                acc = fmaf(e[4 * k], pv[k].x, acc); acc = fmaf(e[4 * k + 1], pv[k].y, acc);
                acc = fmaf(e[4 * k + 2], pv[k].z, acc); acc = fmaf(e[4 * k + 3], pv[k].w, acc);
            }
        } else if (proj) {
            acc = 0.f;
            const float* pr = proj + (int64_t)d * E0;
            for (int k = 0; k < E0; ++k) acc = fmaf(e[k], pr[k], acc);
        } else {
            acc = e[d];
        }
        store_f32<COH>(&out[(int64_t)row * D + d], acc + pe[(int64_t)u * D + d]);
    }
}


// Tied factorised LM head of the last position + the greedy pick in ONE launch (models.py:243-246, system.py:355-411):
// every workgroup recomputes t = P^T h (E0 x E, 128 KB, L2-resident), takes 128 vocabulary rows (logit = emb[v] . t),
// reduces them to its (max, first arg-max), and the last workgroup to arrive (ticket on a zeroed, self-resetting word)
// merges the partials in workgroup order -- lowest index on ties, as torch.argmax -- and writes the token plus the
// layer- / head-averaged attention row.  arg max of log_softmax(x) is taken as arg max of x (the same index unless two
// logits lie within an ulp of each other).
constexpr int LMP_ROWS = 128;
struct LmPickArgs {
    const float* h;          // last prefix row of the decoder output [E]
    const float* attn;       // attention rows: attn[l * layer_stride + hh * head_stride + i]
    int64_t layer_stride, head_stride;
    int S;
    float* partial;          // 2 floats per workgroup
    unsigned* ticket_word;
    float* out;              // {token, row [S] (, sequence word)}
    int64_t* token_out;
    unsigned host_seq;
    const float* bias;       // [V] added to the logits before the arg-max (LM shallow fusion, system.py:368-384) or NULL
};
__device__ __forceinline__ void lm_pick_body(const LmPickArgs& q, const float* __restrict__ proj_t, int E, int K0,
                                             const float* __restrict__ emb, int V, int n_layers, int H, const unsigned bx, const unsigned gx) {
    const float* __restrict__ h = q.h;
    const float* __restrict__ attn = q.attn;
    const int64_t layer_stride = q.layer_stride, head_stride = q.head_stride;
    const int S = q.S;
    float* __restrict__ partial = q.partial;
    unsigned* __restrict__ ticket_word = q.ticket_word;
    float* __restrict__ out = q.out;
    int64_t* __restrict__ token_out = q.token_out;
    const unsigned host_seq = q.host_seq;
    const float* __restrict__ bias = q.bias;
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [E] h | [K0] t | [128] logits
    float* hs = sm;
    float* ts = sm + E;
    float* lg = ts + K0;
    __shared__ unsigned ticket;
    const int tid = threadIdx.x, lane = tid & 63;
    // Everything this workgroup reads from memory is requested up front (h, P^T, its 128 embedding rows): the dependent
    // chain is one round trip, then arithmetic.  Sixteen lanes share a row, so one load instruction covers 4 rows x 256
    // contiguous bytes (a lane-per-row mapping touches 64 cache lines per instruction and the line rate sets the time).
    // Fast path: E = 512, K0 = 64 (the reference's '2x' model with the factorised embedding); otherwise plain loops.
    const int l16 = tid & 15, grp = tid >> 4;
    const bool fast = proj_t && E == 512 && K0 == 64;
    f32x4 pv[4][8], ev[8];
    if (fast) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                pv[ps][c] = *reinterpret_cast<const f32x4*>(proj_t + (int64_t)(grp + 16 * ps) * 512 + (l16 + 16 * c) * 4);
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int v = bx * LMP_ROWS + grp + 16 * ps;
            ev[ps] = *reinterpret_cast<const f32x4*>(emb + (int64_t)(v < V ? v : V - 1) * 64 + l16 * 4);
        }
    }
    for (int i = tid; i < E; i += 256) hs[i] = h[i];
    __syncthreads();
    if (fast) {
        float a[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            a[ps] = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 hv = *reinterpret_cast<const f32x4*>(hs + (l16 + 16 * c) * 4);
                a[ps] = fmaf(pv[ps][c].x, hv.x, a[ps]); a[ps] = fmaf(pv[ps][c].y, hv.y, a[ps]);
                a[ps] = fmaf(pv[ps][c].z, hv.z, a[ps]); a[ps] = fmaf(pv[ps][c].w, hv.w, a[ps]);
            }
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) a[ps] += __shfl_xor(a[ps], off, 64);
        if (l16 == 0)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) ts[grp + 16 * ps] = a[ps];
    } else if (proj_t) {
        // t[j] = sum_d h[d] proj_t[j][d]: four threads per j, a quarter of E each (E % 16 == 0)
        const int q = tid & 3, Eq = E >> 2;
        for (int j = tid >> 2; j < K0; j += 64) {
            const float* pr = proj_t + (int64_t)j * E + q * Eq;
            float a = 0.f;
            for (int d = 0; d < Eq; d += 4) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(pr + d);
                const f32x4 hv = *reinterpret_cast<const f32x4*>(hs + q * Eq + d);
                a = fmaf(p4.x, hv.x, a); a = fmaf(p4.y, hv.y, a); a = fmaf(p4.z, hv.z, a); a = fmaf(p4.w, hv.w, a);
            }
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            if (q == 0) ts[j] = a;
        }
    } else {
        for (int i = tid; i < K0; i += 256) ts[i] = hs[i];
    }
    __syncthreads();
    if (fast) {
        const f32x4 tv = *reinterpret_cast<const f32x4*>(ts + l16 * 4);
        float a[8];
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            a[ps] = ev[ps].x * tv.x;
            a[ps] = fmaf(ev[ps].y, tv.y, a[ps]); a[ps] = fmaf(ev[ps].z, tv.z, a[ps]); a[ps] = fmaf(ev[ps].w, tv.w, a[ps]);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1)
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) a[ps] += __shfl_xor(a[ps], off, 64);
        if (l16 == 0)
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int v = bx * LMP_ROWS + grp + 16 * ps;
                lg[grp + 16 * ps] = v < V ? (bias ? a[ps] + bias[v] : a[ps]) : -INFINITY;
            }
    } else {   // two threads per row, half of K0 each (K0 % 8 == 0)
        const int r = tid >> 1, half = tid & 1, Kh = K0 >> 1;
        const int v = bx * LMP_ROWS + r;
        float a = 0.f;
        if (v < V) {
            const float* er = emb + (int64_t)v * K0 + half * Kh;
            for (int d = 0; d < Kh; d += 4) {
                const f32x4 e4 = *reinterpret_cast<const f32x4*>(er + d);
                const f32x4 tv = *reinterpret_cast<const f32x4*>(ts + half * Kh + d);
                a = fmaf(e4.x, tv.x, a); a = fmaf(e4.y, tv.y, a); a = fmaf(e4.z, tv.z, a); a = fmaf(e4.w, tv.w, a);
            }
        }
        a += __shfl_xor(a, 1, 64);
        if (half == 0) lg[r] = v < V ? (bias ? a + bias[v] : a) : -INFINITY;
    }
    __syncthreads();
    if (tid < 64) {
        float best = lg[lane];
        int bi = lane;
        const float o = lg[lane + 64];
        if (o > best) { best = o; bi = lane + 64; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) {
            st_agent(&partial[2 * bx], best);
            st_agent(&partial[2 * bx + 1], __int_as_float(bx * LMP_ROWS + bi));
        }
    }
    __syncthreads();
    if (tid == 0) ticket = take_ticket(ticket_word);
    __syncthreads();
    if (ticket != gx - 1) return;
    if (tid == 0) reset_ticket(ticket_word);
    // attention row of the new token: mean over layers of (mean over heads), summed in layer / head order.
    // (fixed-trip loops with every load issued first: a run-time-bounded loop of loads is a serial chain of round trips)
    const float inv_h = 1.0f / (float)H;
    constexpr int LM = 8, HM = 8;                         // layers x heads held in registers per position (no run-time
    for (int i = tid; i < S; i += 256) {                  // division in the index arithmetic: 64 of them cost microseconds)
        float a = 0.f;
        if (n_layers <= LM && H <= HM) {
            float rv[LM][HM];
#pragma unroll
            for (int l = 0; l < LM; ++l)
#pragma unroll
                for (int hh = 0; hh < HM; ++hh)
                    rv[l][hh] = (l < n_layers && hh < H) ? attn[l * layer_stride + hh * head_stride + i] : 0.f;
#pragma unroll
            for (int l = 0; l < LM; ++l)
                if (l < n_layers) {
                    float al = rv[l][0];
#pragma unroll
                    for (int hh = 1; hh < HM; ++hh)
                        if (hh < H) al += rv[l][hh];
                    if (H > 1) al *= inv_h;
                    a = l == 0 ? al : a + al;
                }
        } else {
            for (int l = 0; l < n_layers; ++l) {
                const float* r = attn + l * layer_stride + i;
                float al = r[0];
                if (H > 1) {
                    for (int hh = 1; hh < H; ++hh) al += r[hh * head_stride];
                    al *= inv_h;
                }
                a = l == 0 ? al : a + al;
            }
        }
        out[1 + i] = a / (float)n_layers;
    }
    if (tid < 64) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        constexpr int PJ = 4;                             // up to 256 workgroups' partials, all loads in flight
        float pvv[PJ], pii[PJ];
#pragma unroll
        for (int k = 0; k < PJ; ++k) {
            const int j = lane + 64 * k;
            pvv[k] = j < (int)gx ? ld_agent(partial + 2 * j) : -INFINITY;
            pii[k] = j < (int)gx ? ld_agent(partial + 2 * j + 1) : __int_as_float(0x7fffffff);
        }
#pragma unroll
        for (int k = 0; k < PJ; ++k) {                    // ascending workgroup = ascending index: lowest index wins ties
            const int idx = __float_as_int(pii[k]);
            if (pvv[k] > best || (pvv[k] == best && idx < bi)) { best = pvv[k]; bi = idx; }
        }
        for (int j = lane + 64 * PJ; j < (int)gx; j += 64) {
            const float v = ld_agent(partial + 2 * j);
            const int idx = __float_as_int(ld_agent(partial + 2 * j + 1));
            if (v > best || (v == best && idx < bi)) { best = v; bi = idx; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) {
            bi = bi == 0x7fffffff ? 0 : bi;
            out[0] = __int_as_float(bi);
            if (token_out) *token_out = bi;
        }
    }
    if (host_seq) {
        // `out` is pinned host memory mapped into the device's address space: the result lands there without a copy
        // command, and the host, which polls the sequence word behind the row, sees it without a driver wake-up
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(out + 1 + S), host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


}  // namespace tal
