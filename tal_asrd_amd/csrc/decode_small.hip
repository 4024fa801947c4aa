// Latency-oriented kernels for the decoder's short problems (a decode step of System.generate_unaligned /
// System.generate, tal/asr/system.py:113,350: a few dozen prefix rows against 512-wide layers and a 357-frame
// memory window).  A step is a chain of dependent launches, each far too small to fill the chip, so what counts is
// (a) how many launches there are and (b) how long the dependent chain inside each one is.
//
//   skinny_gemm_kernel   Y = epilogue(X . W^T + b) for M <= 512 rows: one workgroup per 16 output columns x 32 rows, the
//       waves split K (one quarter each, every operand fragment a 16-byte global load issued up front, no LDS
//       staging), `v_mfma_f32_16x16x4_f32` (fp32 in / fp32 accumulate, 32 cycles), partial tiles summed through
//       LDS in wave order (deterministic).  N / 16 workgroups instead of N / 32 x M / 32: a 512 x 512 layer runs on
//       32 CUs with 32 MFMAs per SIMD.  Columns >= vt_begin can be stored transposed per batch item: the packed
//       q|k|v in-projection of nn.MultiheadAttention then yields V^T directly (the P.V product wants it).
//   attn_small_kernel    softmax(q k^T + masks) v for one (16-query block, head, batch item) per workgroup: scores
//       into LDS, the row softmax of torch (max, exp, sum, divide) in place, P.V from LDS against V^T -- one launch
//       where the batched-GEMM form needs three.  Optionally writes the per-head probabilities of the rows
//       >= prob_row0 (MultiheadAttention's returned weights are their head average, models.py:517-519).
#include "decode_bodies.h"

namespace tal {

template <int MODE, int MT, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_gemm_kernel(const SkinnyArgs g) {
    skinny_gemm_body<MODE, MT, NW>(g, Blk{blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y});
}

// multi form: grid (N / 16, max row tiles, problem x K slice); all problems share N, K and the K split (the weights' shape)
template <int MODE, int MT, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_gemm_multi_kernel(const ArgPack<SkinnyArgs> p, int KS) {
    const SkinnyArgs& g = p.a[blockIdx.z / KS];
    const unsigned gy = (unsigned)((g.M + 31) / 32);
    if (blockIdx.y >= gy) return;
    skinny_gemm_body<MODE, MT, NW>(g, Blk{blockIdx.x, blockIdx.y, blockIdx.z % KS, gridDim.x, gy});
}

// WIDE form of the multi launch, for merged steps with many row tiles (several sessions at long prefixes): one workgroup takes
// NT neighbouring 16-column blocks of its 32 rows.  The rows' operand fragments -- a wave's whole K slice of 128, 64 registers --
// are loaded ONCE and stay in registers while the W fragments of the NT blocks stream through (the next block's are in flight
// while the current one is multiplied): the narrow form re-reads the 32 rows for every 16 columns, N / 16 times per layer, which
// is two thirds of what a merged step of 8 sessions pulls from L2.  Per output the arithmetic is the narrow form's, instruction
// for instruction (same chunks in the same order per wave, waves added in wave order, K slices in slice order through the same
// partial buffers and tickets), so results are bit-identical to the narrow form and to a session's solo step.
template <int MODE, int MT, int NW, int NT>
__device__ __forceinline__ void skinny_gemm_wide_body(const SkinnyArgs& g, const Blk blk) {
    // Round 5: the NT blocks' products first, back to back, and then ONE pass of partial tiles -> LDS -> wave-order sums -> K-slice
    // hand-over (stores, one drain, NT tickets taken side by side) -> epilogues, instead of that whole chain once per block: a
    // merged step's 2048-deep layer took 20.8 us with four serial hand-overs per workgroup where a session alone takes 6.6.
    __shared__ __attribute__((aligned(16))) float part[NT * NW * MT * 256];   // [block][wave][m tile][row 16][col 16]
    __shared__ unsigned ticket[NT];
    constexpr int UNR = 8;                          // Kw == 128: the wave's K slice is exactly one batch of 8 chunks
    const int m0 = blk.y * 32;
    const int lane = threadIdx.x & 63, w = wave_id();
    const int r16 = lane & 15, kq = lane >> 4;
    const int KS = g.ksplit > 1 ? g.ksplit : 1;
    const int Kw = g.K / (NW * KS);
    const int kofs = ((int)blk.z * NW + w) * Kw;
    f32x4 av[MT][UNR], bw[2][UNR];
    const bool seg2 = g.A2 && kofs >= g.K1;         // (as in skinny_gemm_body: the second column segment of A)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = m0 + mt * 16 + r16;
        const int64_t rc = row < g.M ? row : g.M - 1;
        const float* ap = seg2 ? g.A2 + rc * g.lda2 + (kofs - g.K1) + 4 * kq : g.A + rc * g.lda + kofs + 4 * kq;
#pragma unroll
        for (int u = 0; u < UNR; ++u) av[mt][u] = *reinterpret_cast<const f32x4*>(ap + u * 16);
    }
    const int nb0 = blk.x * NT;                     // first 16-column block
    const bool idle = seg2 && nb0 * 16 < g.k1_cols; // (all NT blocks of a workgroup lie on one side of k1_cols: a multiple of 64)
    auto load_w = [&](int nb, int buf) {
        if (idle) return;
        const float* wp = g.W + (int64_t)(nb * 16 + r16) * g.ldw + kofs + 4 * kq;
#pragma unroll
        for (int u = 0; u < UNR; ++u) bw[buf][u] = *reinterpret_cast<const f32x4*>(wp + u * 16);
    };
    load_w(nb0, 0);
    const int t = threadIdx.x;
    const int mt_t = t >> 6, r = (t & 63) >> 2, c4 = t & 3;
    const int m = m0 + mt_t * 16 + r;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (nt + 1 < NT) load_w(nb0 + nt + 1, (nt + 1) & 1);
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = {0.f, 0.f, 0.f, 0.f};
        if (!idle)
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][u].x, bw[nt & 1][u].x, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][u].y, bw[nt & 1][u].y, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][u].z, bw[nt & 1][u].z, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][u].w, bw[nt & 1][u].w, acc[mt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[(((nt * NW + w) * MT + mt) * 16 + 4 * kq + i) * 16 + r16] = acc[mt][i];
    }
    __syncthreads();
    f32x4 v[NT];
    if (t < MT * 64) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            v[nt] = *reinterpret_cast<const f32x4*>(&part[(((nt * NW + 0) * MT + mt_t) * 16 + r) * 16 + 4 * c4]);
#pragma unroll
            for (int q = 1; q < NW; ++q) v[nt] += *reinterpret_cast<const f32x4*>(&part[(((nt * NW + q) * MT + mt_t) * 16 + r) * 16 + 4 * c4]);   // wave order
        }
    }
    bool finish[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) finish[nt] = true;
    if (KS > 1) {
        const unsigned gx = blk.gx * NT;            // the narrow form's grid extent: same tile numbers, partial buffers and tickets
        const unsigned ntile = gx * blk.gy;
        const unsigned tile0 = blk.y * gx + (unsigned)nb0;
        if (t < MT * 64) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float* mine = g.sk_part + ((size_t)blk.z * ntile + tile0 + nt) * 512 + t * 4;
                st_agent(mine, v[nt].x); st_agent(mine + 1, v[nt].y); st_agent(mine + 2, v[nt].z); st_agent(mine + 3, v[nt].w);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t < NT) ticket[t] = take_ticket(&g.sk_tickets[tile0 + t]);
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            finish[nt] = ticket[nt] == (unsigned)(KS - 1);
            if (finish[nt]) {
                if (t == 0) reset_ticket(&g.sk_tickets[tile0 + nt]);
                if (t < MT * 64) {
                    const float* p0 = g.sk_part + (size_t)(tile0 + nt) * 512 + t * 4;
                    v[nt] = {0.f, 0.f, 0.f, 0.f};
                    for (int q = 0; q < KS; ++q) {                 // split order
                        const float* pq = p0 + (size_t)q * ntile * 512;
                        v[nt].x += ld_agent(pq); v[nt].y += ld_agent(pq + 1); v[nt].z += ld_agent(pq + 2); v[nt].w += ld_agent(pq + 3);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (finish[nt] && t < MT * 64 && m < g.M) {
            const int col = (nb0 + nt) * 16 + 4 * c4;
            f32x4 o = v[nt];
            if (g.bias) o += *reinterpret_cast<const f32x4*>(g.bias + col);
            if (col < g.k1_cols) o += *reinterpret_cast<const f32x4*>(g.A2 + (int64_t)m * g.lda2 + col);
            if (MODE == 1 && col >= g.relu_begin) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (MODE == 2) o = *reinterpret_cast<const f32x4*>(g.res + (int64_t)m * g.ldres + col) + g.alpha * o;
            if (MODE == 3 && (g.scale_cols == 0 || col < g.scale_cols)) o = g.alpha * o;
            if (g.Yt && col >= g.vt_begin) {
                const int b = m / g.U, u = m - b * g.U;
                float* yt = g.Yt + (int64_t)b * g.vt_bs + (int64_t)(col - g.vt_begin) * g.ldt + u;
                yt[0] = o.x;
                yt[g.ldt] = o.y;
                yt[2 * g.ldt] = o.z;
                yt[3 * g.ldt] = o.w;
            } else
                *reinterpret_cast<f32x4*>(g.Y + (int64_t)m * g.ldy + col) = o;
        }
    }
}

template <int MODE, int MT, int NW, int NT>
__global__ __launch_bounds__(64 * NW) void skinny_gemm_wide_multi_kernel(const ArgPack<SkinnyArgs> p, int KS) {
    const SkinnyArgs& g = p.a[blockIdx.z / KS];
    const unsigned gy = (unsigned)((g.M + 31) / 32);
    if (blockIdx.y >= gy) return;
    skinny_gemm_wide_body<MODE, MT, NW, NT>(g, Blk{blockIdx.x, blockIdx.y, blockIdx.z % KS, gridDim.x, gy});
}

bool skinny_gemm_applicable(const SkinnyArgs& g) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (g.ksplit > 1 && (!g.sk_part || !g.sk_tickets || g.K % (64 * g.ksplit) != 0)) return false;
    return g.M >= 1 && g.M <= 512 && g.N % 16 == 0 && g.K % 64 == 0 && g.lda % 4 == 0 && g.ldw % 4 == 0 && g.ldy % 4 == 0 &&
           al16(g.A) && al16(g.W) && al16(g.Y) && (!g.bias || al16(g.bias)) && (!g.res || (al16(g.res) && g.ldres % 4 == 0)) &&
           (!g.Yt || g.vt_begin % 16 == 0) &&
           // two column segments of A: the boundary must fall between the K slices of every form (a wave's slice is K / (NW ksplit), NW <= 16)
           (!g.A2 || (al16(g.A2) && g.lda2 % 4 == 0 && g.K1 > 0 && g.K1 < g.K && g.K1 % 256 == 0 && (g.K - g.K1) % 256 == 0 && g.ksplit <= 1)) &&
           g.relu_begin % 16 == 0 && g.k1_cols % 64 == 0 && (g.k1_cols == 0 || (g.A2 && g.k1_cols <= g.N));
}

template <int MODE, int NW>
static void launch_skinny_mt(const SkinnyArgs& g, hipStream_t s) {
    const dim3 grid((unsigned)(g.N / 16), (unsigned)((g.M + 31) / 32), (unsigned)(g.ksplit > 1 ? g.ksplit : 1));
    if (g.M <= 16) hipLaunchKernelGGL((skinny_gemm_kernel<MODE, 1, NW>), grid, dim3(64 * NW), 0, s, g);
    else hipLaunchKernelGGL((skinny_gemm_kernel<MODE, 2, NW>), grid, dim3(64 * NW), 0, s, g);
}

// waves per workgroup: enough that a wave's K slice is at most 128 wide (8 chunks = one batch of loads), up to 16
template <int MODE>
static void launch_skinny_nw(const SkinnyArgs& g, hipStream_t s) {
    const int k = g.K / (g.ksplit > 1 ? g.ksplit : 1);
    if (k >= 2048 && k % 256 == 0) launch_skinny_mt<MODE, 16>(g, s);
    else if (k >= 1024 && k % 128 == 0) launch_skinny_mt<MODE, 8>(g, s);
    else launch_skinny_mt<MODE, 4>(g, s);
}

template <int MODE, int NW>
static void launch_skinny_multi_mt(const ArgPack<SkinnyArgs>& p, int mmax, int KS, hipStream_t s) {
    const dim3 grid((unsigned)(p.a[0].N / 16), (unsigned)((mmax + 31) / 32), (unsigned)(KS * p.n));
    // (one M tile per workgroup only when EVERY problem has at most 16 rows; a problem of <= 16 rows on the two-tile form computes
    //  the same chains for its rows and drops the clamped duplicates)
    if (mmax <= 16) hipLaunchKernelGGL((skinny_gemm_multi_kernel<MODE, 1, NW>), grid, dim3(64 * NW), 0, s, p, KS);
    else hipLaunchKernelGGL((skinny_gemm_multi_kernel<MODE, 2, NW>), grid, dim3(64 * NW), 0, s, p, KS);
}

template <int MODE>
static void launch_skinny_multi_nw(const ArgPack<SkinnyArgs>& p, int mmax, int KS, hipStream_t s) {
    const int k = p.a[0].K / KS;
    // many row tiles (the narrow form's workgroups would queue up two or more deep on every CU, and the wide form still puts one
    // on three of four CUs): the wide form, 4 column blocks per workgroup with the rows' fragments resident (bit-identical
    // results); needs a K slice of exactly 128 per wave
    constexpr int NT = 4;
    int row_tiles = 0;
    for (int i = 0; i < p.n; ++i) row_tiles += (p.a[i].M + 31) / 32;
    const int wide_opt = opt(OPT_DECODE_WIDE_GEMM);            // 0 = by size, 1 = never, 2 = always
    if (k == 512 && p.a[0].N % (16 * NT) == 0 && mmax > 16 && wide_opt != 1 &&
        (wide_opt == 2 || ((int64_t)(p.a[0].N / 16) * row_tiles * KS >= 2 * (int64_t)device_cus() &&
                           (int64_t)(p.a[0].N / (16 * NT)) * row_tiles * KS * 4 >= 3 * (int64_t)device_cus()))) {
        const dim3 grid((unsigned)(p.a[0].N / (16 * NT)), (unsigned)((mmax + 31) / 32), (unsigned)(KS * p.n));
        hipLaunchKernelGGL((skinny_gemm_wide_multi_kernel<MODE, 2, 4, NT>), grid, dim3(256), 0, s, p, KS);
        return;
    }
    // the folded decoder layers' K = 1024 (eight waves of 128): the same form with two column blocks per workgroup (its partial
    // tiles fill the 64 KB of static LDS at four)
    constexpr int NT8 = 2;
    if (k == 1024 && KS == 1 && p.a[0].N % (16 * NT8) == 0 && mmax > 16 && wide_opt != 1 &&
        (wide_opt == 2 || ((int64_t)(p.a[0].N / 16) * row_tiles >= 2 * (int64_t)device_cus() &&
                           (int64_t)(p.a[0].N / (16 * NT8)) * row_tiles * 4 >= 3 * (int64_t)device_cus()))) {
        const dim3 grid((unsigned)(p.a[0].N / (16 * NT8)), (unsigned)((mmax + 31) / 32), (unsigned)p.n);
        hipLaunchKernelGGL((skinny_gemm_wide_multi_kernel<MODE, 2, 8, NT8>), grid, dim3(512), 0, s, p, KS);
        return;
    }
    if (k >= 2048 && k % 256 == 0) launch_skinny_multi_mt<MODE, 16>(p, mmax, KS, s);
    else if (k >= 1024 && k % 128 == 0) launch_skinny_multi_mt<MODE, 8>(p, mmax, KS, s);
    else launch_skinny_multi_mt<MODE, 4>(p, mmax, KS, s);
}

// G problems with the same W shape (N, K, ldw) and K split in one launch (see Blk)
int launch_skinny_gemm_multi(const SkinnyArgs* g, int G, int mode, hipStream_t s) {
    TAL_CHECK_ARG(g && G >= 1 && G <= TAL_GROUP_MAX, "skinny gemm (multi): %d problems", G);
    TAL_CHECK_ARG(mode >= 0 && mode <= 3, "skinny gemm (multi): mode %d", mode);
    ArgPack<SkinnyArgs> p;
    p.n = G;
    int mmax = 0;
    double work = 0.0;
    const int KS = g[0].ksplit > 1 ? g[0].ksplit : 1;
    for (int i = 0; i < G; ++i) {
        TAL_CHECK_ARG(skinny_gemm_applicable(g[i]) && (mode != 2 || g[i].res), "skinny gemm (multi): problem %d (M=%d N=%d K=%d) not supported", i, g[i].M, g[i].N, g[i].K);
        TAL_CHECK_ARG(g[i].N == g[0].N && g[i].K == g[0].K && (g[i].ksplit > 1 ? g[i].ksplit : 1) == KS,
                      "skinny gemm (multi): problems must share N, K and the K split");
        p.a[i] = g[i];
        mmax = g[i].M > mmax ? g[i].M : mmax;
        work += 2.0 * g[i].M * (double)g[i].N * g[i].K;
    }
    ProfScope prof(PROF_GEMM, work, s);
    switch (mode) {
        case 0: launch_skinny_multi_nw<0>(p, mmax, KS, s); break;
        case 1: launch_skinny_multi_nw<1>(p, mmax, KS, s); break;
        case 2: launch_skinny_multi_nw<2>(p, mmax, KS, s); break;
        default: launch_skinny_multi_nw<3>(p, mmax, KS, s); break;
    }
    TAL_CHECK_LAUNCH("skinny gemm (multi)");
    return TAL_OK;
}

int launch_skinny_gemm(const SkinnyArgs& g, int mode, hipStream_t s) {
    TAL_CHECK_ARG(skinny_gemm_applicable(g), "skinny gemm: shape M=%d N=%d K=%d not supported", g.M, g.N, g.K);
    TAL_CHECK_ARG(mode >= 0 && mode <= 3 && (mode != 2 || g.res), "skinny gemm: mode %d", mode);
    ProfScope prof(PROF_GEMM, 2.0 * g.M * (double)g.N * g.K, s);
    switch (mode) {
        case 0: launch_skinny_nw<0>(g, s); break;
        case 1: launch_skinny_nw<1>(g, s); break;
        case 2: launch_skinny_nw<2>(g, s); break;
        default: launch_skinny_nw<3>(g, s); break;
    }
    TAL_CHECK_LAUNCH("skinny gemm");
    return TAL_OK;
}

template <int HD>
__global__ __launch_bounds__(64 * ATT_NW) void attn_small_kernel(const AttnArgs g) {
    attn_small_body<HD>(g, Blk{blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y});
}

// multi form: grid (max query blocks, heads, problem); every problem is one batch item
template <int HD>
__global__ __launch_bounds__(64 * ATT_NW) void attn_small_multi_kernel(const ArgPack<AttnArgs> p) {
    const AttnArgs& g = p.a[blockIdx.z];
    const unsigned gx = (unsigned)((g.U + 15) / 16);
    if (blockIdx.x >= gx) return;
    attn_small_body<HD>(g, Blk{blockIdx.x, blockIdx.y, 0u, gx, gridDim.y});
}

bool attn_small_applicable(int U, int S, int hd) { return (hd == 128 || hd == 64 || hd == 32 || hd == 16) && S >= 1 && S <= 960 && U >= 1; }

int launch_attn_small(const AttnArgs& g, int B, int hd, hipStream_t s) {
    TAL_CHECK_ARG(attn_small_applicable(g.U, g.S, hd), "attn_small: U=%d S=%d hd=%d not supported", g.U, g.S, hd);
    TAL_CHECK_ARG(g.ldq % 4 == 0 && g.ldk % 4 == 0 && g.ldvt % 4 == 0 && g.ldvt >= g.S, "attn_small: row pitches must be multiples of 4");
    const int S16 = (g.S + 15) & ~15;
    const size_t lds = (size_t)16 * (S16 + 4) * sizeof(float);
    const dim3 grid((unsigned)((g.U + 15) / 16), (unsigned)g.H, (unsigned)B);
    ProfScope prof(PROF_OTHER, (double)B * g.H * g.U * g.S * 8.0, s);
    switch (hd) {
        case 128: hipLaunchKernelGGL((attn_small_kernel<128>), grid, dim3(64 * ATT_NW), lds, s, g); break;
        case 64: hipLaunchKernelGGL((attn_small_kernel<64>), grid, dim3(64 * ATT_NW), lds, s, g); break;
        case 32: hipLaunchKernelGGL((attn_small_kernel<32>), grid, dim3(64 * ATT_NW), lds, s, g); break;
        default: hipLaunchKernelGGL((attn_small_kernel<16>), grid, dim3(64 * ATT_NW), lds, s, g); break;
    }
    TAL_CHECK_LAUNCH("attn_small");
    return TAL_OK;
}

int launch_attn_small_multi(const AttnArgs* g, int G, int hd, hipStream_t s) {
    TAL_CHECK_ARG(g && G >= 1 && G <= TAL_GROUP_MAX, "attn_small (multi): %d problems", G);
    ArgPack<AttnArgs> p;
    p.n = G;
    int umax = 0, smax = 0;
    double work = 0.0;
    for (int i = 0; i < G; ++i) {
        TAL_CHECK_ARG(attn_small_applicable(g[i].U, g[i].S, hd) && g[i].H == g[0].H, "attn_small (multi): problem %d (U=%d S=%d hd=%d) not supported", i, g[i].U, g[i].S, hd);
        TAL_CHECK_ARG(g[i].ldq % 4 == 0 && g[i].ldk % 4 == 0 && g[i].ldvt % 4 == 0 && g[i].ldvt >= g[i].S, "attn_small (multi): row pitches must be multiples of 4");
        p.a[i] = g[i];
        umax = g[i].U > umax ? g[i].U : umax;
        smax = g[i].S > smax ? g[i].S : smax;
        work += (double)g[i].H * g[i].U * g[i].S * 8.0;
    }
    const size_t lds = (size_t)16 * (((smax + 15) & ~15) + 4) * sizeof(float);
    const dim3 grid((unsigned)((umax + 15) / 16), (unsigned)g[0].H, (unsigned)G);
    ProfScope prof(PROF_OTHER, work, s);
    switch (hd) {
        case 128: hipLaunchKernelGGL((attn_small_multi_kernel<128>), grid, dim3(64 * ATT_NW), lds, s, p); break;
        case 64: hipLaunchKernelGGL((attn_small_multi_kernel<64>), grid, dim3(64 * ATT_NW), lds, s, p); break;
        case 32: hipLaunchKernelGGL((attn_small_multi_kernel<32>), grid, dim3(64 * ATT_NW), lds, s, p); break;
        default: hipLaunchKernelGGL((attn_small_multi_kernel<16>), grid, dim3(64 * ATT_NW), lds, s, p); break;
    }
    TAL_CHECK_LAUNCH("attn_small (multi)");
    return TAL_OK;
}

template <int HD>
__global__ __launch_bounds__(256) void attn_split_kernel(const AttnArgs g, int CB, int NCH, float* __restrict__ scratch,
                                                        unsigned* __restrict__ tickets) {
    attn_split_body<HD>(g, CB, NCH, scratch, tickets, Blk{blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y});
}

// multi form: grid (max query blocks, heads, problem x SPLIT_NCH chunk slots); every problem is one batch item with its own
// scratch and tickets; its key-chunk geometry (CB, NCH) follows from its S as in launch_attn_split
struct AttnSplitAux {
    float* scratch[TAL_GROUP_MAX];
    unsigned* tickets[TAL_GROUP_MAX];
};
template <int HD>
__global__ __launch_bounds__(256) void attn_split_multi_kernel(const ArgPack<AttnArgs> p, const AttnSplitAux aux) {
    const unsigned i = blockIdx.z / SPLIT_NCH, ch = blockIdx.z % SPLIT_NCH;
    const AttnArgs& g = p.a[i];
    const int CB = split_cb(g.S), nblk = (g.S + 15) / 16, NCH = (nblk + CB - 1) / CB;
    const unsigned gx = (unsigned)((g.U + 15) / 16);
    if (blockIdx.x >= gx || (int)ch >= NCH) return;
    attn_split_body<HD>(g, CB, NCH, aux.scratch[i], aux.tickets[i], Blk{blockIdx.x, blockIdx.y, ch, gx, gridDim.y});
}

size_t attn_split_scratch_floats(int B, int U, int S, int H, int hd) {
    return (size_t)B * H * ((U + 15) / 16) * SPLIT_NCH * split_record_floats(hd, split_cb(S));
}
int attn_split_tickets(int B, int U, int H) { return B * H * ((U + 15) / 16); }

// key-split form; scratch: attn_split_scratch_floats floats; tickets: attn_split_tickets words, zero before the first launch
int launch_attn_split(const AttnArgs& g, int B, int hd, float* scratch, unsigned* tickets, hipStream_t s) {
    TAL_CHECK_ARG(attn_small_applicable(g.U, g.S, hd) && scratch && tickets, "attn_split: U=%d S=%d hd=%d not supported", g.U, g.S, hd);
    TAL_CHECK_ARG(g.ldq % 4 == 0 && g.ldk % 4 == 0 && g.ldvt % 4 == 0 && g.ldvt >= g.S, "attn_split: row pitches must be multiples of 4");
    const int CB = split_cb(g.S);
    const int nblk = (g.S + 15) / 16;
    const int NCH = (nblk + CB - 1) / CB;
    const size_t lds = (size_t)16 * (CB * 16 + 4) * sizeof(float);
    const dim3 grid((unsigned)((g.U + 15) / 16), (unsigned)g.H, (unsigned)(B * NCH));
    ProfScope prof(PROF_OTHER, (double)B * g.H * g.U * g.S * 8.0, s);
    switch (hd) {
        case 128: hipLaunchKernelGGL((attn_split_kernel<128>), grid, dim3(256), lds, s, g, CB, NCH, scratch, tickets); break;
        case 64: hipLaunchKernelGGL((attn_split_kernel<64>), grid, dim3(256), lds, s, g, CB, NCH, scratch, tickets); break;
        case 32: hipLaunchKernelGGL((attn_split_kernel<32>), grid, dim3(256), lds, s, g, CB, NCH, scratch, tickets); break;
        default: hipLaunchKernelGGL((attn_split_kernel<16>), grid, dim3(256), lds, s, g, CB, NCH, scratch, tickets); break;
    }
    TAL_CHECK_LAUNCH("attn_split");
    return TAL_OK;
}

int launch_attn_split_multi(const AttnArgs* g, float* const* scratch, unsigned* const* tickets, int G, int hd, hipStream_t s) {
    TAL_CHECK_ARG(g && scratch && tickets && G >= 1 && G <= TAL_GROUP_MAX, "attn_split (multi): %d problems", G);
    ArgPack<AttnArgs> p;
    AttnSplitAux aux;
    p.n = G;
    int umax = 0, cbmax = 0;
    double work = 0.0;
    for (int i = 0; i < G; ++i) {
        TAL_CHECK_ARG(attn_small_applicable(g[i].U, g[i].S, hd) && scratch[i] && tickets[i] && g[i].H == g[0].H, "attn_split (multi): problem %d (U=%d S=%d hd=%d) not supported", i, g[i].U, g[i].S, hd);
        TAL_CHECK_ARG(g[i].ldq % 4 == 0 && g[i].ldk % 4 == 0 && g[i].ldvt % 4 == 0 && g[i].ldvt >= g[i].S, "attn_split (multi): row pitches must be multiples of 4");
        p.a[i] = g[i];
        aux.scratch[i] = scratch[i];
        aux.tickets[i] = tickets[i];
        umax = g[i].U > umax ? g[i].U : umax;
        const int cb = split_cb(g[i].S);
        cbmax = cb > cbmax ? cb : cbmax;
        work += (double)g[i].H * g[i].U * g[i].S * 8.0;
    }
    const size_t lds = (size_t)16 * (cbmax * 16 + 4) * sizeof(float);
    const dim3 grid((unsigned)((umax + 15) / 16), (unsigned)g[0].H, (unsigned)(G * SPLIT_NCH));
    ProfScope prof(PROF_OTHER, work, s);
    switch (hd) {
        case 128: hipLaunchKernelGGL((attn_split_multi_kernel<128>), grid, dim3(256), lds, s, p, aux); break;
        case 64: hipLaunchKernelGGL((attn_split_multi_kernel<64>), grid, dim3(256), lds, s, p, aux); break;
        case 32: hipLaunchKernelGGL((attn_split_multi_kernel<32>), grid, dim3(256), lds, s, p, aux); break;
        default: hipLaunchKernelGGL((attn_split_multi_kernel<16>), grid, dim3(256), lds, s, p, aux); break;
    }
    TAL_CHECK_LAUNCH("attn_split (multi)");
    return TAL_OK;
}

// avg[b][u][s] = mean over heads of probs[b][h][u][s], summed in head order (MultiheadAttention's averaged weights)
__global__ __launch_bounds__(256) void head_average_kernel(const float* __restrict__ probs, float* __restrict__ avg, int H,
                                                          int64_t US, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / US, r = i - b * US;
    const float* p = probs + b * H * US + r;
    float a = 0.f;
    for (int h = 0; h < H; ++h) a += p[(int64_t)h * US];
    avg[i] = a * (1.0f / (float)H);
}

int launch_head_average(const float* probs, float* avg, int B, int H, int U, int S, hipStream_t s) {
    const int64_t total = (int64_t)B * U * S;
    hipLaunchKernelGGL(head_average_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, probs, avg, H, (int64_t)U * S, total);
    TAL_CHECK_LAUNCH("head_average");
    return TAL_OK;
}

}  // namespace tal
