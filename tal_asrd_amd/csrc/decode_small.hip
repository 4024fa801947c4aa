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
#include "common.h"

namespace tal {

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

template <int MODE, int MT, int NW>
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
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = m0 + mt * 16 + r16;
        ap[mt] = g.A + (int64_t)(row < g.M ? row : g.M - 1) * g.lda + kofs + 4 * kq;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = {0.f, 0.f, 0.f, 0.f};
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
    if (MODE == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (MODE == 2) v = *reinterpret_cast<const f32x4*>(g.res + (int64_t)m * g.ldres + col) + g.alpha * v;
    if (MODE == 3 && (g.scale_cols == 0 || col < g.scale_cols)) v = g.alpha * v;
    if (g.Yt && col >= g.vt_begin) {
        const int b = m / g.U, u = m - b * g.U;
        float* yt = g.Yt + (int64_t)b * g.vt_bs + (int64_t)(col - g.vt_begin) * g.ldt + u;
        yt[0] = v.x;
        yt[g.ldt] = v.y;
        yt[2 * g.ldt] = v.z;
        yt[3 * g.ldt] = v.w;
        return;
    }
    *reinterpret_cast<f32x4*>(g.Y + (int64_t)m * g.ldy + col) = v;
}

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
    __shared__ __attribute__((aligned(16))) float part[NW * MT * 256];   // [wave][m tile][row 16][col 16]
    __shared__ unsigned ticket;
    constexpr int UNR = 8;                          // Kw == 128: the wave's K slice is exactly one batch of 8 chunks
    const int m0 = blk.y * 32;
    const int lane = threadIdx.x & 63, w = wave_id();
    const int r16 = lane & 15, kq = lane >> 4;
    const int KS = g.ksplit > 1 ? g.ksplit : 1;
    const int Kw = g.K / (NW * KS);
    const int kofs = ((int)blk.z * NW + w) * Kw;
    f32x4 av[MT][UNR], bw[2][UNR];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = m0 + mt * 16 + r16;
        const float* ap = g.A + (int64_t)(row < g.M ? row : g.M - 1) * g.lda + kofs + 4 * kq;
#pragma unroll
        for (int u = 0; u < UNR; ++u) av[mt][u] = *reinterpret_cast<const f32x4*>(ap + u * 16);
    }
    const int nb0 = blk.x * NT;                     // first 16-column block
    auto load_w = [&](int nb, int buf) {
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
        if (nt > 0) __syncthreads();                // the previous block's sums have been read out of `part`
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[((w * MT + mt) * 16 + 4 * kq + i) * 16 + r16] = acc[mt][i];
        __syncthreads();
        const int n0 = (nb0 + nt) * 16;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < MT * 64) {
            v = *reinterpret_cast<const f32x4*>(&part[((0 * MT + mt_t) * 16 + r) * 16 + 4 * c4]);
#pragma unroll
            for (int q = 1; q < NW; ++q) v += *reinterpret_cast<const f32x4*>(&part[((q * MT + mt_t) * 16 + r) * 16 + 4 * c4]);   // wave order
        }
        bool finish = true;
        if (KS > 1) {
            const unsigned gx = blk.gx * NT;        // the narrow form's grid extent: same tile numbers, partial buffers and tickets
            const unsigned tile = blk.y * gx + (unsigned)(nb0 + nt), ntile = gx * blk.gy;
            float* mine = g.sk_part + ((size_t)blk.z * ntile + tile) * 512 + t * 4;
            if (t < MT * 64) { st_agent(mine, v.x); st_agent(mine + 1, v.y); st_agent(mine + 2, v.z); st_agent(mine + 3, v.w); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) ticket = take_ticket(&g.sk_tickets[tile]);
            __syncthreads();
            finish = ticket == (unsigned)(KS - 1);
            if (finish) {
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
        }
        if (finish && t < MT * 64 && m < g.M) {
            const int col = n0 + 4 * c4;
            if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
            if (MODE == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (MODE == 2) v = *reinterpret_cast<const f32x4*>(g.res + (int64_t)m * g.ldres + col) + g.alpha * v;
            if (MODE == 3 && (g.scale_cols == 0 || col < g.scale_cols)) v = g.alpha * v;
            if (g.Yt && col >= g.vt_begin) {
                const int b = m / g.U, u = m - b * g.U;
                float* yt = g.Yt + (int64_t)b * g.vt_bs + (int64_t)(col - g.vt_begin) * g.ldt + u;
                yt[0] = v.x;
                yt[g.ldt] = v.y;
                yt[2 * g.ldt] = v.z;
                yt[3 * g.ldt] = v.w;
            } else
                *reinterpret_cast<f32x4*>(g.Y + (int64_t)m * g.ldy + col) = v;
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
           (!g.Yt || g.vt_begin % 16 == 0);
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

// ---------------------------------------------------------------------------------------------------------------
// 8 waves per workgroup.  Scores: wave w takes key blocks w, w + 8, ... three at a time (all 3 x HD/16 fragment loads in
// flight before the first MFMA).  P.V: one 16-feature block per wave (HD = 128), the V^T fragments of the next 8 key
// steps in flight while the current 8 are multiplied.
constexpr int ATT_NW = 8;
template <int HD>
__device__ __forceinline__ void attn_small_body(const AttnArgs& g, const Blk blk) {
    extern __shared__ __attribute__((aligned(16))) float sc[];   // [16][SP] scores -> probabilities
    const int u0 = blk.x * 16, h = blk.y, b = blk.z;
    const int lane = threadIdx.x & 63, w = wave_id();
    const int r16 = lane & 15, kq = lane >> 4;
    const int U = g.U, S = g.S;
    const int S16 = (S + 15) & ~15, SP = S16 + 4;
    constexpr int NC = HD / 16;
    // The P.V operand (V^T) does not depend on the scores: its first fragments are requested before anything else, so
    // their round trip runs under the score / softmax phases.
    constexpr int VU = 8;
    const int nstep = S16 / 16;
    const int S4 = (int)g.ldvt;
    const int cb = w < NC ? w : NC - 1;
    const float* vrow = g.vt + (int64_t)b * g.vt_bs + (int64_t)(h * HD + cb * 16 + r16) * g.ldvt + 4 * kq;
    f32x4 vb[3][VU];
    auto load_v = [&](int t0, int buf) {
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
    load_v(0, 0);
    load_v(VU, 1);
    load_v(2 * VU, 2);
    // ---- scores
    {
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
    // ---- row softmax (max, exp, sum, divide, as torch): wave w owns rows 2 w and 2 w + 1, one per 32-lane half, so the two rows'
    // cross-lane reductions (a shuffle is ~100 cycles of latency) run side by side
    {
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
                if (pr && s < S) pr[s] = v;
            }
        }
    }
    __syncthreads();
    // ---- ctx = P . V: feature block cb = w (waves >= HD / 16 are done); batches 0..2 are already in registers
    if (w >= NC) return;
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
        if (t0 + 3 * VU < nstep) load_v(t0 + 3 * VU, 0);
        if (t0 + VU < nstep) mfma_v(t0 + VU, 1);
        if (t0 + 4 * VU < nstep) load_v(t0 + 4 * VU, 1);
        if (t0 + 2 * VU < nstep) mfma_v(t0 + 2 * VU, 2);
        if (t0 + 5 * VU < nstep) load_v(t0 + 5 * VU, 2);
    }
    const int col = h * HD + cb * 16 + r16;
    const float bv = g.vbias ? g.vbias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = u0 + 4 * kq + i;
        if (u < U) g.ctx[(int64_t)b * g.c_bs + (int64_t)u * g.ldc + col] = acc[i] + bv;
    }
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

// ---------------------------------------------------------------------------------------------------------------
// The same attention with the KEY axis cut over workgroups (cross-attention of a decode step: 357 keys x 512 features of
// K and V^T are 1.4 MB, and the 8 workgroups of the kernel above each pull 367 KB through one CU at 25-60 GB/s).
// Workgroup (row block, head, batch item x chunk) handles CB key blocks: scores, chunk-local max m_c, p = exp(s - m_c),
// l_c = sum p, o_c = p . V_c, all written to a scratch record; the last workgroup to arrive at (row block, head, item)
// -- ticket on a zero-initialised, self-resetting counter -- merges the chunks in chunk order,
//     M = max m_c,  w_c = exp(m_c - M),  ctx = sum_c w_c o_c / sum_c w_c l_c (+ v bias),
// and, for the rows >= prob_row0, the per-head probabilities p w_c / L.  Deterministic (fixed merge order).
constexpr int SPLIT_MAX_CB = 8;      // key blocks per chunk (S <= 960 -> at most 8 chunks of 8 blocks)
constexpr int SPLIT_NCH = 8;
__host__ __device__ static inline int split_cb(int S) { const int nblk = (S + 15) / 16; return (nblk + SPLIT_NCH - 1) / SPLIT_NCH; }
__host__ __device__ static inline size_t split_record_floats(int hd, int cb) { return (size_t)16 * hd + 32 + (size_t)16 * 16 * cb; }

template <int HD>
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
            *reinterpret_cast<f32x4*>(g.ctx + (int64_t)b * g.c_bs + (int64_t)u * g.ldc + h * HD + col) = r;
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
            g.probs[(((int64_t)b * g.H + h) * (U - g.prob_row0) + (u - g.prob_row0)) * S + key] = pc * wsh[row * SPLIT_NCH + c] / Lsh[row];
        }
    }
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
