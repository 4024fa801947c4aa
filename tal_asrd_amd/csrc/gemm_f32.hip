// fp32 dense layer on the CDNA4 matrix cores: Y = epilogue(X . W^T + b), optionally batched.
//
// Covers nn.Linear and the 1x1 Conv1d pair of TDSBlock (tal/asr/models.py:312-318,
// 86.9 % of the encoder's MACs), the encoder projections (:100,131), the SD / speaker
// heads (:421-422,143-146), the decoder's projections / FFN (:493-499) and -- through the
// batched form with per-(batch, head) strides -- the QK^T and PV contractions of
// nn.MultiheadAttention (:514-518).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bit-for-bit an fmaf chain),
// so results stay inside the 1e-3 fp32 logit tolerance of BASELINE.json without any
// reduced-precision trick.  Roofline: 157.3 TFLOP/s (fp32 matrix peak of MI355X; a pure-MFMA
// loop reaches 155 on this chip, scripts/ubench/mfma_peak.hip).
//
// Kernels, all 4 waves x (32 rows x 32*NSUB columns) of 32x32 accumulators:
//   gemm_glds_kernel    128(M) x 160(N) x 32(K), the hot one (M > 512, K % 32 == 0).  Operand
//       tiles go HBM/L2 -> LDS directly (LDS-DMA `buffer_load_dwordx4 ... lds`: SGPR descriptor,
//       32-bit lane offset, no VGPR staging, no ds_write), double-buffered, ONE barrier per K step;
//       the LDS image is lane-linear, so bank conflicts are removed by an XOR swizzle applied to the
//       per-lane SOURCE offset and again on the fragment read.  160 divides every TDS width
//       (800/1120/1440): no N-tail waste.  The tiles of the last partial scheduling round can be cut
//       along K (SPLITK launches + gemm_splitk_fixup_kernel) when the caller provides scratch.
//   gemm_splitk4_kernel 32 x 32 tile, 4 waves split K inside the workgroup: the decoder's short,
//       latency-bound problems (M <= 512 rows).
//   gemm_nt_f32_kernel<.,4,5>  128 x 160, register-staged, single LDS buffer: K tails
//       (K % 32 != 0; only small test models);  <.,1,1>  32 x 128: small M with mode 4 / odd shapes.
// Epilogue (shared): the wave's tile is staged through LDS 16 rows at a time and written as whole
// rows with 16-byte buffer stores whose descriptor ends at the last valid row; bias / ReLU /
// ReZero-residual / scale are fused; all bias and residual loads of a half tile are issued before
// any of them is consumed.
// Cost model (measured, DESIGN.md sections 3 and 8): an fp32 MFMA does not issue beside another wave's vector ALU work, and
// an fp16 MFMA only beside plain (not packed) fp32 instructions, so every VALU / vector-memory instruction here is matrix
// time lost -- hence SGPR descriptors, 32-bit offsets and scalar address arithmetic wherever possible.
// The fp16x3 form of the TDS pointwise layers has two more homes: gemm_w64.hip (256 x 160 tiles, one wave per SIMD: long
// inputs) and gemm_s64.hip (64 x 80 tiles: short inputs); launch_gemm below picks by the number of rows.
#include "gemm_common.h"

namespace tal {

// ---------------------------------------------------------------------------------------------
// hot kernel: direct-to-LDS operand loads, double buffer, one barrier per K step
// ---------------------------------------------------------------------------------------------
// (the fp16x3 form carries two accumulator sets: it is held to 256 registers = 2 workgroups per CU explicitly)
template <int MODE, int NSUB, bool SPLITK, int STAGE, bool F16X3 = false, int EPI = 0>
__global__ __launch_bounds__(256, F16X3 ? 2 : 1) void gemm_glds_kernel(const GemmArgs g) {
    // Set-up and epilogue are short VALU / memory sequences; the co-resident workgroup is usually deep in
    // its MFMA loop, and at equal priority every one of these instructions queues behind a 64-cycle MFMA.
    __builtin_amdgcn_s_setprio(3);
    constexpr int BM = 128, BN = 32 * NSUB;
    constexpr int ROWS = BM + BN;                 // 288 operand rows of 32 floats (128 B) per K step
    constexpr int CHUNKS = ROWS / 8;              // 36 wave-loads of 1 KB (8 rows) each
    constexpr int PER_WAVE = CHUNKS / 4;          // 9: t < 4 -> A rows, t >= 4 -> W rows
    __shared__ __attribute__((aligned(16))) float lds[2 * ROWS * 32];  // 73,728 B -> 2 workgroups / CU

    // XCD-remapped tile index.  SPLITK launches carry the whole-round tiles (blocks < tile_base) AND, behind
    // them in dispatch order, the K slices of the last partial round (block -> (tile, K slice)), so the
    // slices fill slots as the whole tiles drain instead of waiting for a kernel boundary.
    const bool is_slice = SPLITK && blockIdx.x >= (unsigned)g.tile_base;
    const unsigned sbid = blockIdx.x - (unsigned)g.tile_base;
    const int slice = is_slice ? (int)(sbid / (unsigned)g.tail_tiles) : 0;
    const unsigned logical = is_slice ? (unsigned)g.tile_base + sbid % (unsigned)g.tail_tiles
                                      : (SPLITK ? logical_tile_of((unsigned)g.tile_base, blockIdx.x) : logical_tile());
    // (tile row by a multiply-high with the launch's magic number: a run-time division is ~40 vector instructions)
    const unsigned tile_m = g.tiles_n == 1 ? logical : __umulhi(logical, g.tiles_n_magic);
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = (int)(logical - tile_m * (unsigned)g.tiles_n) * BN;
    const int64_t M = g.M;
    const int N = g.N, K = g.K;
    const int z1 = EPI ? 0 : (int)blockIdx.y / g.nb2, z2 = EPI ? 0 : (int)blockIdx.y % g.nb2;    // (EPI: never batched)
    const float* A = g.A + z1 * g.a_s1 + z2 * g.a_s2;
    const float* W = g.W + z1 * g.w_s1 + z2 * g.w_s2;
    float* Y = g.Y + z1 * g.y_s1 + z2 * g.y_s2;
    const float* bias = g.bias ? g.bias + z2 * g.bias_s2 : nullptr;
    const float* res = g.res ? g.res + z1 * g.r_s1 + z2 * g.r_s2 : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = wave_id();
    const int wm = w, wn = 0;

    // Wave w loads chunks i = w + 4t.  A chunk is 8 rows x 128 B; lane l lands at LDS byte
    // i*1024 + l*16 = row r = 8i + l/8, 16-byte slot l%8.  Slot s of row r holds the row's logical
    // 16-byte column s ^ ((r >> 1) & 7): the swizzle is applied to the global source address, the LDS
    // destination stays linear as the instruction requires.  gfx950 serves a ds_read_b128 sixteen
    // lanes (256 bytes, 64 banks) at a time: with (r >> 1) & 7 sixteen consecutive rows cover all 64
    // banks (SQ_LDS_BANK_CONFLICT = 0); the (r & 7) form, enough for 8-lane phases, measured 50 %
    // conflict cycles.  (r >> 1) & 7 = (4 (i & 1) + (l >> 4)) & 7, and i & 1 = w & 1: a per-lane constant.
    const int sub = lane >> 3;                    // row inside the chunk
    const int srccol = ((lane & 7) ^ (((w & 1) * 4 + (lane >> 4)) & 7)) * 4;    // logical float column this lane fetches
    const int a_rows = (int)((M - m0) < BM ? (M - m0) : BM) - 1;
    const int b_rows = ((N - n0) < BN ? (N - n0) : BN) - 1;
    const float* src[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int row = 8 * (w + 4 * t) + sub;    // 0..287
        if (t < 4)
            src[t] = A + (m0 + min(row, a_rows)) * g.lda + srccol;
        else
            src[t] = W + (int64_t)(n0 + min(row - BM, b_rows)) * g.ldw + srccol;
    }
    // STAGE 2 (default): the LDS-DMA loads use the buffer form -- descriptor in SGPRs, one 32-bit lane
    // offset, the K offset in an SGPR -- instead of 64-bit lane addresses.  On gfx950 every vector-memory
    // instruction costs the SIMD matrix-pipe issue cycles and the global form (plus the 64-bit VALU add
    // per load that advances its address) costs the most: K-loop step 4.77 -> 4.43 us of an ideal 4.27
    // (scripts/ubench/mfma_mix.hip isolates the ingredients; scripts/bench_gemm_fit.py the real kernel).
    // STAGE 0 keeps the global form for leading dimensions whose lane offsets do not fit 32 bits.
    // (64-bit multiplies run on the VALU even for uniform values; the descriptors must sit in SGPRs)
    auto uniform_ptr = [](const float* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    };
    __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(A + m0 * g.lda), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(W + (int64_t)n0 * g.ldw), 0, 0x7fffffff, 0x00020000);
    int voff[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int row = 8 * (w + 4 * t) + sub;
        voff[t] = t < 4 ? (int)((min(row, a_rows) * g.lda + srccol) * 4) : (int)((min(row - BM, b_rows) * g.ldw + srccol) * 4);
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) {
            float* dst = lds + buf * (ROWS * 32) + (w + 4 * t) * 256;
            if (STAGE == 2) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass drops the whole kernel stub when it meets this builtin)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(t < 4 ? rsrc_a : rsrc_w, (__attribute__((address_space(3))) void*)dst, 16,
                                                         voff[t], kt * (BK * 4), 0, 0);
#endif
            } else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + kt * BK),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x16 acc[NSUB];
    f32x16 accx[F16X3 ? NSUB : 1];     // fp16x3: the cross terms hi*lo + lo*hi (scaled by 2^11)
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        // EPI: the accumulators start from the bias (all 16 elements of acc[j] belong to column n0 + 32 j + lane % 32);
        // a K slice starts from zero, its bias is added by the fix-up kernel
        const float b0 = (EPI && bias && !is_slice) ? bias[n0 + j * 32 + (lane & 31)] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = b0;
    }
#pragma unroll
    for (int j = 0; j < (F16X3 ? NSUB : 1); ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accx[j][e] = 0.f;

    // fragment read: row (l & 31) of a 32-row block, logical 16-byte column 2*kk + (l >> 5)
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    const int fhalf = lane >> 5;
    const int nk_all = K / BK;
    const int kt0 = is_slice ? (int)((int64_t)slice * nk_all / g.split) : 0;
    const int nk = is_slice ? (int)((int64_t)(slice + 1) * nk_all / g.split) : nk_all;
    issue(kt0, kt0 & 1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_assume(kt0 < nk);       // (K >= 32: without this the accumulators are initialised twice, 160 moves)
    for (int kt = kt0; kt < nk; ++kt) {
        // tile kt has landed -- this wave's part of it by the explicit wait: the compiler orders an LDS-DMA load only in front of
        // the wave's OWN LDS reads, not in front of a barrier (csrc/head.hip had a barrier with no wait in front, round 5) -- and
        // every wave is done reading the other buffer (it finished step kt-1 before arriving here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const float* As = lds + (kt & 1) * (ROWS * 32) + (wm * 32 + frow) * 32;
        const float* Bs = lds + (kt & 1) * (ROWS * 32) + (BM + frow) * 32;
        if (F16X3) {
            // operands are hi/lo fp16 splits: a 128-byte row of this K block is [32 hi | 32 lo]; one MFMA takes 16 k,
            // lanes < 32 the first 8 and lanes >= 32 the next 8: 16-byte slot 2g + half (hi), 4 + 2g + half (lo).
            // a.w = sum hi*hi + 2^-11 sum (hi*lo + lo*hi), fp32 accumulation; lo*lo (2^-22 relative) is dropped.
            // Flattened (gk, j) stages q = NSUB gk + j: stage q issues the B fragments of stage q + 2 (and the A
            // fragments of the next gk) before its three MFMAs -- an MFMA is only 32 cycles here, and with two
            // waves per SIMD every LDS read needs ~200 cycles of issued matrix work between issue and use.
            f16x8 ahi[2], alo[2], bhi[3], blo[3];
            auto slh = [&](int gk) { return ((2 * gk + fhalf) ^ fsw) * 4; };
            auto sll = [&](int gk) { return ((4 + 2 * gk + fhalf) ^ fsw) * 4; };
            auto read_b = [&](int q, int slot) {
                const int gk = q / NSUB, j = q % NSUB;
                bhi[slot] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * 32 + slh(gk));
                blo[slot] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * 32 + sll(gk));
            };
            ahi[0] = *reinterpret_cast<const f16x8*>(As + slh(0));
            alo[0] = *reinterpret_cast<const f16x8*>(As + sll(0));
            read_b(0, 0);
            read_b(1, 1);
#pragma unroll
            for (int q = 0; q < 2 * NSUB; ++q) {
                const int gk = q / NSUB, j = q % NSUB;
                if (q + 2 < 2 * NSUB) read_b(q + 2, (q + 2) % 3);
                if (q == NSUB - 2) {
                    ahi[1] = *reinterpret_cast<const f16x8*>(As + slh(1));
                    alo[1] = *reinterpret_cast<const f16x8*>(As + sll(1));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[gk], bhi[q % 3], acc[j], 0, 0, 0);
                accx[F16X3 ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[gk], blo[q % 3], accx[F16X3 ? j : 0], 0, 0, 0);
                accx[F16X3 ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[gk], bhi[q % 3], accx[F16X3 ? j : 0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
        f32x4 fa[2], fb[2][NSUB];
        fa[0] = *reinterpret_cast<const f32x4*>(As + ((fhalf) ^ fsw) * 4);
#pragma unroll
        for (int j = 0; j < NSUB; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * 32 + ((fhalf) ^ fsw) * 4);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 8) {
                const int sl = ((2 * (kk + 1) + fhalf) ^ fsw) * 4;
                fa[nxt] = *reinterpret_cast<const f32x4*>(As + sl);
#pragma unroll
                for (int j = 0; j < NSUB; ++j) fb[nxt][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * 32 + sl);
            }
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur][j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur][j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur][j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur][j].w, acc[j], 0, 0, 0);
            }
        }
    }
    // all waves done with the operand buffers before they become the store stage.  (No LDS-DMA load is in flight here: the last
    // step issues none.  The wait says so in the instruction stream, where tests/test_isa_sync.py checks every barrier of this
    // kernel without knowing the trip count; it retires nothing and costs nothing.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);
    if (F16X3) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 s11 = {1.0f / 2048.0f, 1.0f / 2048.0f};
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 x2 = {accx[F16X3 ? j : 0][e], accx[F16X3 ? j : 0][e + 1]}, a2 = {acc[j][e], acc[j][e + 1]};
                const f32x2 o2 = __builtin_elementwise_fma(x2, s11, a2);
                acc[j][e] = o2[0];
                acc[j][e + 1] = o2[1];
            }
    }
    if (is_slice) {
        // raw accumulators of this K slice -> scratch tile [(tile, slice)][128][BN]; the fix-up kernel
        // adds the slices in a fixed order and applies bias / activation / residual
        GemmArgs gp = g;
        gp.M = m0 + BM;
        gp.N = n0 + BN;
        gp.ldy = BN;
        gp.out_split = 0;            // partial sums stay fp32; the fix-up kernel writes the split form
        float* tile_ws = g.splitk_ws + ((size_t)(logical - g.tile_base) * g.split + slice) * (BM * BN);
        gemm_epilogue<0, NSUB>(gp, acc, lds, tile_ws - (m0 * BN + n0), nullptr, nullptr, m0, n0, lane, w, wm, wn);
    } else {
        if constexpr (EPI != 0) gemm_epilogue_split<MODE, NSUB>(g, acc, lds, Y, res, m0, n0, lane, w);
        else gemm_epilogue<MODE, NSUB>(g, acc, lds, Y, bias, res, m0, n0, lane, w, wm, wn);
    }
}

// ---------------------------------------------------------------------------------------------
// register-staged kernels (K tails, and the small tile)
// ---------------------------------------------------------------------------------------------
template <int MODE, int WAVES_M, int NSUB, int BKT>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const GemmArgs g) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int BM = 32 * WAVES_M;
    constexpr int BN = 32 * NSUB * WAVES_N;
    constexpr int LDS_LD = BKT + 4;            // padded pitch: conflict-free ds_read_b128 of 16 rows x 4 floats
    constexpr int C4 = BKT / 4;                // 16-byte columns per K slab
    constexpr int RPP = 256 / C4;              // rows covered by one pass of the 256 threads
    constexpr int A_LOADS = BM / RPP;
    constexpr int B_LOADS = BN / RPP;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_LD];
    float* As = lds;
    float* Bs = lds + BM * LDS_LD;

    const unsigned logical = logical_tile();
    const int64_t m0 = (int64_t)(logical / (unsigned)g.tiles_n) * BM;
    const int n0 = (int)(logical % (unsigned)g.tiles_n) * BN;
    const int64_t M = g.M;
    const int N = g.N, K = g.K;

    // batch z = z1 * nb2 + z2 with independent strides (e.g. z1 = batch item, z2 = attention head)
    const int z1 = (int)blockIdx.y / g.nb2, z2 = (int)blockIdx.y % g.nb2;
    const float* A = g.A + z1 * g.a_s1 + z2 * g.a_s2;
    const float* W = g.W + z1 * g.w_s1 + z2 * g.w_s2;
    float* Y = g.Y + z1 * g.y_s1 + z2 * g.y_s2;
    const float* bias = g.bias ? g.bias + z2 * g.bias_s2 : nullptr;
    const float* res = g.res ? g.res + z1 * g.r_s1 + z2 * g.r_s2 : nullptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = wave_id();
    const int wm = w / WAVES_N, wn = w % WAVES_N;
    const int lrow = tid / C4;  // row inside a pass
    const int lc4 = tid % C4;   // 16-byte column inside the K slab

    // tile-relative offsets (rows past the end are clamped; their results are never stored)
    const float* At = A + m0 * g.lda + lc4 * 4;
    const float* Wt = W + (int64_t)n0 * g.ldw + lc4 * 4;
    const int a_rows = (int)((M - m0) < BM ? (M - m0) : BM) - 1;
    const int b_rows = ((N - n0) < BN ? (N - n0) : BN) - 1;
    int64_t a_off[A_LOADS], b_off[B_LOADS];
#pragma unroll
    for (int p = 0; p < A_LOADS; ++p) a_off[p] = min(lrow + RPP * p, a_rows) * g.lda;
#pragma unroll
    for (int p = 0; p < B_LOADS; ++p) b_off[p] = min(lrow + RPP * p, b_rows) * g.ldw;

    f32x16 acc[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // K tail (K % 32 != 0): out-of-range 16-byte columns are replaced by zeros when the staged
    // registers are written to LDS (not at load time: a select right behind the loads would make
    // every K step wait for its own global loads).  The loads themselves always use a clamped,
    // valid offset.  K % 4 == 0 is required by the caller.
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[A_LOADS], rb[B_LOADS];
    bool in_cur = lc4 * 4 < K;
    {
        const int ko = in_cur ? 0 : -lc4 * 4;
#pragma unroll
        for (int p = 0; p < A_LOADS; ++p) ra[p] = *reinterpret_cast<const f32x4*>(At + ko + a_off[p]);
#pragma unroll
        for (int p = 0; p < B_LOADS; ++p) rb[p] = *reinterpret_cast<const f32x4*>(Wt + ko + b_off[p]);
    }

    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
    const int nk = (K + BKT - 1) / BKT;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int p = 0; p < A_LOADS; ++p)
            *reinterpret_cast<f32x4*>(&As[(lrow + RPP * p) * LDS_LD + lc4 * 4]) = in_cur ? ra[p] : zero4;
#pragma unroll
        for (int p = 0; p < B_LOADS; ++p)
            *reinterpret_cast<f32x4*>(&Bs[(lrow + RPP * p) * LDS_LD + lc4 * 4]) = in_cur ? rb[p] : zero4;
        __syncthreads();
        if (kt + 1 < nk) {
            in_cur = (kt + 1) * BKT + lc4 * 4 < K;
            const int ko = in_cur ? (kt + 1) * BKT : -lc4 * 4;
#pragma unroll
            for (int p = 0; p < A_LOADS; ++p) ra[p] = *reinterpret_cast<const f32x4*>(At + ko + a_off[p]);
#pragma unroll
            for (int p = 0; p < B_LOADS; ++p) rb[p] = *reinterpret_cast<const f32x4*>(Wt + ko + b_off[p]);
        }
        // Fragments of k-slice kk+1 are fetched from LDS (into a second register set) before the
        // MFMAs of slice kk issue.
        f32x4 fa[2], fb[2][NSUB];
        fa[0] = *reinterpret_cast<const f32x4*>(&As[wm * 32 * LDS_LD + frag_off]);
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
            fb[0][j] = *reinterpret_cast<const f32x4*>(&Bs[(wn * NSUB + j) * 32 * LDS_LD + frag_off]);
#pragma unroll
        for (int kk = 0; kk < BKT / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BKT / 8) {
                fa[nxt] = *reinterpret_cast<const f32x4*>(&As[wm * 32 * LDS_LD + frag_off + (kk + 1) * 8]);
#pragma unroll
                for (int j = 0; j < NSUB; ++j)
                    fb[nxt][j] = *reinterpret_cast<const f32x4*>(
                        &Bs[(wn * NSUB + j) * 32 * LDS_LD + frag_off + (kk + 1) * 8]);
            }
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur][j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur][j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur][j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur][j].w, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    gemm_epilogue<MODE, NSUB>(g, acc, lds, Y, bias, res, m0, n0, lane, w, wm, wn);
}

template <int WAVES_M, int NSUB, int BKT>
static void launch_tile(const GemmArgs& g, int mode, dim3 grid, hipStream_t s) {
    switch (mode) {
        case 0: hipLaunchKernelGGL((gemm_nt_f32_kernel<0, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
        case 1: hipLaunchKernelGGL((gemm_nt_f32_kernel<1, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
        case 2: hipLaunchKernelGGL((gemm_nt_f32_kernel<2, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
        case 3: hipLaunchKernelGGL((gemm_nt_f32_kernel<3, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL((gemm_nt_f32_kernel<4, WAVES_M, NSUB, BKT>), grid, dim3(256), 0, s, g); break;
    }
}

// ---------------------------------------------------------------------------------------------
// latency kernel for the decoder's short problems (M <= 512): 32 x 32 output tile per workgroup,
// the four waves split K between them (intra-workgroup split-K, reduced through LDS in a fixed
// order -> deterministic), operands stream HBM/L2 -> registers with every load of a wave's K range
// in flight at once.  A decode step is ~60 dependent launches of this size: what matters is the
// length of the dependent chain inside a launch (here: one memory round trip + K/8 MFMAs), and
// how many CUs share the weight stream ((M/32) x (N/32) workgroups instead of (M/32) x (N/128)).
// ---------------------------------------------------------------------------------------------
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_splitk4_kernel(const GemmArgs g) {
    // per wave: a [32 rows x 64 k] slab of each operand (272-byte row pitch: conflict-free for the 16-byte
    // row-major writes and for the fragment reads alike); the partial tiles reuse the front of it
    constexpr int PITCH = 68;                      // floats
    constexpr int SLAB = 32 * PITCH;
    __shared__ __attribute__((aligned(16))) float smem[NW * 2 * SLAB];
    float* part = smem;
    const int tiles_n = g.tiles_n;
    const int64_t m0 = (int64_t)(blockIdx.x / (unsigned)tiles_n) * 32;
    const int n0 = (int)(blockIdx.x % (unsigned)tiles_n) * 32;
    const int64_t M = g.M;
    const int N = g.N, K = g.K;
    const int z1 = (int)blockIdx.y / g.nb2, z2 = (int)blockIdx.y % g.nb2;
    const float* A = g.A + z1 * g.a_s1 + z2 * g.a_s2;
    const float* W = g.W + z1 * g.w_s1 + z2 * g.w_s2;
    float* Y = g.Y + z1 * g.y_s1 + z2 * g.y_s2;
    const float* bias = g.bias ? g.bias + z2 * g.bias_s2 : nullptr;
    const float* res = g.res ? g.res + z1 * g.r_s1 + z2 * g.r_s2 : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int r32 = lane & 31, kh = lane >> 5;
    // Loads: one instruction covers 4 rows x 256 contiguous bytes (lane -> row 4j + (lane >> 4), 16-byte column
    // lane & 15), i.e. 8 cache lines.  (Loading the MFMA fragments directly -- lane -> row lane & 31, 16 bytes --
    // touches 32 lines for 32 useful bytes each, and the CU's line-request rate, not latency, then sets the
    // kernel's duration: 7 us + 6.8 us per 1024 of K, scripts/bench_small_gemm.py.)
    const int lrow = lane >> 4, lcol = (lane & 15) * 4;
    const float* aptr[8];
    const float* wptr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t ar = m0 + 4 * j + lrow;
        const int wr = n0 + 4 * j + lrow;
        aptr[j] = A + (ar < M ? ar : M - 1) * g.lda + lcol;
        wptr[j] = W + (int64_t)(wr < N ? wr : N - 1) * g.ldw + lcol;
    }
    float* As = smem + w * (2 * SLAB);
    float* Ws = As + SLAB;

    const int steps = (K + 7) / 8;                 // one step = 8 consecutive k (two 16-byte halves)
    const int per = (steps + NW - 1) / NW;
    const int s0 = w * per;
    const int s1 = (s0 + per < steps) ? s0 + per : steps;
    const int kend = s1 * 8 < K ? s1 * 8 : K;      // this wave's K range is [8 s0, kend)
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int UN = 8;                          // steps per batch = 64 k = one slab row
    for (int sb = s0; sb < s1; sb += UN) {
        const int k0 = sb * 8 + lcol;
        const bool in = k0 < kend;                 // K % 4 == 0: a 16-byte column is in or out as a whole
        const int kc = in ? k0 - lcol : 0;         // (clamped, always valid address; zero-filled below)
        f32x4 la[8], lw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            la[j] = *reinterpret_cast<const f32x4*>(aptr[j] + kc);
            lw[j] = *reinterpret_cast<const f32x4*>(wptr[j] + kc);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *reinterpret_cast<f32x4*>(As + (4 * j + lrow) * PITCH + lcol) = in ? la[j] : zero4;
            *reinterpret_cast<f32x4*>(Ws + (4 * j + lrow) * PITCH + lcol) = in ? lw[j] : zero4;
        }
        // (each wave reads back only what it wrote: LDS operations of one wave are processed in order)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(As + r32 * PITCH + 8 * u + 4 * kh);
            const f32x4 b = *reinterpret_cast<const f32x4*>(Ws + r32 * PITCH + 8 * u + 4 * kh);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
    __syncthreads();                               // every wave is done with its slabs before they become `part`
    // partial tiles -> LDS as [wave][row][col]; C/D layout: col = lane & 31, row = (e&3)+8*(e>>2)+4*(lane>>5)
#pragma unroll
    for (int e = 0; e < 16; ++e) part[w * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + r32] = acc[e];
    __syncthreads();
    if (tid >= 256) return;                        // the first four waves add the partial tiles and write
    const int row = tid >> 3, c = (tid & 7) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(&part[row * 32 + c]);
#pragma unroll
    for (int q = 1; q < NW; ++q) v += *reinterpret_cast<const f32x4*>(&part[q * 1024 + row * 32 + c]);  // waves 0..NW-1 in order
    const int64_t gr = m0 + row;
    const int gc = n0 + c;
    if (gr >= M || gc >= N) return;
    const float alpha = g.alpha;
    const bool vec = gc + 3 < N && (g.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) &&
                     (MODE != 2 || (g.ldres % 4 == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0));
    if (vec) {
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + gc);
        if (MODE == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (MODE == 2) v = *reinterpret_cast<const f32x4*>(res + gr * g.ldres + gc) + alpha * v;
        if (MODE == 3 && (g.scale_cols == 0 || gc < g.scale_cols)) v = alpha * v;
        *reinterpret_cast<f32x4*>(Y + gr * g.ldy + gc) = v;
    } else {
        for (int q = 0; q < 4 && gc + q < N; ++q) {
            float x = v[q] + (bias ? bias[gc + q] : 0.f);
            if (MODE == 1) x = fmaxf(x, 0.f);
            if (MODE == 2) x = res[gr * g.ldres + gc + q] + alpha * x;
            if (MODE == 3 && (g.scale_cols == 0 || gc + q < g.scale_cols)) x = alpha * x;
            Y[gr * g.ldy + gc + q] = x;
        }
    }
}

static void launch_splitk4(const GemmArgs& g, int mode, dim3 grid, hipStream_t s) {
    // (8 waves splitting K were measured for the decoder's K = 2048 layer: 15.6 vs 16.4 us, not worth a variant)
    switch (mode) {
        case 0: hipLaunchKernelGGL((gemm_splitk4_kernel<0, 4>), grid, dim3(256), 0, s, g); break;
        case 1: hipLaunchKernelGGL((gemm_splitk4_kernel<1, 4>), grid, dim3(256), 0, s, g); break;
        case 2: hipLaunchKernelGGL((gemm_splitk4_kernel<2, 4>), grid, dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL((gemm_splitk4_kernel<3, 4>), grid, dim3(256), 0, s, g); break;
    }
}

// fix-up of the split-K tail: FIX_PARTS workgroups per tail tile (32 rows each), slices added in ascending
// order.  All slice loads of a position are issued before the first add; 16-byte loads and stores.
template <int MODE, int NSUB, int BM = 128>
__global__ __launch_bounds__(256) void gemm_splitk_fixup_kernel(const GemmArgs g) {
    constexpr int FIX_PARTS = BM / 32;         // 32 rows per workgroup
    constexpr int BN = 32 * NSUB, RPB = BM / FIX_PARTS, MAXS = 8;
    const unsigned tile = blockIdx.x / FIX_PARTS, part = blockIdx.x % FIX_PARTS;
    const unsigned logical = (unsigned)g.tile_base + tile;
    const int64_t m0 = (int64_t)(logical / (unsigned)g.tiles_n) * BM + part * RPB;
    const int n0 = (int)(logical % (unsigned)g.tiles_n) * BN;
    const float* base = g.splitk_ws + (size_t)tile * g.split * (BM * BN) + part * RPB * BN;
    const bool vec_ok = (g.ldy % 4 == 0) && (MODE != 2 || g.ldres % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.Y) & 15) == 0) &&
                        (MODE != 2 || (reinterpret_cast<uintptr_t>(g.res) & 15) == 0) &&
                        (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < RPB * BN / 4; i += 256) {
        const int r = i / (BN / 4), c = (i - r * (BN / 4)) * 4;
        const int64_t row = m0 + r;
        const int col = n0 + c;
        if (row >= g.M || col >= g.N) continue;
        f32x4 p[MAXS];
#pragma unroll
        for (int q = 0; q < MAXS; ++q)
            p[q] = q < g.split ? *reinterpret_cast<const f32x4*>(base + (size_t)q * (BM * BN) + r * BN + c) : zero4;
        const bool full = vec_ok && col + 3 < g.N;
        f32x4 rv = zero4, bv = zero4;
        if (full) {
            if (MODE == 2 && g.res_split) {
                const _Float16* rr = reinterpret_cast<const _Float16*>(g.res + row * g.ldres) + (col >> 5) * 64 + (col & 31);
                const f16x4 h4 = *reinterpret_cast<const f16x4*>(rr), l4 = *reinterpret_cast<const f16x4*>(rr + 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) rv[q] = (float)h4[q] + (float)l4[q] * (1.0f / 2048.0f);
            } else if (MODE == 2)
                rv = *reinterpret_cast<const f32x4*>(g.res + row * g.ldres + col);
            if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
        }
        f32x4 v = p[0];
#pragma unroll
        for (int q = 1; q < MAXS; ++q)
            if (q < g.split) v += p[q];
        if (full) {
            v += bv;
            if (MODE == 1) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (MODE == 2) v = rv + g.alpha * v;
            if (MODE == 3 && (g.scale_cols == 0 || col < g.scale_cols)) v = g.alpha * v;
            if (g.out_split) {
                f16x4 hi, lo;
                note_range(amax4(0.f, v), g.range_flag);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    _Float16 h, l;
                    split_f16x3(v[q], h, l);
                    hi[q] = h;
                    lo[q] = l;
                }
                _Float16* yr = reinterpret_cast<_Float16*>(g.Y + row * g.ldy) + (col >> 5) * 64 + (col & 31);
                *reinterpret_cast<f16x4*>(yr) = hi;
                *reinterpret_cast<f16x4*>(yr + 32) = lo;
                continue;
            }
            *reinterpret_cast<f32x4*>(g.Y + row * g.ldy + col) = v;
            continue;
        }
        for (int e = 0; e < 4 && col + e < g.N; ++e) {
            float x = v[e] + (g.bias ? g.bias[col + e] : 0.f);
            if (MODE == 1) x = fmaxf(x, 0.f);
            if (MODE == 2) x = g.res[row * g.ldres + col + e] + g.alpha * x;
            if (MODE == 3 && (g.scale_cols == 0 || col + e < g.scale_cols)) x = g.alpha * x;
            g.Y[row * g.ldy + col + e] = x;
        }
    }
}

template <int NSUB, int BM = 128>
static void launch_fixup(const GemmArgs& g, int mode, dim3 grid, hipStream_t s) {
    switch (mode) {
        case 0: hipLaunchKernelGGL((gemm_splitk_fixup_kernel<0, NSUB, BM>), grid, dim3(256), 0, s, g); break;
        case 1: hipLaunchKernelGGL((gemm_splitk_fixup_kernel<1, NSUB, BM>), grid, dim3(256), 0, s, g); break;
        case 2: hipLaunchKernelGGL((gemm_splitk_fixup_kernel<2, NSUB, BM>), grid, dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL((gemm_splitk_fixup_kernel<3, NSUB, BM>), grid, dim3(256), 0, s, g); break;
    }
}


constexpr int SPLITK_MAX_SLICES = 1024;  // tail tiles x slices (84 MB of scratch)
size_t gemm_splitk_ws_bytes() { return (size_t)SPLITK_MAX_SLICES * 128 * 160 * sizeof(float); }

template <int MODE, int NSUB, bool SPLITK>
static void launch_glds_stage(const GemmArgs& g, dim3 grid, hipStream_t s) {
    // the buffer form of the operand loads (default) needs the tile's lane offsets to fit 32 bits
    const int want = opt(OPT_GEMM_GLOBAL_LOADS) ? 0 : 2;
    if (g.f16x3) {
        if constexpr (MODE == 1 || MODE == 2) {
            // the TDS block's layers: split-form output under the range guard (MODE 2: split-form residual too)
            const bool y_ok = g.ldy % 4 == 0 && g.ldy < (1 << 21) && (reinterpret_cast<uintptr_t>(g.Y) & 15) == 0 && g.N % (32 * NSUB) == 0;
            const bool r_ok = MODE != 2 || (g.res_split && g.ldres % 4 == 0 && g.ldres < (1 << 21) && (reinterpret_cast<uintptr_t>(g.res) & 15) == 0);
            if (g.out_split && g.range_flag && y_ok && r_ok && grid.y == 1) {
                hipLaunchKernelGGL((gemm_glds_kernel<MODE, NSUB, SPLITK, 2, true, 1>), grid, dim3(256), 0, s, g);
                return;
            }
        }
        if constexpr (MODE <= 3) hipLaunchKernelGGL((gemm_glds_kernel<MODE, NSUB, SPLITK, 2, true>), grid, dim3(256), 0, s, g);
        return;
    }
    if (want == 2 && g.lda < (1 << 21) && g.ldw < (1 << 21))
        hipLaunchKernelGGL((gemm_glds_kernel<MODE, NSUB, SPLITK, 2>), grid, dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_glds_kernel<MODE, NSUB, SPLITK, 0>), grid, dim3(256), 0, s, g);
}

// splitk: the grid is [whole-round tiles | K slices of the remaining tiles] (modes 0..3 only)
template <int NSUB>
static void launch_glds(const GemmArgs& g, int mode, bool splitk, dim3 grid, hipStream_t s) {
    if (splitk) {
        switch (mode) {
            case 0: launch_glds_stage<0, NSUB, true>(g, grid, s); break;
            case 1: launch_glds_stage<1, NSUB, true>(g, grid, s); break;
            case 2: launch_glds_stage<2, NSUB, true>(g, grid, s); break;
            default: launch_glds_stage<3, NSUB, true>(g, grid, s); break;
        }
        return;
    }
    switch (mode) {
        case 0: launch_glds_stage<0, NSUB, false>(g, grid, s); break;
        case 1: launch_glds_stage<1, NSUB, false>(g, grid, s); break;
        case 2: launch_glds_stage<2, NSUB, false>(g, grid, s); break;
        case 3: launch_glds_stage<3, NSUB, false>(g, grid, s); break;
        default: launch_glds_stage<4, NSUB, false>(g, grid, s); break;
    }
}

int launch_gemm(GemmArgs g, int mode, int nbatch, hipStream_t s) {
    TAL_CHECK_ARG(g.A && g.W && (g.Y || mode == 4), "gemm: null pointer");
    TAL_CHECK_ARG(mode != 4 || (g.part_val && g.part_idx && g.part_ld >= gemm_mode4_partials(g.M, g.N)),
                  "gemm: mode 4 needs partial buffers with part_ld >= %d", gemm_mode4_partials(g.M, g.N));
    TAL_CHECK_ARG(g.M >= 0 && g.N > 0 && g.K > 0, "gemm: bad shape M=%lld N=%d K=%d", (long long)g.M, g.N, g.K);
    TAL_CHECK_ARG(g.K % 4 == 0 && g.lda % 4 == 0 && g.ldw % 4 == 0, "gemm: K=%d lda=%lld ldw=%lld must be multiples of 4", g.K, (long long)g.lda, (long long)g.ldw);
    TAL_CHECK_ARG(mode >= 0 && mode <= 4, "gemm: mode %d", mode);
    TAL_CHECK_ARG(mode != 2 || g.res, "gemm: mode 2 needs a residual");
    TAL_CHECK_ARG(g.scale_cols % 4 == 0, "gemm: scale_cols must be a multiple of 4");
    TAL_CHECK_ARG(nbatch >= 1 && nbatch <= 65535 && g.nb2 >= 1, "gemm: batch %d", nbatch);
    if (g.M == 0) return TAL_OK;
    if (g.f16x3 || g.out_split) {
        TAL_CHECK_ARG(g.f16x3, "gemm: out_split needs the fp16x3 form");
        TAL_CHECK_ARG(g.M > 128 && g.K % BK == 0 && mode <= 3 && nbatch == 1, "gemm: the fp16x3 form needs M > 128, K %% 32 == 0, modes 0-3");
        TAL_CHECK_ARG(!g.out_split || (g.N % 160 == 0 && g.ldy % 4 == 0), "gemm: split output needs N %% 160 == 0");
        TAL_CHECK_ARG(g.lda < (1 << 21) && g.ldw < (1 << 21), "gemm: leading dimension too large for the fp16x3 form");
    }
    TAL_CHECK_ARG(!g.res_split || (g.f16x3 && mode == 2 && g.N % 160 == 0 && g.ldres % 32 == 0 && g.ldy < (1 << 21) && g.ldres < (1 << 21) &&
                                  (reinterpret_cast<uintptr_t>(g.res) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.Y) & 15) == 0),
                  "gemm: a split-form residual needs the fp16x3 form, mode 2, N %% 160 == 0");
    // short inputs: 64 x 80 tiles, whole tiles only (gemm_s64.hip), while they fit one round of two workgroups per CU (measured
    // against the K-sliced launches below: ahead up to 470 / 336 / 432 tiles at N = K = 800 / 1120 / 1440, behind from 590 / 658 / 846,
    // profiles/r3_gemm_short_inputs.txt)
    if (g.f16x3 && nbatch == 1 && cdiv(g.M, 64) * (g.N / 80) <= (int64_t)opt(OPT_GEMM_S64_BELOW) * device_cus() &&
        gemm_s64_ok(g, mode)) {
        ProfScope prof(PROF_GEMM, 2.0 * (double)g.M * (double)g.N * (double)g.K, s);
        launch_gemm_s64(g, mode, s);
        TAL_CHECK_LAUNCH("gemm (64 x 80 tiles)");
        return TAL_OK;
    }
    // fp16x3 launches with at least one full round of 256 x 160 tiles (one workgroup per CU) take the one-wave-per-SIMD
    // kernel (gemm_w64.hip); the tiles of its last partial round are cut along K like the 128 x 160 kernel's
    if (g.f16x3 && nbatch == 1 && !opt(OPT_GEMM_NO_W64) && !opt(OPT_GEMM_NO_GLDS) &&
        ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15) == 0) {
        const int64_t slots = device_cus();
        GemmArgs h = g;
        h.tiles_n = (int)cdiv(g.N, 160);
        h.tiles_n_magic = h.tiles_n > 1 ? (unsigned)((1ull << 32) / (unsigned)h.tiles_n) + 1u : 0u;
        const int64_t nb = cdiv(g.M, 256) * h.tiles_n;
        if (nb >= slots && nb < (1ll << 31) && nb * h.tiles_n < (1ll << 32)) {
            // ONE full round and a remainder of a few thousand rows (a 5-minute clip's first stage: 295 tiles): the rows of the
            // whole round's row blocks run here, the rest goes back through this dispatcher as a launch of its own (64 x 80 or
            // 128 x 96 / 160 tiles) -- K slices of the 39 leftover tiles plus their fix-up launch cost 34 us on top of the
            // round's 45, the second launch ~16.  Rows are independent: results do not change.
            const int64_t rb = slots / h.tiles_n, rows1 = rb * 256;
            if (nb < 2 * slots && nb % slots != 0 && g.M - rows1 > 128 && g.M - rows1 <= 4096 && !opt(OPT_GEMM_NO_ROW_SPLIT)) {
                {
                    ProfScope prof(PROF_GEMM, 2.0 * (double)rows1 * (double)g.N * (double)g.K, s);
                    GemmArgs a = h;
                    a.M = rows1;
                    launch_gemm_w64(a, mode, false, dim3((unsigned)(rb * h.tiles_n)), s);
                    TAL_CHECK_LAUNCH("gemm (256 x 160 tiles)");
                }
                GemmArgs r = g;
                r.M = g.M - rows1;
                r.A = g.A + rows1 * g.lda;
                r.Y = g.Y ? g.Y + rows1 * g.ldy : nullptr;
                r.res = g.res ? g.res + rows1 * g.ldres : nullptr;
                return launch_gemm(r, mode, 1, s);
            }
            ProfScope prof(PROF_GEMM, 2.0 * (double)g.M * (double)g.N * (double)g.K, s);
            const int64_t rem = nb % slots, full = nb - rem;
            const int nk = g.K / BK;
            int split = 0;
            if (rem > 0 && g.splitk_ws && !opt(OPT_GEMM_NO_SPLITK_TAIL)) {
                // as below: slice rounds + scratch traffic (a slice writes and the fix-up re-reads a 256 x 160 fp32 tile) against
                // one more round (measured: ~1.5 us per K step + ~12 us per tile, scripts/ubench/gemm_f16x3_w64.hip)
                const double round_us = 1.5 * nk + 12.0;
                double best = 0.97;
                for (int sp = 2; sp <= 8 && sp <= nk / 4; ++sp) {
                    if ((size_t)rem * sp * 256 * 160 * sizeof(float) > g.splitk_ws_bytes) break;
                    const double traffic_us = (double)rem * sp * (2.0 * 256 * 160 * 4) / 4.0e6;
                    const double tail = (double)cdiv(rem * sp, slots) / sp + traffic_us / round_us;
                    if (tail < best) { best = tail; split = sp; }
                }
            }
            if (split < 2) {
                launch_gemm_w64(h, mode, false, dim3((unsigned)nb), s);
            } else {
                h.tile_base = (int)full;
                h.split = split;
                h.tail_tiles = (int)rem;
                launch_gemm_w64(h, mode, true, dim3((unsigned)(full + rem * split)), s);
                launch_fixup<5, 256>(h, mode, dim3((unsigned)rem * 8), s);
            }
            TAL_CHECK_LAUNCH("gemm (256 x 160 tiles)");
            return TAL_OK;
        }
    }
    const bool small = g.M <= 512 && !g.f16x3;
    const int bm = small ? 32 : 128, bn = small ? 128 : 160;
    g.tiles_n = (int)cdiv(g.N, bn);
    g.tiles_n_magic = g.tiles_n > 1 ? (unsigned)((1ull << 32) / (unsigned)g.tiles_n) + 1u : 0u;
    const int64_t nb = cdiv(g.M, bm) * g.tiles_n;
    TAL_CHECK_ARG(nb < (1ll << 31) && nb * g.tiles_n < (1ll << 32), "gemm: grid too large");
    dim3 grid((unsigned)nb, (unsigned)nbatch);
    ProfScope prof(PROF_GEMM, 2.0 * (double)g.M * (double)g.N * (double)g.K * nbatch, s);
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15) == 0;
    // measurement switches (tal_set_option; three relaxed atomic loads per launch)
    const bool no_splitk4 = opt(OPT_GEMM_NO_SPLITK4) != 0, no_glds = opt(OPT_GEMM_NO_GLDS) != 0, no_tail = opt(OPT_GEMM_NO_SPLITK_TAIL) != 0;
    // small problems are latency-bound (a K step costs one L2 round trip, not its 16 MFMAs): a
    // 128-deep K slab quarters the number of dependent round trips
    if (small && mode != 4 && aligned16 && !no_splitk4) {
        GemmArgs h = g;
        h.tiles_n = (int)cdiv(g.N, 32);
        const int64_t nb2 = cdiv(g.M, 32) * h.tiles_n;
        TAL_CHECK_ARG(nb2 < (1ll << 31), "gemm: grid too large");
        launch_splitk4(h, mode, dim3((unsigned)nb2, (unsigned)nbatch), s);
    } else if (small && g.K >= 256)
        launch_tile<1, 1, 128>(g, mode, grid, s);
    else if (small)
        launch_tile<1, 1, 32>(g, mode, grid, s);
    else if (g.K % BK == 0 && aligned16 && !no_glds) {
        // (a 128 x 96 tile -- 10.3 instead of 6.2 rounds on the 1-hour stage-3 shape -- was measured at
        //  +2 %, inside run-to-run noise: the 160-wide tile stays)
        // Stream-K-lite: 2 workgroups per CU are resident, so a launch proceeds in rounds of 2*CUs
        // tiles; the tiles of the last, partial round are cut along K into `split` slices each, so that
        // they occupy the whole chip for a fraction of a round instead of part of it for a whole one
        // (1-hour stage-3 shape: 3168 tiles = 6.19 rounds).  The slices are blocks of the same launch,
        // dispatched behind the whole tiles; a fix-up kernel adds them in a fixed order.  `split` is
        // the one that minimises the tail, ceil(rem*split/slots)/split rounds, within the scratch size.
        const int64_t slots = 2 * (int64_t)device_cus();
        const int64_t rem = nb % slots, full = nb - rem;
        const int nk = g.K / BK;
        int split = 0;
        // (a launch with less than a quarter of a round of tiles -- a few hundred rows -- is cut along K as a whole: its
        //  few tiles become up to 8x as many K slices, so that most of the chip takes part)
        if (rem > 0 && (full > 0 || 4 * rem <= slots) && nbatch == 1 && mode <= 3 && g.splitk_ws && !no_tail) {
            // cost of a candidate in rounds: the slices' own rounds + their scratch traffic (each slice
            // writes and the fix-up re-reads a 128x160 fp32 tile, ~4 TB/s) relative to one round's duration
            // (measured: 4.45 us per K step + ~14 us per tile, scripts/bench_gemm_fit.py)
            // (fp16x3 form: 1.56 us per K step + ~10.5 us per tile, scripts/ubench/gemm_f16x3.hip)
            const double round_us = g.f16x3 ? 1.56 * nk + 10.5 : 4.45 * nk + 14.0;
            double best = 0.97;
            for (int sp = 2; sp <= 8 && sp <= nk / 4; ++sp) {
                if ((size_t)rem * sp * 128 * 160 * sizeof(float) > g.splitk_ws_bytes) break;
                const double traffic_us = (double)rem * sp * (2.0 * 128 * 160 * 4) / 4.0e6;
                const double tail = (double)cdiv(rem * sp, slots) / sp + traffic_us / round_us;
                if (tail < best) { best = tail; split = sp; }
            }
        }
        // One partial round of 160-wide tiles that already puts two workgroups on some CUs (256 < tiles <= 512) runs as long as
        // a full one.  When N is a multiple of 96 and the 5/3 as many 96-wide tiles still fit the round, they finish in ~0.6 of
        // it with every CU busy (5-minute clip, stage 3: 270 -> 450 tiles).  Same MFMA chain per output: bit-identical results.
        // Cost per K step of a round, us (scripts/bench_gemm_short.py): 0.82 with one workgroup per CU, 1.27 with two, x NSUB / 5.
        bool n96 = false;
        int64_t nb3 = 0;
        // (only the split-form layers of a TDS block -- the launches launch_glds_stage sends to the static-addressing epilogue; the
        //  generic epilogue's row / column arithmetic is written for 160- and 32-column waves)
        const bool epi_ok = g.out_split && g.range_flag && g.ldy % 4 == 0 && g.ldy < (1 << 21) && (reinterpret_cast<uintptr_t>(g.Y) & 15) == 0 &&
                            (mode != 2 || (g.res_split && g.ldres % 4 == 0 && g.ldres < (1 << 21) && (reinterpret_cast<uintptr_t>(g.res) & 15) == 0));
        if (split < 2 && epi_ok && g.f16x3 && nbatch == 1 && (mode == 1 || mode == 2) && g.N % 96 == 0 && !opt(OPT_GEMM_NO_N96)) {
            nb3 = cdiv(g.M, 128) * (g.N / 96);
            auto cost = [&](int64_t tiles, double scale) {
                const int64_t fr = tiles / slots, r = tiles % slots;
                const double two = 10.0 + 1.27 * scale * nk, one = 10.0 + 0.82 * scale * nk;
                return fr * two + (r == 0 ? 0.0 : (r <= slots / 2 ? one : two));
            };
            n96 = nb3 < (1ll << 31) && cost(nb3, 0.6) < 0.95 * cost(nb, 1.0);
        }
        if (n96) {
            GemmArgs h = g;
            h.tiles_n = g.N / 96;
            h.tiles_n_magic = h.tiles_n > 1 ? (unsigned)((1ull << 32) / (unsigned)h.tiles_n) + 1u : 0u;
            if (mode == 1) launch_glds_stage<1, 3, false>(h, dim3((unsigned)nb3), s);
            else launch_glds_stage<2, 3, false>(h, dim3((unsigned)nb3), s);
        } else if (split < 2) {
            launch_glds<5>(g, mode, false, grid, s);
        } else {
            GemmArgs h = g;
            h.tile_base = (int)full;
            h.split = split;
            h.tail_tiles = (int)rem;
            launch_glds<5>(h, mode, true, dim3((unsigned)(full + rem * split)), s);   // whole tiles, then K slices -> scratch
            launch_fixup<5>(h, mode, dim3((unsigned)rem * 4), s);                                 // ordered sum + epilogue (4 x 32 rows per tile)
        }
    }
    else
        launch_tile<4, 5, 32>(g, mode, grid, s);
    TAL_CHECK_LAUNCH("gemm");
    return TAL_OK;
}

int launch_linear_ws(const float* x, const float* w, const float* b, const float* res, float alpha, int mode, int64_t M,
                     int N, int K, float* y, float* ws, size_t ws_bytes, hipStream_t s) {
    TAL_CHECK_ARG(x && w && y, "tal_linear_fwd: null pointer");
    GemmArgs g = {};
    g.A = x; g.W = w; g.bias = b; g.res = res; g.Y = y;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldy = N; g.ldres = N;
    g.nb2 = 1;
    g.alpha = alpha;
    g.splitk_ws = ws;
    g.splitk_ws_bytes = ws_bytes;
    return launch_gemm(g, mode, 1, s);
}

// fp32 -> hi / lo fp16 split in fp32-row geometry: per row and 32-wide K block, 32 hi halves then 32 lo halves
// (lo = (x - hi) * 2^11): one thread per 4 consecutive floats
__global__ __launch_bounds__(256) void split_f16x3_kernel(const float* __restrict__ x, _Float16* __restrict__ out, int64_t n4,
                                                         int* __restrict__ range_flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    note_range(amax4(0.f, v), range_flag);
    f16x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        _Float16 h, l;
        split_f16x3(v[q], h, l);
        hi[q] = h;
        lo[q] = l;
    }
    // element index e = 4 i inside a row-major [rows, K] array with K % 32 == 0: block e / 32 (64 halves), slot e % 32
    const int64_t e = 4 * i;
    _Float16* o = out + (e >> 5) * 64 + (e & 31);
    *reinterpret_cast<f16x4*>(o) = hi;
    *reinterpret_cast<f16x4*>(o + 32) = lo;
}

int launch_split_f16x3(const float* x, void* out, int64_t rows, int K, hipStream_t s, int* range_flag) {
    TAL_CHECK_ARG(x && out && rows >= 0 && K > 0 && K % 32 == 0, "split_f16x3: bad argument (K %% 32 == 0 required)");
    TAL_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "split_f16x3: 16-byte alignment required");
    const int64_t n4 = rows * K / 4;
    if (n4 == 0) return TAL_OK;
    ProfScope prof(PROF_OTHER, (double)rows * K * 8.0, s);
    hipLaunchKernelGGL(split_f16x3_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, s, x, reinterpret_cast<_Float16*>(out), n4, range_flag);
    TAL_CHECK_LAUNCH("split_f16x3");
    return TAL_OK;
}

int launch_linear_f16x3(const void* xs, const void* wsplit, const float* b, const float* res, float alpha, int mode, int64_t M,
                        int N, int K, void* y, int out_split, float* ws, size_t ws_bytes, hipStream_t s, int* range_flag, int res_split) {
    TAL_CHECK_ARG(xs && wsplit && y, "linear_f16x3: null pointer");
    GemmArgs g = {};
    g.A = reinterpret_cast<const float*>(xs); g.W = reinterpret_cast<const float*>(wsplit); g.bias = b; g.res = res;
    g.Y = reinterpret_cast<float*>(y);
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldy = N; g.ldres = N;
    g.nb2 = 1;
    g.alpha = alpha;
    g.splitk_ws = ws;
    g.splitk_ws_bytes = ws_bytes;
    g.f16x3 = 1;
    g.out_split = out_split;
    g.range_flag = range_flag;
    g.res_split = res_split;
    return launch_gemm(g, mode, 1, s);
}

// number of (value, index) partials per row a mode-4 launch of this shape produces
int gemm_mode4_partials(int64_t M, int N) { return M <= 512 ? (int)cdiv(N, 128) * 4 : (int)cdiv(N, 160); }

int launch_linear(const float* x, const float* w, const float* b, const float* res, float alpha, int mode, int64_t M,
                  int N, int K, float* y, hipStream_t s) {
    TAL_CHECK_ARG(x && w && y, "tal_linear_fwd: null pointer");
    GemmArgs g = {};
    g.A = x; g.W = w; g.bias = b; g.res = res; g.Y = y;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldy = N; g.ldres = N;
    g.nb2 = 1;
    g.alpha = alpha;
    return launch_gemm(g, mode, 1, s);
}

}  // namespace tal
