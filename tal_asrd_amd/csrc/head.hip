// Speaker-logit arg-max of the diarization head without the logits:
//   ids[r] = argmax_s ( feat[r, :] . W[s, :] + b[s] ),  feat [M, 128], W [S, 128]   (S = 6008)
// SDModel.decode's spk_logit_proj followed by the arg-max of tal/baseline/reconcile.py:84.
//
// The generic dense-layer kernel spends more time around this GEMM's 4 K steps per 128x160 tile than in
// them (set-up, arg-max epilogue and partial stores per tile: 13,376 tiles per hour of audio).  Here the
// [128 x 128] feature strip of a workgroup is A-STATIONARY IN REGISTERS (each wave keeps its 32 x 128 strip as
// 16 MFMA-ready fragments, 64 VGPRs), only W streams through LDS (LDS-DMA, double-buffered 128 x 32 chunks,
// one barrier per chunk), and the arg-max is a running (value, column) pair per accumulator element that
// stays in registers across N tiles -- no cross-lane work and no memory traffic until a row block is done.
// Work is cut into equal runs of consecutive (row block, N tile) units -- 2 x CUs runs for long inputs, so a row block's 47 N
// tiles (N tile = 128 columns: 6008 = 46 x 128 + 120) are covered by at most 3 workgroups; inputs of a few thousand rows get
// shorter runs (down to 4 N tiles: up to 16 partial slots per row, head_argmax_plan) so that a clip of minutes still puts a
// workgroup on every CU.  Each workgroup leaves one (value, column) partial per row; the existing
// argmax_partials_kernel merges them in ascending column order (strict '>': lowest index wins ties, as torch.argmax does).
#include "common.h"

namespace tal {

namespace {

constexpr int HK = 128;            // feature width (K)
constexpr int HBM = 128, HBN = 128, HNSUB = 4;
constexpr int HP_MAX = 16;         // most partial slots per row: a run is at least NT / 15 N tiles (4 of 47) long

__device__ __forceinline__ void* uniform_ptr(const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
}

// F16X3: W arrives as the hi / lo fp16 split of tal_split_f16x3_fwd (same bytes per row and 32-wide K block, so the
// LDS-DMA stream and swizzle are unchanged), the feature strip is split once per row block into MFMA-ready fp16
// fragments, and every fp32 product block becomes three v_mfma_f32_32x32x16_f16 with fp32 accumulation (the dense
// layers' fp16x3 form: error below an fp32 fmaf chain, a sixth of the matrix-pipe time of the fp32 MFMAs).
template <bool F16X3>
__global__ __launch_bounds__(256, 2) void head_argmax_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                            const float* __restrict__ bias, int64_t M, int S, int NT,
                                                            int64_t U, float* __restrict__ part_val,
                                                            int32_t* __restrict__ part_idx, int HP) {
    __shared__ __attribute__((aligned(16))) float lds[2 * HBN * 32];   // 40,960 B
    const int64_t G = gridDim.x;
    const int64_t u0 = (int64_t)blockIdx.x * U / G, u1 = ((int64_t)blockIdx.x + 1) * U / G;
    if (u0 >= u1) return;
    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int frow = lane & 31, fsw = (frow >> 1) & 7, fhalf = lane >> 5;     // swizzle: see gemm_glds_kernel
    const int sub = lane >> 3, srccol = ((lane & 7) ^ (((w & 1) * 4 + (lane >> 4)) & 7)) * 4;

    // W: one descriptor over the whole matrix; rows past S read as 0 (their columns are masked below)
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(W), 0, S * HK * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(bias), 0, S * 4, 0x00020000);
    int wrow[HNSUB];      // byte offset of this lane's source row / 16-byte column inside an N tile
#pragma unroll
    for (int t = 0; t < HNSUB; ++t) wrow[t] = ((8 * (w + 4 * t) + sub) * HK + srccol) * 4;
    // chunk (n, kt): W rows [160 n, 160 n + 160), k in [32 kt, 32 kt + 32) -> LDS buffer `buf` (same image as
    // the dense-layer kernel's W part: chunk i = w + 4t holds 8 rows x 128 B, XOR-swizzled on the source side)
    auto issue = [&](int n, int kt, int buf) {
        const int nbase = n * (HBN * HK * 4);      // row offset goes into the lane offset: it is range-checked
#pragma unroll
        for (int t = 0; t < HNSUB; ++t) {
            float* dst = lds + buf * (HBN * 32) + (w + 4 * t) * 256;
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)dst, 16, wrow[t] + nbase,
                                                     kt * 128, 0, 0);
#endif
        }
    };

    f32x4 a[F16X3 ? 1 : HK / 8];     // this wave's 32 x 128 strip: a[kk] = A[row, 8 kk + 4 (lane >> 5) .. + 3]
    f16x8 ah[F16X3 ? HK / 16 : 1], al[F16X3 ? HK / 16 : 1];   // fp16x3: (hi, lo) of A[row, 16 q + 8 (lane >> 5) .. + 7], q = 2 kt + g
    float best[16];
    int bidx[16];
    f32x16 acc[HNSUB], accx[F16X3 ? HNSUB : 1];
#pragma unroll
    for (int j = 0; j < HNSUB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int j = 0; j < (F16X3 ? HNSUB : 1); ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accx[j][e] = 0.f;

    auto block_of = [&](int64_t u) {          // workgroup whose run contains unit u
        int64_t b = u * G / U;
        while ((b + 1) * U / G <= u) ++b;
        while (b * U / G > u) --b;
        return b;
    };

    issue((int)(u0 % NT), 0, 0);
    for (int64_t u = u0; u < u1; ++u) {
        const int64_t m = u / NT;
        const int n = (int)(u - m * NT);
        const int64_t row0 = m * HBM + w * 32;
        if (u == u0 || n == 0) {
            int64_t r = row0 + frow;
            r = r < M ? r : M - 1;
            if (F16X3) {
                const float* ap = feat + r * HK + 8 * fhalf;
#pragma unroll
                for (int q = 0; q < HK / 16; ++q) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(ap + 16 * q), v1 = *reinterpret_cast<const f32x4*>(ap + 16 * q + 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        _Float16 h, l;
                        split_f16x3(v0[i], h, l);
                        ah[q][i] = h; al[q][i] = l;
                        split_f16x3(v1[i], h, l);
                        ah[q][4 + i] = h; al[q][4 + i] = l;
                    }
                }
            } else {
                const float* ap = feat + r * HK + 4 * fhalf;
#pragma unroll
                for (int kk = 0; kk < HK / 8; ++kk) a[kk] = *reinterpret_cast<const f32x4*>(ap + 8 * kk);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                best[e] = -INFINITY;
                bidx[e] = 0x7fffffff;
            }
        }
        float bv[HNSUB];
#pragma unroll
        for (int kt = 0; kt < HK / 32; ++kt) {
            const int buf = kt & 1;
            // THIS wave's quarter of chunk (n, kt) has landed before it arrives at the barrier: nothing orders an LDS-DMA load in
            // front of a barrier by itself (the compiler only waits for it in front of this wave's OWN LDS reads; at the kt == 0
            // barrier of a unit that does not start a row block it emitted vmcnt(21), i.e. no wait) -- the other waves then read
            // rows that are still in flight: a few wrong ids per ~10 calls on 239 k rows (round 5, scripts/r5_head_determinism.py)
#ifndef HEAD_NO_DMA_WAIT      // (ablation build of the measurement in profiles/r5_head_lds_dma_race.txt only)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __syncthreads();      // chunk (n, kt) has landed; everyone is done with the other buffer
            if (kt + 1 < HK / 32)
                issue(n, kt + 1, buf ^ 1);
            else if (u + 1 < u1)
                issue((int)((u + 1) % NT), 0, buf ^ 1);
            if (kt == 0) {        // (behind the barrier: the wait above is not for these)
#pragma unroll
                for (int j = 0; j < HNSUB; ++j) bv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_b, (n * HBN + j * 32 + frow) * 4, 0, 0));
            }
            const float* Bs = lds + buf * (HBN * 32) + frow * 32;
            if (F16X3) {
                // a 128-byte W row of this K block is [32 hi | 32 lo]: 16-byte slot 2g + half (hi), 4 + 2g + half (lo)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int sh = ((2 * g + fhalf) ^ fsw) * 4, sl = ((4 + 2 * g + fhalf) ^ fsw) * 4;
                    f16x8 bh[HNSUB], bl[HNSUB];
#pragma unroll
                    for (int j = 0; j < HNSUB; ++j) {
                        bh[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * 32 + sh);
                        bl[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * 32 + sl);
                    }
                    const f16x8 fah = ah[F16X3 ? kt * 2 + g : 0], fal = al[F16X3 ? kt * 2 + g : 0];
#pragma unroll
                    for (int j = 0; j < HNSUB; ++j) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bh[j], acc[j], 0, 0, 0);
                        accx[F16X3 ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bl[j], accx[F16X3 ? j : 0], 0, 0, 0);
                        accx[F16X3 ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, bh[j], accx[F16X3 ? j : 0], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int sl = ((2 * k4 + fhalf) ^ fsw) * 4;
                f32x4 fb[HNSUB];
#pragma unroll
                for (int j = 0; j < HNSUB; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * 32 + sl);
                const f32x4 fa = a[kt * 4 + k4];
#pragma unroll
                for (int j = 0; j < HNSUB; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[j].x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[j].y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[j].z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[j].w, acc[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting every slice's fragment reads)
            }
        }
        // running arg-max: a lane owns column n*160 + 32 j + (lane & 31) of 16 rows; columns arrive in ascending order
#pragma unroll
        for (int j = 0; j < HNSUB; ++j) {
            const int col = n * HBN + j * 32 + frow;
            const bool ok = col < S;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = (F16X3 ? fmaf(accx[F16X3 ? j : 0][e], 1.0f / 2048.0f, acc[j][e]) : acc[j][e]) + bv[j];
                if (ok && v > best[e]) {
                    best[e] = v;
                    bidx[e] = col;
                }
                acc[j][e] = 0.f;
                if (F16X3) accx[F16X3 ? j : 0][e] = 0.f;
            }
        }
        if (n == NT - 1 || u == u1 - 1) {
            // this workgroup's share of row block m is complete: reduce over the 32 columns-lanes of each half wave
            const int64_t bf = block_of(m * NT), bl = block_of(m * NT + NT - 1);
            const int slot = (int)((int64_t)blockIdx.x - bf);
            const int nslots = (int)(bl - bf + 1);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float bvv = best[e];
                int bi = bidx[e];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(bvv, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ov > bvv || (ov == bvv && oi < bi)) {
                        bvv = ov;
                        bi = oi;
                    }
                }
                const int64_t row = row0 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
                if (frow == 0 && row < M) {
                    part_val[row * HP + slot] = bvv;
                    part_idx[row * HP + slot] = bi;
                    if (slot == 0)
                        for (int sfill = nslots; sfill < HP; ++sfill) {
                            part_val[row * HP + sfill] = -INFINITY;
                            part_idx[row * HP + sfill] = 0x7fffffff;
                        }
                }
            }
        }
    }
}

}  // namespace

// Workgroups of a launch and partial slots per row.  A row block's NT units may be covered by at most `hp` workgroups, i.e. a run
// must be at least NT / (hp - 1) units long.  Long inputs: 2 workgroups per CU with hp = 3 (a batch of eight 5-minute segments,
// 29,864 rows = 10,998 units, takes 458 runs instead of 512).  Fewer rows: the smallest hp of the ladder that still gives every
// CU a workgroup (a 5-minute clip, 30 row blocks: hp = 12, 282 runs of 5 tiles).  Runs shorter than 4 tiles do not pay: below
// ~2,700 rows the generic fused path is as fast or faster (measured on 30-second to 4-minute clips, profiles/r4_head_mid_sizes.txt).
static void head_argmax_plan(int64_t M, int S, int64_t& grid, int& hp) {
    const int64_t nt = cdiv(S, HBN), units = cdiv(M, HBM) * nt, full = 2 * (int64_t)device_cus(), want = device_cus();
    static const int ladder[] = {3, 4, 6, 8, 12, HP_MAX};
    grid = 0;
    hp = 3;
    for (int h : ladder) {
        const int64_t lmin = cdiv(nt, (int64_t)(h - 1)), g = units / lmin < full ? units / lmin : full;
        grid = g;
        hp = h;
        if (g >= want) break;
    }
}

// true when the shape is one this kernel is meant for (the caller falls back to the generic fused path otherwise): at least
// one workgroup per CU
bool head_argmax_applicable(int64_t M, int S, int E) {
    if (E != HK || S < HBN || opt(OPT_HEAD_NO_ASTATIONARY)) return false;
    int64_t grid;
    int hp;
    head_argmax_plan(M, S, grid, hp);
    return grid >= device_cus();
}

int head_argmax_partials(int64_t M, int S) {
    int64_t grid;
    int hp;
    head_argmax_plan(M, S, grid, hp);
    return hp;
}

// w_split != nullptr: W as the hi / lo split of tal_split_f16x3_fwd -> the fp16x3 kernel; else w (fp32) -> the fp32 MFMA kernel
int launch_head_argmax(const float* feat, const float* w, const void* w_split, const float* b, int64_t M, int S, float* part_val,
                       int32_t* part_idx, hipStream_t s) {
    TAL_CHECK_ARG(feat && (w || w_split) && b && part_val && part_idx, "head_argmax: null pointer");
    TAL_CHECK_ARG(((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(w_split)) & 15) == 0, "head_argmax: operands must be 16-byte aligned");
    const int NT = (int)cdiv(S, HBN);
    const int64_t U = cdiv(M, HBM) * NT;
    int64_t grid;
    int hp;
    head_argmax_plan(M, S, grid, hp);
    TAL_CHECK_ARG(grid >= 1, "head_argmax: too few rows (M=%lld)", (long long)M);
    ProfScope prof(PROF_GEMM, 2.0 * (double)M * (double)S * HK, s);
    if (w_split)
        hipLaunchKernelGGL(head_argmax_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, feat, reinterpret_cast<const float*>(w_split), b, M,
                           S, NT, U, part_val, part_idx, hp);
    else
        hipLaunchKernelGGL(head_argmax_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, feat, w, b, M, S, NT, U, part_val, part_idx, hp);
    TAL_CHECK_LAUNCH("head_argmax");
    return TAL_OK;
}

}  // namespace tal
