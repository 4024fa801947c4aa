// Shared helpers for the gfx950 kernels (internal; the public ABI is include/tal_asrd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/tal_asrd.h"

namespace tal {

void set_error(const char* fmt, ...);

#define TAL_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            tal::set_error(__VA_ARGS__);         \
            return TAL_EINVAL;                   \
        }                                        \
    } while (0)

#define TAL_CHECK_LAUNCH(what)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            tal::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));    \
            return TAL_EHIP;                                                          \
        }                                                                             \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// fp32 -> (hi, lo) fp16 pair of the fp16x3 operand form: hi = fp16(x), lo = fp16((x - hi) * 2^11).  Both halves are
// clamped to the finite fp16 range, so an activation beyond +-65504 degrades gracefully instead of becoming inf.
__device__ __forceinline__ void split_f16x3(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)fminf(fmaxf(x, -65504.f), 65504.f);
    lo = (_Float16)fminf(fmaxf((x - (float)hi) * 2048.f, -65504.f), 65504.f);
}

// In-launch hand-off between workgroups (split-K / key-split merges): the payload is written with agent-scope relaxed
// atomic stores (`sc1`, write-through: visible to every XCD without a cache write-back), the writer drains them
// (s_waitcnt vmcnt(0)) and takes a ticket; the last arriver reads with agent-scope relaxed atomic loads (they bypass the
// non-coherent caches).  A __threadfence() here costs a whole-L2 write-back + invalidate (~10 us measured).
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// N agent-scope 16-byte loads issued back to back, then ONE wait: rows p + i * stride (floats).  Inline asm because
// __hip_atomic_load stops at 8 bytes and a 4-byte sc1 load per element makes a merge 4x as many memory transactions.
template <int N>
__device__ __forceinline__ void ld_agent_x4(const float* p, size_t stride, f32x4 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v[i]) : "v"(p + i * stride) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));      // uses of v[i] stay behind the wait
}
__device__ __forceinline__ unsigned take_ticket(unsigned* word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void reset_ticket(unsigned* word) {
    __hip_atomic_store(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// fp16-range guard of the fp16x3 form: every place that turns fp32 data into hi / lo halves tracks max |x| and raises a
// status word when a value lies outside the finite fp16 range (the halves are clamped, so the results of that launch
// are wrong, not inf).  tal_tds_fwd exposes the word; the host re-runs the affected call on the exact fp32 kernels.
__device__ __forceinline__ float amax4(float a, const f32x4& v) {
    return fmaxf(fmaxf(a, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ void note_range(float amax, int* flag) {
    if (flag && amax > 65504.f) atomicOr(flag, 1);
}

// The same split on a pair without clamps, for kernels that run under the fp16-range guard (an out-of-range value makes
// inf / NaN halves, the status word is raised and the caller discards the call's result): packed conversions and packed
// fp32 arithmetic, ~4 VALU instructions per pair instead of 14.
typedef float f32x2p __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_f16x3_pair(float a, float b, f16x2p& hi, f16x2p& lo) {
    const f32x2p x = {a, b};
    hi = __builtin_convertvector(x, f16x2p);
    const f32x2p hf = __builtin_convertvector(hi, f32x2p);
    lo = __builtin_convertvector((x - hf) * 2048.f, f16x2p);
}

// Process-wide behaviour switches (tal_set_option / tal_get_option, include/tal_asrd.h).  Nothing in the library reads the
// environment: which kernels a caller of the C ABI gets depends on its arguments and on these explicit calls only.
enum Option {
    OPT_TDS_EXACT_F32 = 0,        // every layer of tal_tds_fwd / the head arg-max on the exact fp32-input kernels
    OPT_TDS_FP32_ACTIVATIONS,     // fp16x3 layers but fp32 activations between the kernels of a stage (round-1 data flow)
    OPT_GCONV_FUSE_SPLIT,         // fp32-activation flow: the TDSBlock conv also stores the split form (instead of the split pass)
    OPT_GCONV_C1_GENERIC,         // first resize conv (1 -> 10 channels per group) on the generic tiled kernel
    OPT_HEAD_NO_ASTATIONARY,      // speaker-logit arg-max through the dense layer's fused arg-max epilogue
    OPT_GEMM_GLOBAL_LOADS,        // dense layers: 64-bit-address LDS-DMA loads instead of the buffer form
    OPT_GEMM_NO_SPLITK4,          // short problems (M <= 512) on the register-staged tile instead of the split-K-in-workgroup kernel
    OPT_GEMM_NO_GLDS,             // large problems on the register-staged kernel instead of the LDS-DMA kernel
    OPT_GEMM_NO_SPLITK_TAIL,      // no K slices for the tiles of the last partial scheduling round
    OPT_GEMM_NO_W64,              // fp16x3 layers on the 128 x 160 kernel (2 workgroups per CU) instead of the 256 x 160 one (1 wave per SIMD)
    OPT_LOGMEL_NO_FOLD,           // plans built afterwards use the direct 400-term DFT (no symmetric-window folding)
    OPT_DECODE_NO_SMALL,          // decoder layers on the batched-GEMM path even for a decode step
    OPT_DECODE_SMALL_ROWS,        // largest prefix (rows) the latency-oriented decoder layer takes (default 256)
    OPT_GEMM_NO_ROW_SPLIT,        // never split a one-round-plus-remainder fp16x3 launch into two launches by rows
    OPT_GEMM_NO_N96,              // never the 128 x 96 tiles for mid-sized fp16x3 launches (N % 96 == 0)
    OPT_GEMM_S64_BELOW,           // fp16x3 relu / residual layers take 64 x 80 tiles while those number at most this many per CU (default 2; 0: never)
    OPT_GCONV_SHORT_BELOW,        // grouped convs take 64-step tiles while the long tiles give a CU fewer workgroups than this (default 4)
    OPT_GCONV_NO_SHIFT18,         // 18-channel TDSBlock conv (split form in / out) on the two-M-tile kernel instead of the time-shift-packed one
    OPT_GCONV_GRID_XYZ,           // matrix-core grouped convs on the plain (time tile, group block, item) grid instead of the XCD-aware 1-D order
    OPT_GEMM_W64_STAGGER,         // 256 x 160 dense launches: odd first-round workgroups start this many percent of a tile's duration late (default 0)
    OPT_GEMM_S64_ORDER,           // 64 x 80 dense launches: tile order 0 = by shape (n-major when M < N), 1 = m-major, 2 = n-major
    OPT_DECODE_WIDE_GEMM,         // merged decode steps: 4 column blocks per workgroup in the skinny dense layers (0 = when the launch has >= 2 workgroups per CU, 1 = never, 2 = always)
    OPT_GEMM_S64_ROWS,            // short-input dense launches: 0 = 32-row tiles (two waves) while they number at most one per CU, 1 = always 64 rows, 2 = always 32
    OPT_DECODE_PERSIST,           // decode steps as ONE launch (csrc/decode_persist.hip) where the step's shapes allow: 0 = launch chain, 1 = one launch
    OPT_DECODE_PERSIST_WGS,       // workgroups per session of the one-launch step (default 32)
    OPT_LOGMEL_MFMA,              // log-mel as the float64 matrix-core DFT of rounds 1-4 instead of the fast transform on the vector ALU
    OPT_GRU_UNFUSED,              // GRU cell as two dense launches + a gate kernel (rounds 1-5) instead of the one-launch step
    OPT_DECODE_NO_FOLD,           // latency-oriented decoder layer on its eight launches even when the folded weights are there (six)
    OPT_GCONV_C1_FUSE,            // first resize conv computed inside the first TDSBlock conv's launch (round 6: bit-identical, no faster -- off)
    OPT_DECODE_FOLD_ROWS,         // the folded decoder layer is taken up to this many rows (prefix tokens) per problem (default 64)
    OPT_GCONV_LONG_TT,            // long inputs, 10- / 14-channel TDSBlock convs: 0 = 256- / 128-step tiles (default), 256 / 128 = that length for both
    OPT_COUNT
};
int opt(Option o);
int device_cus();      // compute units of the current device

// XCD-aware bijective remap of a 1-D grid: block b runs on XCD b % 8 (observed placement, a speed assumption only); XCD x walks a
// CONTIGUOUS range of logical blocks, so blocks that are neighbours in the logical order share that XCD's L2
__device__ __forceinline__ unsigned xcd_logical_block(unsigned nb, unsigned bid) {
    const unsigned xcd = bid & 7u, q = nb >> 3, r = nb & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// wave-uniform wave index inside the workgroup, provably uniform to the compiler
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Optional per-launch HIP-event timing (tal_prof_*): bench.py brackets every launch of a
// kernel class with two events on the launch stream and reads the average duration back.
enum ProfClass { PROF_GEMM = 0, PROF_GCONV_RES = 1, PROF_GCONV_S2 = 2, PROF_LOGMEL = 3, PROF_OTHER = 4, PROF_NCLASS = 5 };
struct ProfScope {
    int slot;
    hipStream_t s;
    ProfScope(int cls, double work, hipStream_t stream);
    ~ProfScope();
};

// Y[z] = epilogue(A[z] . W[z]^T + bias) for z = z1 * nb2 + z2 (all strides in floats).
struct GemmArgs {
    const float* A;
    const float* W;
    const float* bias;
    const float* res;
    float* Y;
    int64_t M;
    int N, K;
    int64_t lda, ldw, ldy, ldres;
    int nb2;
    int64_t a_s1, a_s2, w_s1, w_s2, y_s1, y_s2, r_s1, r_s2, bias_s2;
    float alpha;
    int scale_cols;  // mode 3: alpha applies to columns < scale_cols only (0 = all; must be a multiple of 4)
    int tiles_n;  // filled by launch_gemm
    unsigned tiles_n_magic;   // 2^32 / tiles_n + 1: tile / tiles_n == umulhi(tile, magic) for tile * tiles_n < 2^32 (launch_gemm)
    // mode 4 (row arg-max instead of a store): per row and per 32*NSUB-column wave slice the best
    // (value, column) goes to part_val / part_idx [M, part_ld]; Y is not written.
    float* part_val;
    int32_t* part_idx;
    int part_ld;
    // split-K tail (stream-K-lite, filled by launch_gemm): the tiles of the partial last scheduling
    // round are cut into `split` K slices that write raw accumulators to `splitk_ws`; a fix-up kernel
    // adds the slices in order and applies the epilogue.  tile_base offsets the logical tile index.
    int tile_base, split, tail_tiles;
    float* splitk_ws;
    size_t splitk_ws_bytes;
    // fp16x3 form (dense layers of the TDS blocks): A and W are hi/lo fp16 splits of fp32 data in the byte geometry of
    // fp32 rows (per row and 32-wide K block: 32 hi halves, then 32 lo halves scaled by 2^11); out_split: write Y in
    // the same split form (it is the next layer's A) instead of fp32.  Needs K % 32 == 0 (and N % 32 == 0 for out_split).
    int f16x3, out_split;
    int res_split;     // mode 2: `res` is in the split form (hi / lo halves, fp32-row geometry); rebuilt as hi + lo * 2^-11
    int* range_flag;   // out_split: raised when an output value lies outside the fp16 range (may be NULL)
    // gemm_w64_kernel: the odd workgroups among the first `stagger_blocks` (the launch's first round, one per CU) start
    // `stagger_ticks` (100 MHz) late, so that half of the chip is in its K loop while the other half writes its tiles
    int stagger_ticks, stagger_blocks;
    // gemm_s64_kernel: tile order.  n_major = 1 walks the M tiles of one N tile first (an XCD's contiguous range of tiles then
    // shares a few W panels and reads all of X) -- for inputs with fewer rows than the layer has columns, where W is the
    // larger operand; 0 walks the N tiles of one M tile first (X panel shared, every XCD reads all of W).
    int n_major, tiles_m;
    unsigned tiles_m_magic;
};
// mode 0: acc+b | 1: relu(acc+b) | 2: res + alpha*(acc+b) | 3: alpha*(acc+b) | 4: row arg-max partials of acc+b
int launch_gemm(GemmArgs g, int mode, int nbatch, hipStream_t s);
// fp16x3 layer on 256 x 160 tiles, one wave per SIMD (gemm_w64.hip); splitk: grid = [whole tiles | K slices of the tail tiles]
void launch_gemm_w64(const GemmArgs& g, int mode, bool splitk, dim3 grid, hipStream_t s);
// fp16x3 relu / residual layer on 64 x 80 tiles for short inputs (gemm_s64.hip): whole tiles only, no K slices
bool gemm_s64_ok(const GemmArgs& g, int mode);
void launch_gemm_s64(GemmArgs g, int mode, hipStream_t s);
int gemm_mode4_partials(int64_t M, int N);

// internal launchers shared between translation units
int launch_linear(const float* x, const float* w, const float* b, const float* res, float alpha, int mode,
                  int64_t M, int N, int K, float* y, hipStream_t s);
// same, with a scratch buffer that enables the split-K treatment of the last partial round
int launch_linear_ws(const float* x, const float* w, const float* b, const float* res, float alpha, int mode,
                     int64_t M, int N, int K, float* y, float* ws, size_t ws_bytes, hipStream_t s);
size_t gemm_splitk_ws_bytes();
// fp32 [rows, K] -> hi/lo fp16 split in fp32-row geometry (K % 32 == 0)
int launch_split_f16x3(const float* x, void* out, int64_t rows, int K, hipStream_t s, int* range_flag = nullptr);
// dense layer on pre-split operands (see GemmArgs::f16x3); ws as launch_linear_ws
int launch_linear_f16x3(const void* xs, const void* wsplit, const float* b, const float* res, float alpha, int mode, int64_t M,
                        int N, int K, void* y, int out_split, float* ws, size_t ws_bytes, hipStream_t s, int* range_flag = nullptr,
                        int res_split = 0);
// y_split != NULL: the output goes out in the hi / lo split form ONLY (first resize conv, 1 -> 10 channels per group)
// in_mean != NULL: x is a log-mel before its mean subtraction; the scalar in_mean[0] is folded into the conv's bias (1 -> 10 only)
int launch_gconv_s2(const float* x, const float* wp, const float* bias, int B, int64_t T_in, int C_in, int C_out,
                    int groups, float* y, hipStream_t s, void* y_split = nullptr, int* range_flag = nullptr, const float* in_mean = nullptr);
bool gconv_s2_can_split(int C_in, int C_out, int groups, const float* x);
bool gconv_s2_can_fold_mean(int C_in, int C_out, int groups, const float* x);
int launch_gconv_res(const float* x, const float* wp, const float* bias, float alpha, int B, int64_t T, int C,
                     int groups, float* y, hipStream_t s);
size_t gconv_f16x3_weight_bytes(int C_in, int C_out, int groups, int stride);
bool gconv_f16x3_fits(int64_t T, int C);
// first resize conv (1 -> 10 channels per group) + first TDSBlock conv of the stage in one launch (gconv_mfma_kernel<.., FROMC1>)
bool gconv_c1_res_fusable(int C_in, int C_out, int groups, const float* mel);
int launch_gconv_c1_res_f16x3(const float* mel, const float* w1, const float* b1, const float* in_mean, const void* w_frag, const float* bias,
                              float alpha, int B, int64_t T_mel, int groups, void* y_split, hipStream_t s, int* range_flag);
int launch_pack_gconv_f16x3(const float* w_ref, void* w_frag, int C_in, int C_out, int groups, int stride, hipStream_t s);
int launch_gconv_s2_f16x3(const float* x, const void* w_frag, const float* bias, int B, int64_t T_in, int C_in, int C_out, int groups,
                          float* y, hipStream_t s, int* range_flag = nullptr, bool x_split = false, void* y_split = nullptr);
// x_split: x is in the hi / lo split form; y == NULL: only the split form of the output (y_split) is written
int launch_gconv_res_f16x3(const float* x, const void* w_frag, const float* bias, float alpha, int B, int64_t T, int C, int groups,
                           float* y, void* y_split, hipStream_t s, int* range_flag = nullptr, bool x_split = false);
int launch_argmax_rows(const float* x, int64_t M, int N, int32_t* ids, hipStream_t s);
// A-stationary speaker-logit arg-max (csrc/head.hip): partials [M, head_argmax_partials(M, S)] for argmax_partials_kernel
bool head_argmax_applicable(int64_t M, int S, int E);
int head_argmax_partials(int64_t M, int S);
int launch_head_argmax(const float* feat, const float* w, const void* w_split, const float* b, int64_t M, int S, float* part_val,
                       int32_t* part_idx, hipStream_t s);

// ---- latency-oriented kernels of the decode step (csrc/decode_small.hip) ----
struct SkinnyArgs {
    const float* A;
    const float* W;
    const float* bias;
    const float* res;
    float* Y;
    int M, N, K;
    int64_t lda, ldw, ldy, ldres;
    float alpha;
    int scale_cols;   // mode 3: alpha applies to columns < scale_cols only (0 = all)
    // columns >= vt_begin (a multiple of 16) go to Yt[b][col - vt_begin][u] for row m = b * U + u
    float* Yt;
    int vt_begin, U;
    int64_t ldt, vt_bs;
    // split K over `ksplit` workgroups per tile (deep layers: one CU pulls only ~25-60 GB/s): raw partial tiles go to
    // sk_part [ksplit][tiles][512], the last workgroup to arrive at a tile (ticket on sk_tickets[tile], left at zero) adds
    // them in split order and applies the epilogue -> deterministic.
    int ksplit;
    float* sk_part;
    unsigned* sk_tickets;
    // A in TWO column segments (round 6, the folded decoder layers): k < K1 comes from A (pitch lda), k >= K1 from A2 (pitch lda2)
    // at column k - K1; a wave's K slice never straddles K1 (K1 is a multiple of every slice width).  A2 == NULL: one segment.
    const float* A2;
    int64_t lda2;
    int K1;
    int relu_begin;   // mode 1: the relu applies to columns >= relu_begin only (0 = all)
    // columns < k1_cols (a multiple of 64) contract over the FIRST segment only and get A2[m][col] added (the ReZero skip path of the
    // folded layers: y = x + (ctx . W'^T + b')): the waves of the second segment neither load nor multiply for those column blocks
    int k1_cols;
};

struct AttnArgs {
    const float* q;      // [B][U][..] rows of pitch ldq (already scaled by hd^-0.5), head h at column h * hd
    const float* k;      // [B][S][..] rows of pitch ldk
    const float* vt;     // [B][E][ldvt]: V^T without bias (added after P.V: softmax rows sum to one)
    const float* vbias;  // [E] or NULL
    const float* mask;   // additive [U][S] or NULL
    const uint8_t* kpm;  // [B][S], non-zero = ignore, or NULL
    float* ctx;          // [B][U][..] rows of pitch ldc
    float* probs;        // per-head probabilities [B][H][U - prob_row0][S] of the rows >= prob_row0, or NULL
    int64_t ldq, q_bs, ldk, k_bs, ldvt, vt_bs, ldc, c_bs;
    int U, S, H, prob_row0;
};

// several independent problems in one launch (the decode steps of several sessions): argument structs by value
constexpr int TAL_GROUP_MAX = 16;     // (16 argument sets of <= 152 bytes stay well inside the 4 KB kernel argument segment)
template <typename T>
struct ArgPack {
    T a[TAL_GROUP_MAX];
    int n;
};

// ---- the decode step as one launch (csrc/decode_persist.hip) ----
constexpr int TAL_GREEDY_TICKETS = 256;      // words of a session's ticket block (tal_greedy_ctx.tickets); the last one belongs to the pick
// words of the ticket block the one-launch step uses for itself (the FFN's split-K tickets start at 64: a step joins only while
// they end below PS_BAR): arrival counter of the phase barriers, exit counter, error word (raised when a wait gives up)
constexpr int PS_BAR = 252, PS_DONE = 253, PS_ERR = 254;
constexpr int TAL_PS_MAX_SESS = 8, TAL_PS_MAX_LAYERS = 6;
struct PsModel {
    tal_decoder_layer_w layer[TAL_PS_MAX_LAYERS];
    int n_layers, E, H, FF, V, K0;      // K0: embedding width (E0, or E without the factorised embedding)
    float qscale;                       // head_dim^-0.5, computed once on the host (as the launch chain does)
    const float *emb, *proj, *proj_t, *pe;
};
struct PsSession {
    const int64_t* tokens;              // first token of the live prefix
    int64_t* token_out;                 // where the picked token is appended
    int U, S;
    float *h0, *h1, *qkv, *vt, *ctx, *x1, *x2, *ff, *probs, *sk_part, *pick_part, *out;
    unsigned* tickets;
    const float* k_cache[TAL_PS_MAX_LAYERS];
    const float* vt_cache[TAL_PS_MAX_LAYERS];
    const uint8_t* kpm;
    int64_t k_pitch;
    unsigned host_seq;
};
struct PsArgs {
    PsModel m;
    PsSession s[TAL_PS_MAX_SESS];
    int n, G;
};
static_assert(sizeof(PsArgs) <= 3800, "the one-launch step's arguments travel by value in the kernel argument segment");
int launch_greedy_persist(const PsArgs& a, hipStream_t s);

bool skinny_gemm_applicable(const SkinnyArgs& g);
int launch_skinny_gemm(const SkinnyArgs& g, int mode, hipStream_t s);
bool attn_small_applicable(int U, int S, int hd);
int launch_attn_small(const AttnArgs& g, int B, int hd, hipStream_t s);
size_t attn_split_scratch_floats(int B, int U, int S, int H, int hd);
int attn_split_tickets(int B, int U, int H);
int launch_attn_split(const AttnArgs& g, int B, int hd, float* scratch, unsigned* tickets, hipStream_t s);
int launch_head_average(const float* probs, float* avg, int B, int H, int U, int S, hipStream_t s);
// the multi forms: G <= TAL_GROUP_MAX problems (one batch item each for the attention kernels) in one launch, every problem's
// results bit-identical to its own launch
int launch_skinny_gemm_multi(const SkinnyArgs* g, int G, int mode, hipStream_t s);
int launch_attn_small_multi(const AttnArgs* g, int G, int hd, hipStream_t s);
int launch_attn_split_multi(const AttnArgs* g, float* const* scratch, unsigned* const* tickets, int G, int hd, hipStream_t s);

}  // namespace tal
