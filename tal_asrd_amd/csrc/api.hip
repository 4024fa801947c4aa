// C-ABI entry points (include/tal_asrd.h) that compose the kernels: error string, dense
// layer, TDS encoder driver, diarization head.
#include <string.h>

#include <atomic>

#include "common.h"

namespace tal {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- behaviour switches (tal_set_option) ---------------------------------------------------
static const char* const g_opt_names[OPT_COUNT] = {
    "tds_exact_f32", "tds_fp32_activations", "gconv_fuse_split", "gconv_c1_generic", "head_no_astationary", "gemm_global_loads",
    "gemm_no_splitk4", "gemm_no_glds", "gemm_no_splitk_tail", "gemm_no_w64", "logmel_no_fold", "decode_no_small", "decode_small_rows", "gemm_no_row_split", "gemm_no_n96", "gemm_s64_below", "gconv_short_below", "gconv_no_shift18", "gconv_grid_xyz", "gemm_w64_stagger", "gemm_s64_order", "decode_wide_gemm", "gemm_s64_rows", "decode_persist", "decode_persist_wgs", "logmel_mfma", "gru_unfused", "decode_no_fold", "gconv_c1_fuse", "decode_fold_rows", "gconv_long_tt"};
static std::atomic<int> g_opt[OPT_COUNT] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {256}, {0}, {0}, {2}, {4}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {32}, {0}, {0}, {0}, {0}, {64}, {0}};
int opt(Option o) { return g_opt[o].load(std::memory_order_relaxed); }

int device_cus() {
    // per device (a process may drive several), filled on first use; racing first uses write the same value
    static std::atomic<int> table[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = table[dev].load(std::memory_order_relaxed);
    if (!cus) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
        table[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

// ---- per-launch event timing ------------------------------------------------------------
// (a process-wide switch for benchmarking; slots are handed out atomically so that launches from several host threads
//  cannot corrupt the table -- with the switch off, the default, nothing here is touched)
static const int PROF_MAX = 8192;
static std::atomic<bool> g_prof_on{false};
static std::atomic<int> g_prof_n{0};
static hipEvent_t g_prof_ev[PROF_MAX][2];
static std::atomic<int> g_prof_have[PROF_MAX];       // 1: the slot's event pair exists
static int g_prof_cls[PROF_MAX];                     // -1: slot taken but not timed
static double g_prof_work[PROF_MAX];

ProfScope::ProfScope(int cls, double work, hipStream_t stream) : slot(-1), s(stream) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    const int mine = g_prof_n.fetch_add(1, std::memory_order_relaxed);
    if (mine >= PROF_MAX) return;
    g_prof_cls[mine] = -1;
    if (!g_prof_have[mine].load(std::memory_order_acquire)) {
        // (a slot index is owned by one launch at a time, so nobody else creates this pair)
        if (hipEventCreate(&g_prof_ev[mine][0]) != hipSuccess || hipEventCreate(&g_prof_ev[mine][1]) != hipSuccess)
            return;              // (measurement hook only: a launch without events is simply not timed)
        g_prof_have[mine].store(1, std::memory_order_release);
    }
    slot = mine;
    g_prof_cls[slot] = cls;
    g_prof_work[slot] = work;
    (void)hipEventRecord(g_prof_ev[slot][0], s);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof_ev[slot][1], s);
}

// Row-wise argmax, one wave per row; ties resolve to the lowest index (torch.argmax).
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int64_t M, int N,
                                                         int32_t* __restrict__ ids) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * N;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < N; i += 64) {
        const float v = xr[i];
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
    if (lane == 0) ids[row] = bi == 0x7fffffff ? 0 : bi;
}

// final reduction of the fused arg-max partials (ascending column slices; strict '>' keeps the lowest index)
// One wave per row: lanes take partials p = lane, lane + 64, ... (a thread per row walking P partials is a chain of P
// dependent loads: 46 us for 188 partials); the partials are in ascending column order, so the lowest column wins ties.
__global__ __launch_bounds__(256) void argmax_partials_kernel(const float* __restrict__ pv,
                                                             const int32_t* __restrict__ pi, int64_t M, int P, int ld,
                                                             int32_t* __restrict__ ids) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int p = lane; p < P; p += 64) {
        const float v = pv[row * ld + p];
        const int idx = pi[row * ld + p];
        if (v > best || (v == best && idx < bi)) {
            best = v;
            bi = idx;
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
    if (lane == 0) ids[row] = bi == 0x7fffffff ? 0 : bi;
}

// one workgroup per row (few long rows: the arg-max of a decode step's [1, V] logits)
__global__ __launch_bounds__(256) void argmax_row_block_kernel(const float* __restrict__ x, int N,
                                                              int32_t* __restrict__ ids) {
    __shared__ float bv[4];
    __shared__ int bix[4];
    const float* xr = x + (int64_t)blockIdx.x * N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float v = xr[i];
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
        bv[w] = best;
        bix[w] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q)
            if (bv[q] > best || (bv[q] == best && bix[q] < bi)) {
                best = bv[q];
                bi = bix[q];
            }
        ids[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;
    }
}

int launch_argmax_rows(const float* x, int64_t M, int N, int32_t* ids, hipStream_t s) {
    TAL_CHECK_ARG(x && ids && N > 0 && M >= 0, "tal_argmax_rows: bad argument");
    if (M == 0) return TAL_OK;
    ProfScope prof(PROF_OTHER, (double)M * N * 4.0, s);
    if (M <= 512 && N >= 2048)
        hipLaunchKernelGGL(argmax_row_block_kernel, dim3((unsigned)M), dim3(256), 0, s, x, N, ids);
    else
        hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, s, x, M, N, ids);
    TAL_CHECK_LAUNCH("tal_argmax_rows");
    return TAL_OK;
}

static int64_t conv_out_len(int64_t t) { return t < 21 ? 0 : (t - 21) / 2 + 1; }

}  // namespace tal

using namespace tal;

extern "C" int tal_version(void) { return 500; /* 0.5.0: tal_greedy_ctx grew (needs_reset, pick_bias, no_fold), tal_decoder_layer_w grew (fold_*), tal_greedy_step_poll takes a non-const context, word 1 of the TDS status block = output form, options gru_unfused / decode_no_fold / decode_fold_rows / gconv_c1_fuse; 0.4.3: log-mel as a fast transform (option logmel_mfma: the matrix form), the plan grew; 0.4.2: TAL_TDS_OUT_SPLIT, tal_tds_out_split, tal_sd_head_split_fwd; 0.4.1: tal_tds_premean_fwd / tal_tds_premean_ok; 0.4.0: tal_greedy_ctx grew (k_pitch, episode-wide K | V table), merged decode steps, tal_unaligned_*, tal_logmel_f16_fwd; 0.3.0: tal_set_option */ }

// Host-side helper of the decode loop (no device work): ngram_repeat_mask(row, n).sum() of tal/asr/util.py:5-17 -- the number
// of positions covered by an n-gram that already occurred earlier in the row; like the reference, n-gram starts run to
// len - n - 1.  Called once per generated token (system.py:418-421), where the Python set-of-tuples form costs more than
// the GPU step it sits beside.
extern "C" int64_t tal_ngram_repeat_count(const int64_t* row, int64_t len, int n) {
    if (!row || n <= 0 || len - n <= 0) return 0;
    const int64_t starts = len - n;
    int64_t covered_until = 0, count = 0;
    for (int64_t j = 0; j < starts; ++j) {
        bool seen = false;
        for (int64_t i = 0; i < j && !seen; ++i) {
            if (row[i] != row[j]) continue;
            bool eq = true;
            for (int k = 1; k < n && eq; ++k) eq = row[i + k] == row[j + k];
            seen = eq;
        }
        if (seen) {
            const int64_t lo = j > covered_until ? j : covered_until;
            if (j + n > lo) count += j + n - lo;
            covered_until = j + n;
        }
    }
    return count;
}

extern "C" const char* tal_last_error(void) { return g_err; }

static int find_option(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, g_opt_names[i]) == 0) return i;
    return -1;
}

extern "C" int tal_set_option(const char* name, int value) {
    const int i = find_option(name);
    TAL_CHECK_ARG(i >= 0, "tal_set_option: unknown option '%s'", name ? name : "(null)");
    TAL_CHECK_ARG(i != OPT_DECODE_SMALL_ROWS || value >= 0, "tal_set_option: decode_small_rows must be >= 0");
    g_opt[i].store(value, std::memory_order_relaxed);
    return TAL_OK;
}

extern "C" int tal_get_option(const char* name, int* value) {
    const int i = find_option(name);
    TAL_CHECK_ARG(i >= 0 && value, "tal_get_option: unknown option '%s'", name ? name : "(null)");
    *value = g_opt[i].load(std::memory_order_relaxed);
    return TAL_OK;
}

extern "C" const char* tal_option_name(int index) { return index >= 0 && index < OPT_COUNT ? g_opt_names[index] : nullptr; }

extern "C" int tal_prof_enable(int on) {
    g_prof_on = on != 0;
    return TAL_OK;
}

extern "C" int tal_prof_reset(void) {
    g_prof_n = 0;
    return TAL_OK;
}

extern "C" int tal_prof_collect(int cls, double* total_ms, int64_t* launches, double* total_work) {
    TAL_CHECK_ARG(cls >= 0 && cls < PROF_NCLASS && total_ms && launches && total_work, "tal_prof_collect: bad argument");
    double ms = 0.0, work = 0.0;
    int64_t n = 0;
    const int n_used = g_prof_n.load() < PROF_MAX ? g_prof_n.load() : PROF_MAX;
    for (int i = 0; i < n_used; ++i) {
        if (g_prof_cls[i] != cls) continue;
        if (hipEventSynchronize(g_prof_ev[i][1]) != hipSuccess) {
            set_error("tal_prof_collect: event sync failed");
            return TAL_EHIP;
        }
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof_ev[i][0], g_prof_ev[i][1]) != hipSuccess) {
            set_error("tal_prof_collect: cannot read the event pair of launch %d", i);
            return TAL_EHIP;
        }
        ms += t;
        work += g_prof_work[i];
        ++n;
    }
    *total_ms = ms;
    *launches = n;
    *total_work = work;
    return TAL_OK;
}

extern "C" int tal_linear_fwd(const float* x, const float* w, const float* b, const float* res, float alpha, int mode,
                              int64_t M, int N, int K, float* y, void* stream) {
    return launch_linear(x, w, b, res, alpha, mode, M, N, K, y, (hipStream_t)stream);
}

extern "C" size_t tal_linear_workspace_bytes(int64_t M, int N, int K) {
    return (M > 512 && K % 32 == 0 && K >= 256 && N > 0) ? gemm_splitk_ws_bytes() : 0;
}

extern "C" int tal_linear_ws_fwd(const float* x, const float* w, const float* b, const float* res, float alpha, int mode,
                                 int64_t M, int N, int K, float* y, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    TAL_CHECK_ARG(workspace || workspace_bytes == 0, "tal_linear_ws_fwd: null workspace with %zu bytes", workspace_bytes);
    return launch_linear_ws(x, w, b, res, alpha, mode, M, N, K, y, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int tal_gconv_s2_fwd(const float* x, const float* w_packed, const float* bias, int B, int64_t T_in,
                                int C_in, int C_out, int groups, float* y, void* stream) {
    return launch_gconv_s2(x, w_packed, bias, B, T_in, C_in, C_out, groups, y, (hipStream_t)stream);
}

extern "C" int tal_gconv_res_fwd(const float* x, const float* w_packed, const float* bias, float alpha, int B,
                                 int64_t T, int C, int groups, float* y, void* stream) {
    return launch_gconv_res(x, w_packed, bias, alpha, B, T, C, groups, y, (hipStream_t)stream);
}

extern "C" size_t tal_gconv_f16x3_weight_bytes(int C_in, int C_out, int groups, int stride) {
    return gconv_f16x3_weight_bytes(C_in, C_out, groups, stride);
}

extern "C" int tal_pack_gconv_f16x3_weight(const float* w_ref, void* w_frag, int C_in, int C_out, int groups, int stride, void* stream) {
    return launch_pack_gconv_f16x3(w_ref, w_frag, C_in, C_out, groups, stride, (hipStream_t)stream);
}

extern "C" int tal_gconv_res_f16x3_fwd(const float* x, const void* w_frag, const float* bias, float alpha, int B,
                                       int64_t T, int C, int groups, float* y, void* y_split, void* stream) {
    return launch_gconv_res_f16x3(x, w_frag, bias, alpha, B, T, C, groups, y, y_split, (hipStream_t)stream);
}

extern "C" int tal_gconv_s2_f16x3_fwd(const float* x, const void* w_frag, const float* bias, int B, int64_t T_in, int C_in,
                                      int C_out, int groups, float* y, void* stream) {
    return launch_gconv_s2_f16x3(x, w_frag, bias, B, T_in, C_in, C_out, groups, y, (hipStream_t)stream);
}

extern "C" int tal_gconv_res_split_fwd(const void* x_split, const void* w_frag, const float* bias, float alpha, int B, int64_t T,
                                       int C, int groups, void* y_split, void* stream) {
    TAL_CHECK_ARG(C % 32 == 0, "tal_gconv_res_split_fwd: the split form needs C %% 32 == 0 (C=%d)", C);
    return launch_gconv_res_f16x3(reinterpret_cast<const float*>(x_split), w_frag, bias, alpha, B, T, C, groups, nullptr, y_split,
                                  (hipStream_t)stream, nullptr, true);
}

extern "C" int tal_gconv_s2_split_fwd(const void* x, int x_is_split, const void* w_frag, const float* bias, int B, int64_t T_in,
                                      int C_in, int C_out, int groups, void* y_split, void* stream) {
    TAL_CHECK_ARG(C_out % 32 == 0 && (!x_is_split || C_in % 32 == 0), "tal_gconv_s2_split_fwd: the split form needs channels %% 32 == 0");
    return launch_gconv_s2_f16x3(reinterpret_cast<const float*>(x), w_frag, bias, B, T_in, C_in, C_out, groups, nullptr, (hipStream_t)stream,
                                 nullptr, x_is_split != 0, y_split);
}

extern "C" int tal_argmax_rows(const float* x, int64_t M, int N, int32_t* ids, void* stream) {
    return launch_argmax_rows(x, M, N, ids, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// TDS encoder driver
// ---------------------------------------------------------------------------------------
static int check_desc(const tal_tds_desc* d) {
    TAL_CHECK_ARG(d, "tal_tds: null descriptor");
    TAL_CHECK_ARG(d->n_stages >= 1 && d->n_stages <= TAL_MAX_STAGES, "tal_tds: n_stages=%d", d->n_stages);
    TAL_CHECK_ARG(d->groups > 0, "tal_tds: groups=%d", d->groups);
    for (int i = 0; i <= d->n_stages; ++i)
        TAL_CHECK_ARG(d->channels[i] > 0 && d->channels[i] % d->groups == 0, "tal_tds: channels[%d]=%d", i, d->channels[i]);
    for (int i = 0; i < d->n_stages; ++i) {
        TAL_CHECK_ARG(d->depths[i] >= 0 && d->depths[i] <= TAL_MAX_DEPTH, "tal_tds: depths[%d]=%d", i, d->depths[i]);
        TAL_CHECK_ARG(d->channels[i + 1] % 4 == 0 || d->depths[i] == 0, "tal_tds: channels[%d]=%d must be a multiple of 4 for the pointwise layers", i + 1, d->channels[i + 1]);
    }
    return TAL_OK;
}

extern "C" int64_t tal_tds_out_len(const tal_tds_desc* d, int64_t T) {
    if (!d) return -1;
    for (int i = 0; i < d->n_stages; ++i) T = conv_out_len(T);
    return T;
}

static size_t tds_buf_floats(const tal_tds_desc* d, int B, int64_t T) {
    size_t mx = 0;
    for (int i = 0; i < d->n_stages; ++i) {
        T = conv_out_len(T);
        const size_t n = (size_t)B * (size_t)T * (size_t)d->channels[i + 1];
        if (n > mx) mx = n;
    }
    return (mx + 63) & ~(size_t)63;
}

// workspace = 4 rotating activation buffers | split-K scratch | 64-byte status block (word 0: fp16-range flag)
// Clears the 64-byte status block of a call.  A kernel, not hipMemsetAsync: a memset node captured into a HIP graph replays
// with a stale fill pattern on ROCm 7.2 (measured: the block came back holding two pointers, profiles/r4_short_clip_graph.txt),
// which read as "out of fp16 range" on every replay of a captured SD call (tests/test_gpu_parity.py).
// Word 1 receives the FORM the call writes y in (1: hi / lo split, TAL_TDS_OUT_SPLIT honoured; 0: fp32), decided by the call
// itself from its real arguments and the options in force when it runs.
__global__ void clear_status_kernel(int* __restrict__ status, int y_form = 0) { status[threadIdx.x] = threadIdx.x == 1 ? y_form : 0; }

extern "C" size_t tal_tds_status_offset(const tal_tds_desc* d, int B, int64_t T) {
    if (!d || B <= 0 || T <= 0) return 0;
    return 4 * tds_buf_floats(d, B, T) * sizeof(float) + gemm_splitk_ws_bytes();
}

extern "C" size_t tal_tds_workspace_bytes(const tal_tds_desc* d, int B, int64_t T) {
    if (!d || B <= 0 || T <= 0) return 0;
    return tal_tds_status_offset(d, B, T) + 64;
}

static int tds_fwd_impl(const tal_tds_desc* d, const float* x, const float* x_mean, int B, int64_t T, float* y, void* workspace,
                        size_t workspace_bytes, void* stream);
static bool tds_last_stage_allsplit(const tal_tds_desc* d, int B, int64_t T, const float* x0);

// 1: a call with TAL_TDS_OUT_SPLIT in d->flags leaves y in the hi / lo split form (the last stage runs all-split: long inputs on
// the fp16x3 kernels); 0: y is fp32 as always (short inputs, odd widths, the exact mode)
extern "C" int tal_tds_out_split(const tal_tds_desc* d, int B, int64_t T) {
    if (check_desc(d) || B <= 0 || tal_tds_out_len(d, T) <= 0 || !(d->flags & TAL_TDS_OUT_SPLIT)) return 0;
    // (a PREDICTION for a 16-byte aligned x under the options in force now; the call records what it did in its status block)
    return tds_last_stage_allsplit(d, B, T, reinterpret_cast<const float*>(static_cast<uintptr_t>(16))) ? 1 : 0;
}

extern "C" int tal_tds_fwd(const tal_tds_desc* d, const float* x, int B, int64_t T, float* y, void* workspace,
                           size_t workspace_bytes, void* stream) {
    return tds_fwd_impl(d, x, nullptr, B, T, y, workspace, workspace_bytes, stream);
}

extern "C" int tal_tds_premean_ok(const tal_tds_desc* d, const float* x) {
    if (check_desc(d) || d->n_stages < 1) return 0;
    return gconv_s2_can_fold_mean(d->channels[0], d->channels[1], d->groups, x) ? 1 : 0;
}

extern "C" int tal_tds_premean_fwd(const tal_tds_desc* d, const float* x, const float* x_mean, int B, int64_t T, float* y, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    TAL_CHECK_ARG(x_mean, "tal_tds_premean_fwd: null mean pointer");
    TAL_CHECK_ARG(tal_tds_premean_ok(d, x), "tal_tds_premean_fwd: the first resize conv of this stack (%d -> %d channels, %d groups) has no mean-folding "
                  "kernel: subtract the mean (tal_subtract_scalar) and call tal_tds_fwd", d->channels[0], d->channels[1], d->groups);
    return tds_fwd_impl(d, x, x_mean, B, T, y, workspace, workspace_bytes, stream);
}

// Which form a stage of the stack runs in is decided from the descriptor and the shapes alone (tal_tds_fwd and tal_tds_out_split share it).
static bool tds_s2_mfma_ok(const tal_tds_desc* d, int B, int i, int64_t Tin, bool force_f32) {       // stride-2 resize conv of stage i on the matrix cores
    return !force_f32 && d->down_w_frag[i] && (int64_t)B * conv_out_len(Tin) > 64 &&
           gconv_f16x3_weight_bytes(d->channels[i], d->channels[i + 1], d->groups, 2) > 0 && gconv_f16x3_fits(Tin, d->channels[i]);
}
static bool tds_stage_allsplit(const tal_tds_desc* d, int B, int64_t T, int i, const float* xin, bool in_split, bool force_f32, bool no_allsplit) {
    if (force_f32 || no_allsplit || i >= d->n_stages || d->depths[i] == 0) return false;
    const int c = d->channels[i + 1], cin = d->channels[i];
    int64_t Tin = T;
    for (int q = 0; q < i; ++q) Tin = conv_out_len(Tin);
    const int64_t To = conv_out_len(Tin), M = (int64_t)B * To;
    if (M <= 128 || c % 160 != 0 || c % 32 != 0 || !gconv_f16x3_fits(To, c) || gconv_f16x3_weight_bytes(c, c, d->groups, 1) == 0) return false;
    for (int j = 0; j < d->depths[i]; ++j)
        if (!d->blocks[i][j].fc0_w_split || !d->blocks[i][j].fc3_w_split || !d->blocks[i][j].conv_w_frag) return false;
    // the resize conv must be able to write the split form: the 1 -> 10 channel kernel or a matrix-core kernel
    if (tds_s2_mfma_ok(d, B, i, Tin, force_f32)) return !in_split || cin % 32 == 0;
    return !in_split && gconv_s2_can_split(cin, c, d->groups, xin);
}
// the last stage's form, with the input form it will see (the previous stage's output is split iff that stage and the matrix-core
// resize conv between them allow it: the same chain of decisions tds_fwd_impl makes)
static bool tds_last_stage_allsplit(const tal_tds_desc* d, int B, int64_t T, const float* x0) {
    const bool force_f32 = opt(OPT_TDS_EXACT_F32) != 0 || (d->flags & TAL_TDS_EXACT_F32) != 0;
    const bool no_allsplit = opt(OPT_TDS_FP32_ACTIVATIONS) != 0;
    bool cur_split = false;
    int64_t Tc = T;
    bool allsplit = false;
    for (int i = 0; i < d->n_stages; ++i) {
        const int64_t To = conv_out_len(Tc);
        // (stage 0 reads the caller's fp32 x: its alignment matters to the 1 -> 10 kernel; later stages read workspace buffers)
        allsplit = tds_stage_allsplit(d, B, T, i, i == 0 ? x0 : reinterpret_cast<const float*>(static_cast<uintptr_t>(16)), cur_split, force_f32, no_allsplit);
        const bool last_stage = i == d->n_stages - 1;
        const bool next_split = !last_stage && d->channels[i + 1] % 32 == 0 && tds_s2_mfma_ok(d, B, i + 1, To, force_f32) &&
                                tds_stage_allsplit(d, B, T, i + 1, nullptr, true, force_f32, no_allsplit);
        cur_split = d->depths[i] > 0 && allsplit && next_split;
        Tc = To;
    }
    return allsplit && d->depths[d->n_stages - 1] > 0 && d->channels[d->n_stages] % 32 == 0;
}

static int tds_fwd_impl(const tal_tds_desc* d, const float* x, const float* x_mean, int B, int64_t T, float* y, void* workspace,
                        size_t workspace_bytes, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    TAL_CHECK_ARG(x && y && workspace, "tal_tds_fwd: null pointer");
    TAL_CHECK_ARG(B > 0 && tal_tds_out_len(d, T) > 0, "tal_tds_fwd: T=%lld too short for %d stride-2 k=21 stages", (long long)T, d->n_stages);
    if (workspace_bytes < tal_tds_workspace_bytes(d, B, T)) {
        set_error("tal_tds_fwd: workspace %zu < %zu bytes", workspace_bytes, tal_tds_workspace_bytes(d, B, T));
        return TAL_ENOMEM;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t nf = tds_buf_floats(d, B, T);
    float* buf[4];
    for (int q = 0; q < 4; ++q) buf[q] = reinterpret_cast<float*>(workspace) + q * nf;
    float* skws = reinterpret_cast<float*>(workspace) + 4 * nf;   // split-K scratch of the dense layers
    const bool force_f32 = opt(OPT_TDS_EXACT_F32) != 0 || (d->flags & TAL_TDS_EXACT_F32) != 0;
    // status word: raised by any kernel that turns an fp32 value outside the finite fp16 range into hi / lo halves
    int* range_flag = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + tal_tds_status_offset(d, B, T));
    // the form y will be written in, from the real x and the options of THIS call (tal_tds_out_split is a prediction made earlier,
    // for an aligned x): recorded in word 1 of the status block, where the caller that consumes y in the split form checks it
    const bool y_split_out = (d->flags & TAL_TDS_OUT_SPLIT) != 0 && tds_last_stage_allsplit(d, B, T, x);
    clear_status_kernel<<<1, 16, 0, s>>>(range_flag, y_split_out ? 1 : 0);
    if (hipGetLastError() != hipSuccess) {
        set_error("tal_tds_fwd: cannot clear the status word");
        return TAL_EHIP;
    }
    // four rotating buffers; `ia` = index of the buffer holding the live activations (-1: caller's x).
    // No launch ever reads and writes the same buffer (workgroups read halos of their neighbours).
    //
    // Activation format.  Long inputs run a stage with every inter-kernel activation in the hi / lo SPLIT form of the
    // fp16x3 dense layers (same bytes as fp32; the exact value is hi + lo * 2^-11, 22 mantissa bits): the resize conv writes
    // it, the TDSBlock conv reads it (no conversion arithmetic in its slab phase) and writes it, fc0 reads / writes it, fc3
    // reads it twice (operand and residual) and writes it for the next block.  No fp32 copy of an activation and no
    // separate split pass exist inside such a stage: 7 activation-sized transfers per block instead of 10.  Short inputs,
    // odd widths and the exact mode keep fp32 activations (the kernels below the `else`).
    const bool no_allsplit = opt(OPT_TDS_FP32_ACTIVATIONS) != 0;
    auto stage_len = [&](int i) { int64_t t = T; for (int q = 0; q <= i; ++q) t = conv_out_len(t); return t; };
    auto s2_mfma_ok = [&](int i, int64_t Tin) { return tds_s2_mfma_ok(d, B, i, Tin, force_f32); };
    auto stage_allsplit = [&](int i, const float* xin, bool in_split) { return tds_stage_allsplit(d, B, T, i, xin, in_split, force_f32, no_allsplit); };
    const float* cur = x;
    bool cur_split = false;
    int ia = -1;
    int64_t Tc = T;
    for (int i = 0; i < d->n_stages; ++i) {
        const int cin = d->channels[i], c = d->channels[i + 1];
        const int64_t To = conv_out_len(Tc);
        const bool last_stage = i == d->n_stages - 1;
        const int64_t M = (int64_t)B * To;
        const bool allsplit = stage_allsplit(i, cur, cur_split);
        // resize conv: cur -> a (a buffer other than cur's)
        const int io = (ia + 1) % 4;
        float* a = (last_stage && d->depths[i] == 0) ? y : buf[io];
        // option gconv_c1_fuse (round 6, off by default): the first stage's resize conv (1 mel bin -> 10 channels per group) runs INSIDE the
        // first TDSBlock conv's launch -- its output is computed straight into that kernel's LDS slab and never touches memory.  Bit-identical
        // and one launch + 1.15 GB of traffic per hour of audio less, but the conv's 3 G multiply-adds cost the same ~0.25 ms inside the
        // matrix-core kernel as in a launch of their own (profiles/r6_gconv_c1_fusion.txt): measured, kept for the record, not the default
        const bool c1_fused = allsplit && i == 0 && !cur_split && !s2_mfma_ok(i, Tc) && d->depths[i] > 0 && opt(OPT_GCONV_C1_FUSE) &&
                              !opt(OPT_GCONV_C1_GENERIC) && gconv_c1_res_fusable(cin, c, d->groups, cur);
        if (c1_fused) {
            rc = TAL_OK;
        } else if (allsplit) {
            if (s2_mfma_ok(i, Tc))
                rc = launch_gconv_s2_f16x3(cur, d->down_w_frag[i], d->down_b[i], B, Tc, cin, c, d->groups, nullptr, s, range_flag, cur_split, a);
            else
                rc = launch_gconv_s2(cur, d->down_w[i], d->down_b[i], B, Tc, cin, c, d->groups, nullptr, s, a, range_flag, i == 0 ? x_mean : nullptr);
        } else {
            TAL_CHECK_ARG(!cur_split, "tal_tds_fwd: internal: stage %d would read a split activation through an fp32 kernel", i);
            // stride-2 resize conv: on the matrix cores when the fragments are there (10 -> 14, 14 -> 18 per group); it pays from
            // ~a hundred output steps on: 358 steps of 18 channels per group take 13 us against 57
            if (s2_mfma_ok(i, Tc))
                rc = launch_gconv_s2_f16x3(cur, d->down_w_frag[i], d->down_b[i], B, Tc, cin, c, d->groups, a, s, range_flag);
            else
                rc = launch_gconv_s2(cur, d->down_w[i], d->down_b[i], B, Tc, cin, c, d->groups, a, s, nullptr, nullptr, i == 0 ? x_mean : nullptr);
        }
        if (rc) return rc;
        ia = io;
        // what the next consumer of this stage's output can read: the next stage's resize conv takes the split form only if
        // that stage runs all-split itself through a matrix-core resize conv
        const bool next_split = !last_stage && d->channels[i + 1] % 32 == 0 && s2_mfma_ok(i + 1, To) && stage_allsplit(i + 1, nullptr, true);
        for (int j = 0; j < d->depths[i]; ++j) {
            const tal_tds_block_w& bw = d->blocks[i][j];
            TAL_CHECK_ARG(bw.conv_w && bw.conv_b && bw.fc0_w && bw.fc0_b && bw.fc3_w && bw.fc3_b, "tal_tds_fwd: null weight in block %d.%d", i, j);
            float* x1 = buf[(ia + 1) % 4];
            float* h = buf[(ia + 2) % 4];
            float* x1s = buf[(ia + 3) % 4];
            const bool last_block = j == d->depths[i] - 1;
            float* outp = (last_stage && last_block) ? y : buf[ia];
            if (allsplit) {
                // x1 = x + rw * relu(gconv(x)): a (split) -> x1 (split); h = relu(fc0(x1)) (split); out = x1 + rw * fc3(h)
                if (c1_fused && j == 0)
                    rc = launch_gconv_c1_res_f16x3(cur, d->down_w[i], d->down_b[i], x_mean, bw.conv_w_frag, bw.conv_b, bw.resweight, B, Tc, d->groups, x1, s, range_flag);
                else
                    rc = launch_gconv_res_f16x3(a, bw.conv_w_frag, bw.conv_b, bw.resweight, B, To, c, d->groups, nullptr, x1, s, range_flag, true);
                if (rc) return rc;
                rc = launch_linear_f16x3(x1, bw.fc0_w_split, bw.fc0_b, nullptr, 0.f, 1, M, c, c, h, 1, skws, gemm_splitk_ws_bytes(), s, range_flag);
                if (rc) return rc;
                // (TAL_TDS_OUT_SPLIT: the caller's consumer -- the diarization head's embedding layer -- takes the split form itself)
                const bool out_split = !last_block || next_split || (last_stage && (d->flags & TAL_TDS_OUT_SPLIT) != 0);
                rc = launch_linear_f16x3(h, bw.fc3_w_split, bw.fc3_b, x1, bw.resweight, 2, M, c, c, outp, out_split ? 1 : 0, skws,
                                         gemm_splitk_ws_bytes(), s, range_flag, 1);
                if (rc) return rc;
                a = outp;
                cur_split = out_split;
                continue;
            }
            const bool f16x3 = !force_f32 && bw.fc0_w_split && bw.fc3_w_split && M > 128 && c % 160 == 0;
            const bool conv_mfma = !force_f32 && M > 64 && bw.conv_w_frag && gconv_f16x3_weight_bytes(c, c, d->groups, 1) > 0 && gconv_f16x3_fits(To, c);
            // x1 = x + rw * relu(gconv(x))            : a -> x1
            const bool fuse_split = opt(OPT_GCONV_FUSE_SPLIT) != 0;
            if (conv_mfma)
                rc = launch_gconv_res_f16x3(a, bw.conv_w_frag, bw.conv_b, bw.resweight, B, To, c, d->groups, x1, (fuse_split && f16x3) ? x1s : nullptr, s,
                                            range_flag);
            else
                rc = launch_gconv_res(a, bw.conv_w, bw.conv_b, bw.resweight, B, To, c, d->groups, x1, s);
            if (rc) return rc;
            if (f16x3) {
                // the two dense layers in the fp16x3 form: x1 is split once, fc0 writes its output already split
                if (!(conv_mfma && fuse_split)) rc = launch_split_f16x3(x1, x1s, M, c, s, range_flag);
                if (rc) return rc;
                rc = launch_linear_f16x3(x1s, bw.fc0_w_split, bw.fc0_b, nullptr, 0.f, 1, M, c, c, h, 1, skws, gemm_splitk_ws_bytes(), s, range_flag);
                if (rc) return rc;
                rc = launch_linear_f16x3(h, bw.fc3_w_split, bw.fc3_b, x1, bw.resweight, 2, M, c, c, outp, 0, skws,
                                         gemm_splitk_ws_bytes(), s);
                if (rc) return rc;
            } else {
                // h = relu(fc0(x1))                        : x1 -> h
                rc = launch_linear_ws(x1, bw.fc0_w, bw.fc0_b, nullptr, 0.f, 1, M, c, c, h, skws, gemm_splitk_ws_bytes(), s);
                if (rc) return rc;
                // x2 = x1 + rw * fc3(h)                    : h (+res x1) -> a's buffer (dead since the gconv)
                rc = launch_linear_ws(h, bw.fc3_w, bw.fc3_b, x1, bw.resweight, 2, M, c, c, outp, skws,
                                      gemm_splitk_ws_bytes(), s);
                if (rc) return rc;
            }
            a = outp;
            cur_split = false;
        }
        if (d->depths[i] == 0) cur_split = false;
        cur = a;
        Tc = To;
    }
    if (cur_split != y_split_out) {      // (the two decision chains disagree: never silently)
        set_error("tal_tds_fwd: internal: y was written %s but the status block says %s", cur_split ? "split" : "fp32", y_split_out ? "split" : "fp32");
        return TAL_EINVAL;
    }
    return TAL_OK;
}

// ---------------------------------------------------------------------------------------
// Time-tiled encoder (SURVEY section 8b `halo_mode`): one long item as tiles with the receptive-field halo
// ---------------------------------------------------------------------------------------
// Output frame t of the stack reads the input frames [stride t - left, stride t + right]: per stage (from the last one
// down) `depth` TDSBlocks of +-10 frames at that resolution, below them the stride-2 k = 21 conv (frame u reads [2 u, 2 u + 20]).
extern "C" int tal_tds_halo(const tal_tds_desc* d, int64_t* left, int64_t* right, int64_t* stride) {
    int rc = check_desc(d);
    if (rc) return rc;
    int64_t lo = 0, hi = 0, st = 1;
    for (int i = d->n_stages - 1; i >= 0; --i) {
        lo -= (int64_t)d->depths[i] * 10;
        hi += (int64_t)d->depths[i] * 10;
        lo = 2 * lo;
        hi = 2 * hi + 20;
        st *= 2;
    }
    if (left) *left = -lo;
    if (right) *right = hi;
    if (stride) *stride = st;
    return TAL_OK;
}

namespace {
struct TilePlan { int64_t in_start, in_stop, out_start, out_stop, skip; };
// tile k of `out_tile` output frames: the slice starts at a multiple of the total stride (every stage's index 0 of the slice
// is then a stage index of the whole sequence) and carries the halo; at a true end of the sequence there is nothing to carry
// -- the zero padding of the block convs is the same in the slice
TilePlan plan_tile(int64_t T, int64_t t_out, int64_t out_tile, int64_t k, int64_t left, int64_t right, int64_t stride) {
    TilePlan p;
    p.out_start = k * out_tile;
    p.out_stop = p.out_start + out_tile < t_out ? p.out_start + out_tile : t_out;
    const int64_t a = stride * (p.out_start - (left + stride - 1) / stride);
    p.in_start = a > 0 ? a : 0;
    const int64_t b = stride * (p.out_stop - 1) + right + 1;
    p.in_stop = b < T ? b : T;
    p.skip = p.out_start - p.in_start / stride;
    return p;
}
__global__ void or_status_kernel(const int* __restrict__ tile_flag, int* __restrict__ call_flag) {
    if (threadIdx.x == 0 && *tile_flag) atomicOr(call_flag, *tile_flag);
}
size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }
}  // namespace

// workspace = [slice output | tal_tds_fwd workspace of the longest slice | 64-byte status block of the whole call]
static int tiled_layout(const tal_tds_desc* d, int64_t T, int64_t out_tile, size_t* slice_out_bytes, size_t* tds_bytes, int64_t* max_slice) {
    int64_t left, right, stride;
    int rc = tal_tds_halo(d, &left, &right, &stride);
    if (rc) return rc;
    const int64_t t_out = tal_tds_out_len(d, T);
    TAL_CHECK_ARG(t_out > 0 && out_tile > 0, "tal_tds_tiled: T=%lld too short or out_tile=%lld not positive", (long long)T, (long long)out_tile);
    int64_t longest = 0;
    const int64_t n_tiles = cdiv(t_out, out_tile);
    for (int64_t k = 0; k < n_tiles; ++k) {       // (the first, an interior and the last tile would do; tiles are few)
        const TilePlan p = plan_tile(T, t_out, out_tile, k, left, right, stride);
        if (p.in_stop - p.in_start > longest) longest = p.in_stop - p.in_start;
    }
    *max_slice = longest;
    *slice_out_bytes = up256((size_t)tal_tds_out_len(d, longest) * d->channels[d->n_stages] * sizeof(float));
    *tds_bytes = up256(tal_tds_workspace_bytes(d, 1, longest));
    return TAL_OK;
}

extern "C" size_t tal_tds_tiled_status_offset(const tal_tds_desc* d, int64_t T, int64_t out_tile) {
    size_t a, b;
    int64_t m;
    if (!d || tiled_layout(d, T, out_tile, &a, &b, &m)) return 0;
    return a + b;
}

extern "C" size_t tal_tds_tiled_workspace_bytes(const tal_tds_desc* d, int64_t T, int64_t out_tile) {
    const size_t off = tal_tds_tiled_status_offset(d, T, out_tile);
    return off ? off + 64 : 0;
}

extern "C" int tal_tds_tiled_fwd(const tal_tds_desc* d, const float* x, int64_t T, float* y, int64_t out_tile, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    TAL_CHECK_ARG(x && y && workspace && d, "tal_tds_tiled_fwd: null pointer");
    // tiles are stitched as fp32 rows: the split output form is not offered here, whatever the caller's flags say
    tal_tds_desc fp32_out = *d;
    fp32_out.flags &= ~(uint32_t)TAL_TDS_OUT_SPLIT;
    d = &fp32_out;
    size_t so, tb;
    int64_t longest;
    int rc = tiled_layout(d, T, out_tile, &so, &tb, &longest);
    if (rc) return rc;
    if (workspace_bytes < so + tb + 64) {
        set_error("tal_tds_tiled_fwd: workspace %zu < %zu bytes", workspace_bytes, so + tb + 64);
        return TAL_ENOMEM;
    }
    hipStream_t s = (hipStream_t)stream;
    int64_t left, right, stride;
    tal_tds_halo(d, &left, &right, &stride);
    const int64_t t_out = tal_tds_out_len(d, T), n_tiles = cdiv(t_out, out_tile);
    const int c_in = d->channels[0], c_out = d->channels[d->n_stages];
    char* base = reinterpret_cast<char*>(workspace);
    float* slice_out = reinterpret_cast<float*>(base);
    void* tds_ws = base + so;
    int* call_flag = reinterpret_cast<int*>(base + so + tb);
    clear_status_kernel<<<1, 16, 0, s>>>(call_flag);
    if (hipGetLastError() != hipSuccess) {
        set_error("tal_tds_tiled_fwd: cannot clear the status word");
        return TAL_EHIP;
    }
    for (int64_t k = 0; k < n_tiles; ++k) {
        const TilePlan p = plan_tile(T, t_out, out_tile, k, left, right, stride);
        const int64_t Ts = p.in_stop - p.in_start;
        // a tile that is the whole sequence goes straight into y
        const bool direct = n_tiles == 1;
        rc = tal_tds_fwd(d, x + p.in_start * c_in, 1, Ts, direct ? y : slice_out, tds_ws, tb, stream);
        if (rc) return rc;
        const int* tile_flag = reinterpret_cast<const int*>(reinterpret_cast<char*>(tds_ws) + tal_tds_status_offset(d, 1, Ts));
        hipLaunchKernelGGL(or_status_kernel, dim3(1), dim3(64), 0, s, tile_flag, call_flag);
        TAL_CHECK_LAUNCH("tal_tds_tiled_fwd(status)");
        if (!direct && hipMemcpyAsync(y + p.out_start * c_out, slice_out + p.skip * c_out, (size_t)(p.out_stop - p.out_start) * c_out * sizeof(float),
                                      hipMemcpyDeviceToDevice, s) != hipSuccess) {
            set_error("tal_tds_tiled_fwd: copy of tile %lld failed", (long long)k);
            return TAL_EHIP;
        }
    }
    return TAL_OK;
}

extern "C" int tal_split_f16x3_fwd(const float* x, void* out, int64_t rows, int K, void* stream) {
    return launch_split_f16x3(x, out, rows, K, (hipStream_t)stream);
}

extern "C" int tal_linear_f16x3_fwd(const void* x_split, const void* w_split, const float* b, const float* res, float alpha,
                                    int mode, int64_t M, int N, int K, void* y, int out_split, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    TAL_CHECK_ARG(workspace || workspace_bytes == 0, "tal_linear_f16x3_fwd: null workspace with %zu bytes", workspace_bytes);
    return launch_linear_f16x3(x_split, w_split, b, res, alpha, mode, M, N, K, y, out_split, (float*)workspace, workspace_bytes,
                               (hipStream_t)stream);
}

// the same layer under the fp16-range guard of tal_tds_fwd (range_flag: a device int that is OR-ed with 1 when a value
// turned into halves lies outside the fp16 range) and, for mode 2, with the residual in the split form too
extern "C" int tal_linear_f16x3_guarded_fwd(const void* x_split, const void* w_split, const float* b, const void* res, int res_split,
                                            float alpha, int mode, int64_t M, int N, int K, void* y, int out_split,
                                            int* range_flag, void* workspace, size_t workspace_bytes, void* stream) {
    TAL_CHECK_ARG(workspace || workspace_bytes == 0, "tal_linear_f16x3_guarded_fwd: null workspace with %zu bytes", workspace_bytes);
    return launch_linear_f16x3(x_split, w_split, b, reinterpret_cast<const float*>(res), alpha, mode, M, N, K, y, out_split,
                               (float*)workspace, workspace_bytes, (hipStream_t)stream, range_flag, res_split);
}

// ---------------------------------------------------------------------------------------
// Diarization head
// ---------------------------------------------------------------------------------------
static int sd_head_part_ld(int S) {
    const int64_t a = cdiv(S, 32), b = 4 * cdiv(S, 128);
    return (int)(a > b ? a : b);
}

static size_t sd_head_partials_bytes(int64_t M, int S) { return (size_t)(M > 0 ? M : 1) * (size_t)sd_head_part_ld(S) * 8; }

extern "C" size_t tal_sd_head_workspace_bytes(int64_t M, int S) {
    // (value, index) partials of the fused arg-max: one pair per row and per 32-column wave slice.  The small-M
    // kernel (32 x 128 tile, 4 waves side by side) writes 4 slots per 128-column tile, the last tile included
    // even when it is partly past S, hence the second term.
    // ... and behind them (a multiple of 256 on) room for the speaker-logit weights as hi / lo fp16 halves (S x 128 x 4 bytes), which
    // the A-stationary arg-max kernel multiplies in the fp16x3 form
    const size_t partials = ((sd_head_partials_bytes(M, S) + 255) & ~(size_t)255) + (size_t)S * 128 * 4;
    // The same bytes serve first as K-slice scratch of the embedding layer (1440 -> 128: ONE column tile, so a clip of minutes is
    // a few dozen 128-row tiles for 256 CUs -- 120 us on 30 workgroups for a 5-minute clip; cut along K 8 ways it is ~30 us).
    // launch_gemm slices only launches below a quarter round (128 tiles).
    const int64_t tiles = cdiv(M > 0 ? M : 1, 128);
    const size_t slices = tiles <= 128 ? (size_t)tiles * 8 * 128 * 160 * sizeof(float) : 0;
    return partials > slices ? partials : slices;
}

static int sd_head_after_feat(int64_t M, int E, const float* w_logit, const float* b_logit, int S, float* feat, float* logits,
                              int32_t* ids, void* workspace, size_t workspace_bytes, hipStream_t s);

extern "C" int tal_sd_head_fwd(const float* x, int64_t M, int C, const float* w_embed, const float* b_embed, int E,
                               const float* w_logit, const float* b_logit, int S, float* feat, float* logits,
                               int32_t* ids, void* workspace, size_t workspace_bytes, void* stream) {
    TAL_CHECK_ARG(x && w_embed && w_logit && feat, "tal_sd_head_fwd: null pointer");
    TAL_CHECK_ARG(M >= 0 && C > 0 && E > 0 && S > 0, "tal_sd_head_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    // (with a workspace the embedding layer of a short / medium input is cut along K: see tal_sd_head_workspace_bytes; the
    //  workspace's later use -- arg-max partials -- is ordered behind it on the stream)
    int rc = workspace && workspace_bytes >= tal_sd_head_workspace_bytes(M, S) && M > 512
                 ? launch_linear_ws(x, w_embed, b_embed, nullptr, 0.f, 0, M, E, C, feat, reinterpret_cast<float*>(workspace), workspace_bytes, s)
                 : launch_linear(x, w_embed, b_embed, nullptr, 0.f, 0, M, E, C, feat, s);
    if (rc) return rc;
    return sd_head_after_feat(M, E, w_logit, b_logit, S, feat, logits, ids, workspace, workspace_bytes, s);
}

// The same head on an encoder output that is still in the hi / lo split form (tal_tds_fwd with TAL_TDS_OUT_SPLIT): the 1440 -> 128
// embedding layer runs in the fp16x3 form of the dense layers on the split rows and on pre-split weights (tal_split_f16x3_fwd of
// spk_embed_proj.weight) -- 77 instead of 245 us per hour of audio, max error against float64 3e-6 instead of 9e-6 -- and the
// last TDS layer never writes an fp32 copy.  Everything behind the features is tal_sd_head_fwd's.  M > 128, C % 32 == 0.
extern "C" int tal_sd_head_split_fwd(const void* x_split, int64_t M, int C, const void* w_embed_split, const float* b_embed, int E,
                                     const float* w_logit, const float* b_logit, int S, float* feat, float* logits,
                                     int32_t* ids, void* workspace, size_t workspace_bytes, void* stream) {
    TAL_CHECK_ARG(x_split && w_embed_split && w_logit && feat, "tal_sd_head_split_fwd: null pointer");
    TAL_CHECK_ARG(M > 128 && C > 0 && C % 32 == 0 && E > 0 && S > 0, "tal_sd_head_split_fwd: needs M > 128 rows and C %% 32 == 0 (M=%lld, C=%d)", (long long)M, C);
    hipStream_t s = (hipStream_t)stream;
    const bool ws_ok = workspace && workspace_bytes >= tal_sd_head_workspace_bytes(M, S);
    int rc = launch_linear_f16x3(x_split, w_embed_split, b_embed, nullptr, 0.f, 0, M, E, C, feat, 0, ws_ok ? reinterpret_cast<float*>(workspace) : nullptr,
                                 ws_ok ? workspace_bytes : 0, s);
    if (rc) return rc;
    return sd_head_after_feat(M, E, w_logit, b_logit, S, feat, logits, ids, workspace, workspace_bytes, s);
}

static int sd_head_after_feat(int64_t M, int E, const float* w_logit, const float* b_logit, int S, float* feat, float* logits,
                              int32_t* ids, void* workspace, size_t workspace_bytes, hipStream_t s) {
    int rc;
    if (!logits && !ids) return TAL_OK;
    if (logits) {
        rc = launch_linear(feat, w_logit, b_logit, nullptr, 0.f, 0, M, S, E, logits, s);
        if (rc) return rc;
        return ids ? launch_argmax_rows(logits, M, S, ids, s) : TAL_OK;
    }
    TAL_CHECK_ARG(workspace, "tal_sd_head_fwd: ids without logits needs a workspace");
    // (the arg-max partials are what the call cannot do without; the larger figure tal_sd_head_workspace_bytes returns for short /
    //  medium inputs only enables the K-sliced embedding layer above)
    if (workspace_bytes < sd_head_partials_bytes(M, S)) {
        set_error("tal_sd_head_fwd: workspace %zu < %zu bytes", workspace_bytes, sd_head_partials_bytes(M, S));
        return TAL_ENOMEM;
    }
    if (M == 0) return TAL_OK;
    if (b_logit && head_argmax_applicable(M, S, E)) {
        // long inputs: feature strip stationary in registers, running arg-max across N tiles (csrc/head.hip)
        const int P = head_argmax_partials(M, S);
        float* pv = reinterpret_cast<float*>(workspace);
        int32_t* pi = reinterpret_cast<int32_t*>(pv + (size_t)M * P);
        // the speaker-logit weights as hi / lo fp16 split in the unused tail of the workspace (3 MB, one ~5 us pass per call):
        // the arg-max GEMM then runs in the fp16x3 form of the dense layers
        const bool head_f32 = opt(OPT_TDS_EXACT_F32) != 0;
        void* wsplit = nullptr;
        const size_t used = ((size_t)M * P * 8 + 255) & ~(size_t)255;
        if (!head_f32 && E % 32 == 0 && (reinterpret_cast<uintptr_t>(w_logit) & 15) == 0 && used + (size_t)S * E * 4 <= workspace_bytes) {
            wsplit = reinterpret_cast<char*>(workspace) + used;
            rc = launch_split_f16x3(w_logit, wsplit, S, E, s);
            if (rc) return rc;
        }
        rc = launch_head_argmax(feat, w_logit, wsplit, b_logit, M, S, pv, pi, s);
        if (rc) return rc;
        ProfScope prof(PROF_OTHER, (double)M * P * 8.0, s);
        hipLaunchKernelGGL(argmax_partials_kernel, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, s, pv, pi, M, P, P, ids);
        TAL_CHECK_LAUNCH("tal_sd_head_fwd(argmax)");
        return TAL_OK;
    }
    // logits are never materialised: the dense layer's epilogue reduces each row of its tile to a
    // (max, arg-max) pair, a tiny second kernel merges the pairs of a row's column tiles
    const int ld = sd_head_part_ld(S);
    GemmArgs g = {};
    g.A = feat; g.W = w_logit; g.bias = b_logit;
    g.M = M; g.N = S; g.K = E;
    g.lda = E; g.ldw = E; g.ldy = S; g.nb2 = 1;
    g.part_val = reinterpret_cast<float*>(workspace);
    g.part_idx = reinterpret_cast<int32_t*>(g.part_val + (size_t)M * ld);
    g.part_ld = ld;
    rc = launch_gemm(g, 4, 1, s);
    if (rc) return rc;
    {
        ProfScope prof(PROF_OTHER, (double)M * ld * 8.0, s);
        hipLaunchKernelGGL(argmax_partials_kernel, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, s, g.part_val, g.part_idx, M,
                           gemm_mode4_partials(M, S), ld, ids);
    }
    TAL_CHECK_LAUNCH("tal_sd_head_fwd(argmax)");
    return TAL_OK;
}
