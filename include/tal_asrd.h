/*
 * tal_asrd.h -- C ABI of the MI355X-native acoustic hot path of tal-asrd.
 *
 * The reference (calclavia/tal-asrd) has no FFI / plugin boundary: the hot path
 * sits behind Python nn.Module methods.  This header is
 * the boundary a replacement of those methods binds to; every entry point
 * names the reference method it computes (paths relative to the reference
 * repository root).  The Python host side (tal_asrd_amd/models.py) mirrors the
 * reference's classes 1:1 and calls these functions through ctypes
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - all data pointers are DEVICE pointers (HBM) owned by the caller, fp32
 *     unless stated; activations are time-major row-major [B, T, C]
 *     (channels contiguous) -- the layout LogMelSpec.forward returns and
 *     encode_features returns, so the reference's two permutes
 *     (tal/asr/models.py:167,170) disappear;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *     nothing synchronises, nothing allocates;
 *   - return value: 0 (TAL_OK) or a negative TAL_E* code; no C++ exception
 *     crosses the boundary; tal_last_error() gives a thread-local message;
 *   - re-entrant per stream; no global mutable state besides the error string.
 */
#ifndef TAL_ASRD_H
#define TAL_ASRD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAL_OK 0
#define TAL_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define TAL_ENOMEM (-2)   /* workspace too small */
#define TAL_EHIP (-3)     /* HIP runtime / launch failure */

#define TAL_MAX_STAGES 4
#define TAL_MAX_DEPTH 8

int tal_version(void);
const char* tal_last_error(void);

/* ------------------------------------------------------------------ *
 * Log-mel front-end: LogMelSpec.forward, tal/asr/models.py:35-53
 * (torchaudio 0.4.0 MelSpectrogram(sr=16000, n_fft=400, win=400, hop=160,
 * n_mels=80) -> log(mel+eps) -> minus ONE global scalar mean).
 * ------------------------------------------------------------------ */
/* frames for L samples: T = 1 + L / 160 (center=True STFT). */
int64_t tal_logmel_num_frames(int64_t L);
/* bytes of the device-resident plan (window-folded DFT basis + sparse mel filters). */
size_t tal_logmel_plan_bytes(void);
/* Build the plan from the module's buffers: window [400] (periodic Hann) and
 * fb [201, 80] (HTK triangles), the two torchaudio buffers reference
 * checkpoints carry. */
int tal_logmel_plan_init(const float* window, const float* fb, void* plan, void* stream);
size_t tal_logmel_workspace_bytes(int B, int64_t L);
/* audio [B, L] -> out [B, T, 80].  If subtract_mean != 0 the global mean of the
 * whole [B,T,80] tensor (models.py:52) is subtracted in place.  The mean that
 * was (or would be) subtracted is written to *mean_out (device float, may be NULL).
 * sum_out (device double[2] = {sum, count}, may be NULL) exposes the reduction so
 * a caller that splits one reference "call" across GPUs can all-reduce it. */
int tal_logmel_fwd(const void* plan, const float* audio, int B, int64_t L, float eps,
                   int subtract_mean, float* out, float* mean_out, double* sum_out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* x[i] -= *mean for n floats (second half of the two-step / multi-GPU form). */
int tal_subtract_scalar(float* x, int64_t n, const float* mean, void* stream);

/* ------------------------------------------------------------------ *
 * Dense layer: y = epilogue(x . W^T + b), nn.Linear / 1x1 Conv1d.
 * x [M, K] row-major (ld = K), W [N, K] row-major (the nn.Linear /
 * Conv1d(k=1) weight as stored in reference checkpoints), y [M, N].
 *   mode 0: y = acc + b             mode 1: y = relu(acc + b)
 *   mode 2: y = res + alpha * (acc + b)   (ReZero residual, models.py:330,516-527)
 * b may be NULL.  K % 4 == 0 required (rows 16-byte aligned).
 * ------------------------------------------------------------------ */
int tal_linear_fwd(const float* x, const float* w, const float* b, const float* res, float alpha,
                   int mode, int64_t M, int N, int K, float* y, void* stream);

/* ------------------------------------------------------------------ *
 * Grouped temporal convolutions of the TDS encoder.
 * Packed weight layout [G][C_in/G][21][C_out/G] (tal_pack_gconv_weight
 * converts from the reference's Conv1d layout [C_out, C_in/G, 21]).
 * ------------------------------------------------------------------ */
int tal_pack_gconv_weight(const float* w_ref, float* w_packed, int c_out, int c_in_per_group,
                          int ksize, int groups, void* stream);
/* Conv1d(C_in->C_out, k=21, stride=2, groups=G, padding=0), models.py:363-364.
 * x [B, T_in, C_in] -> y [B, T_out, C_out], T_out = (T_in-21)/2 + 1. */
int tal_gconv_s2_fwd(const float* x, const float* w_packed, const float* bias, int B, int64_t T_in,
                     int C_in, int C_out, int groups, float* y, void* stream);
/* y = x + alpha * relu(Conv1d(C->C, k=21, groups=G, padding=10)(x)), models.py:304-308,329.
 * Zero padding at the true ends of each batch item. */
int tal_gconv_res_fwd(const float* x, const float* w_packed, const float* bias, float alpha, int B,
                      int64_t T, int C, int groups, float* y, void* stream);

/* ------------------------------------------------------------------ *
 * Whole TDS encoder: TDS.forward, tal/asr/models.py:349-397 (+TDSBlock :298-331)
 * ------------------------------------------------------------------ */
typedef struct tal_tds_block_w {
    const float* conv_w;   /* packed grouped conv weight            (conv.0.weight) */
    const float* conv_b;   /* [C]                                    (conv.0.bias)   */
    const float* fc0_w;    /* [C, C]                                 (fc.0.weight)   */
    const float* fc0_b;    /* [C]                                    (fc.0.bias)     */
    const float* fc3_w;    /* [C, C]                                 (fc.3.weight)   */
    const float* fc3_b;    /* [C]                                    (fc.3.bias)     */
    float resweight;       /* host scalar                            (resweight)     */
    int32_t _pad;
} tal_tds_block_w;

typedef struct tal_tds_desc {
    int32_t n_stages;                 /* 3 */
    int32_t groups;                   /* 80 (= input_size) */
    int32_t channels[TAL_MAX_STAGES + 1]; /* {80, 800, 1120, 1440} */
    int32_t depths[TAL_MAX_STAGES];   /* {2, 3, 6} */
    const float* down_w[TAL_MAX_STAGES];  /* packed stride-2 conv weight (blocks.i.0.weight) */
    const float* down_b[TAL_MAX_STAGES];  /* (blocks.i.0.bias) */
    tal_tds_block_w blocks[TAL_MAX_STAGES][TAL_MAX_DEPTH];
} tal_tds_desc;

/* output length after all stride-2 stages: T' = f(f(f(T))), f(t) = (t-21)/2+1 */
int64_t tal_tds_out_len(const tal_tds_desc* d, int64_t T);
size_t tal_tds_workspace_bytes(const tal_tds_desc* d, int B, int64_t T);
/* x [B, T, channels[0]] -> y [B, T', channels[n_stages]] */
int tal_tds_fwd(const tal_tds_desc* d, const float* x, int B, int64_t T, float* y,
                void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ *
 * Diarization head: SDModel.decode + reconcile.get_speaker_ids,
 * tal/asr/models.py:473-481, tal/baseline/reconcile.py:76-85.
 * x [M, C] -> feat [M, E] = spk_embed_proj(x); logits [M, S] = spk_logit_proj(feat)
 * (logits may be NULL); ids [M] int32 = argmax over S (may be NULL; first
 * maximum wins, as torch.argmax).  workspace: tal_sd_head_workspace_bytes.
 * ------------------------------------------------------------------ */
size_t tal_sd_head_workspace_bytes(int64_t M, int S);
int tal_sd_head_fwd(const float* x, int64_t M, int C, const float* w_embed, const float* b_embed, int E,
                    const float* w_logit, const float* b_logit, int S, float* feat, float* logits,
                    int32_t* ids, void* workspace, size_t workspace_bytes, void* stream);
/* Row-wise argmax of a [M, N] fp32 matrix -> int32 ids (first maximum wins). */
int tal_argmax_rows(const float* x, int64_t M, int N, int32_t* ids, void* stream);

/* ------------------------------------------------------------------ *
 * Measurement hooks (bench.py): when enabled, every launch of the hot kernels
 * is bracketed by two hipEvents on its launch stream.  class: 0 dense-layer
 * GEMM, 1 TDSBlock grouped conv, 2 stride-2 grouped conv, 3 log-mel, 4 other.
 * tal_prof_collect synchronises on the recorded events and returns the summed
 * duration (ms), the number of launches and the summed algorithmic work
 * (flops for 0-2, HBM bytes for 3-4) since the last tal_prof_reset.
 * Not thread-safe; a process-wide switch meant for single-stream benchmarking.
 * ------------------------------------------------------------------ */
int tal_prof_enable(int on);
int tal_prof_reset(void);
int tal_prof_collect(int cls, double* total_ms, int64_t* launches, double* total_work);

#ifdef __cplusplus
}
#endif
#endif /* TAL_ASRD_H */
