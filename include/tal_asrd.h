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
 *     nothing allocates and nothing synchronises -- except tal_logmel_plan_init
 *     (one-time), tal_prof_collect and tal_greedy_step_fwd with sync != 0, which
 *     returns the token the host loop steers by;
 *   - return value: 0 (TAL_OK) or a negative TAL_E* code; no C++ exception
 *     crosses the boundary; tal_last_error() gives a thread-local message;
 *   - re-entrant per stream; no global mutable state besides the error string,
 *     the tal_prof_* measurement hooks (a process-wide switch for
 *     single-stream benchmarking, off by default) and the tal_set_option
 *     switches; the environment is never read.
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

int tal_version(void);          /* 500 = 0.5.0 */
const char* tal_last_error(void);

/* Process-wide behaviour switches.  The library never reads the environment: which kernels a caller gets depends on its
 * arguments and on these calls only (the host-side mirror applies TAL_OPTIONS="name=value,..." at load time, for the
 * measurement scripts).  Names (all default 0 unless stated):
 *   tds_exact_f32         every layer of tal_tds_fwd and the head arg-max on the exact fp32-input kernels
 *                         (per call: tal_tds_desc.flags & TAL_TDS_EXACT_F32)
 *   tds_fp32_activations  fp16x3 layers, but fp32 activations between the kernels of a stage
 *   gconv_fuse_split, gconv_c1_generic, head_no_astationary, gemm_global_loads, gemm_no_splitk4, gemm_no_glds,
 *   gemm_no_splitk_tail, gemm_no_w64, gemm_no_n96, gemm_no_row_split, logmel_no_fold, decode_no_small, gconv_no_shift18,
 *   gconv_grid_xyz, gemm_w64_stagger, gemm_s64_order, decode_wide_gemm, gemm_s64_rows
 *                         kernel-selection switches of the ablation measurements
 *   decode_small_rows     largest prefix the latency-oriented decoder layer takes (default 256)
 *   decode_persist        1: tal_greedy_step_fwd runs a step as ONE launch where its shapes allow (persistent workgroups walk the
 *                         step's phases behind counter barriers and run the launch chain's own kernel bodies: bit-identical results;
 *                         measured SLOWER than the chain, profiles/r5_decode_persistent_step.txt -- kept as a measurement switch)
 *   decode_persist_wgs    workgroups per session of that launch (default 32)
 *   logmel_mfma           1: the log-mel front-end as the float64 matrix-core DFT instead of the fast transform (see tal_logmel_fwd)
 *   gemm_s64_below        fp16x3 relu / residual layers run on 64 x 80 tiles without K slices while those tiles number
 *                         at most this many per CU (default 2; 0: never)
 *   gconv_short_below     grouped convs use 64-step tiles while the long tiles would give a CU fewer workgroups than
 *                         this (default 4; 0: always the long tiles).  Results do not depend on it.
 * tal_set_option returns TAL_EINVAL for an unknown name; tal_option_name(i) enumerates the names (NULL past the end). */
int tal_set_option(const char* name, int value);
int tal_get_option(const char* name, int* value);
const char* tal_option_name(int index);

/* ------------------------------------------------------------------ *
 * Log-mel front-end: LogMelSpec.forward, tal/asr/models.py:35-53
 * (torchaudio 0.4.0 MelSpectrogram(sr=16000, n_fft=400, win=400, hop=160,
 * n_mels=80) -> log(mel+eps) -> minus ONE global scalar mean).
 * ------------------------------------------------------------------ */
/* frames for L samples: T = 1 + L / 160 (center=True STFT). */
int64_t tal_logmel_num_frames(int64_t L);
/* bytes of the device-resident plan (window, twiddles of the 400 = 20 x 20 transform, sparse mel filters; the window-folded
 * DFT basis of the matrix form). */
size_t tal_logmel_plan_bytes(void);
/* Build the plan from the module's buffers: window [400] (periodic Hann) and
 * fb [201, 80] (HTK triangles), the two torchaudio buffers reference
 * checkpoints carry. */
int tal_logmel_plan_init(const float* window, const float* fb, void* plan, void* stream);
size_t tal_logmel_workspace_bytes(int B, int64_t L);
/* Two forms of the same arithmetic (window . frame, 400-point DFT, power: all float64; mel projection, log: float32):
 *   default        a fast transform on the float64 vector ALU: two real frames per complex transform, 400 = 20 x 20 with a
 *                  20-point prime-factor transform per lane (csrc/dft20.h); any window, applied in the time domain as given
 *   logmel_mfma=1  the float64 matrix-core DFT of rounds 1-4 (a symmetric window folded and symmetrised to 4e-7; 2.4x slower)
 * Also taken when the workspace is smaller than tal_logmel_workspace_bytes says (sized by an older header). */
/* audio [B, L] -> out [B, T, 80].  If subtract_mean != 0 the global mean of the
 * whole [B,T,80] tensor (models.py:52) is subtracted in place.  The mean that
 * was (or would be) subtracted is written to *mean_out (device float, may be NULL).
 * sum_out (device double[2] = {sum, count}, may be NULL) exposes the reduction so
 * a caller that splits one reference "call" across GPUs can all-reduce it. */
int tal_logmel_fwd(const void* plan, const float* audio, int B, int64_t L, float eps,
                   int subtract_mean, float* out, float* mean_out, double* sum_out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* The same call on a half-precision waveform (fp16 [B, L]): the reference's GPU-era call sites hand the model
 * `audio.half()` (tal/asr/system.py:92,285; tal/baseline/reconcile.py:78).  Samples are widened to fp32 while they are
 * staged (exact), so the result equals tal_logmel_fwd on the widened waveform bit for bit; out stays fp32. */
int tal_logmel_f16_fwd(const void* plan, const void* audio_f16, int B, int64_t L, float eps,
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
/* Same layer with caller-provided scratch: lets a large layer (M > 512) cut the tiles of its last,
 * partial scheduling round along K (results then differ from tal_linear_fwd in the last bits of those
 * rows: two partial sums instead of one chain).  tal_linear_workspace_bytes returns 0 when scratch
 * would not be used; workspace may then be NULL. */
size_t tal_linear_workspace_bytes(int64_t M, int N, int K);
int tal_linear_ws_fwd(const float* x, const float* w, const float* b, const float* res, float alpha,
                      int mode, int64_t M, int N, int K, float* y, void* workspace,
                      size_t workspace_bytes, void* stream);

/* fp16x3 form of a large dense layer: fp32 data is carried as hi = fp16(x), lo = fp16((x - hi) * 2^11) in the byte
 * geometry of fp32 rows (per row and 32-wide K block: 32 hi halves, then 32 lo halves), and
 *   x . w = sum hi_x hi_w + 2^-11 sum (hi_x lo_w + lo_x hi_w)        (lo_x lo_w, 2^-22 relative, dropped)
 * runs as three fp16 MFMAs with fp32 accumulation: measured error against float64 is below that of an fp32 fmaf
 * chain (scripts/ubench/gemm_f16x3.hip), at ~2.4x the fp32 matrix rate.  Requires |x| < 65504 (fp16 range).
 * tal_split_f16x3_fwd: x [rows, K] fp32 -> out (rows * K * 4 bytes), K % 32 == 0.
 * tal_linear_f16x3_fwd: tal_linear_ws_fwd on pre-split x / w (M > 512, K % 32 == 0, modes 0-3); out_split != 0
 * writes y in the split form too (N % 160 == 0), as the next layer's input. */
int tal_split_f16x3_fwd(const float* x, void* out, int64_t rows, int K, void* stream);
int tal_linear_f16x3_fwd(const void* x_split, const void* w_split, const float* b, const float* res, float alpha,
                         int mode, int64_t M, int N, int K, void* y, int out_split, void* workspace,
                         size_t workspace_bytes, void* stream);
/* The same layer as tal_tds_fwd runs it inside a TDS block: under the fp16-range guard (range_flag: device int, OR-ed
 * with 1 when a value turned into halves lies outside the fp16 range; may be NULL) and, for mode 2, optionally with the
 * residual in the split form (res_split != 0: res points to split data, N % 160 == 0). */
int tal_linear_f16x3_guarded_fwd(const void* x_split, const void* w_split, const float* b, const void* res, int res_split,
                                 float alpha, int mode, int64_t M, int N, int K, void* y, int out_split,
                                 int* range_flag, void* workspace, size_t workspace_bytes, void* stream);

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

/* The grouped convs on the fp16 matrix cores in the fp16x3 form of the dense layers (three fp16 MFMAs per fp32 product
 * block, fp32 accumulation; error against float64 below an fp32 fmaf chain).  Built for the TDSBlock conv (stride 1,
 * C_in = C_out, models.py:304-308,329) with C / groups = 10, 14, 18 and for the stride-2 resize convs (models.py:363-364)
 * 10 -> 14 and 14 -> 18 channels per group, groups % 4 == 0; tal_gconv_f16x3_weight_bytes returns 0 for anything else.
 * tal_pack_gconv_f16x3_weight: reference Conv1d weight [C_out, C_in/G, 21] -> MFMA operand fragments (hi / lo halves).
 * tal_gconv_res_f16x3_fwd: y = x + alpha * relu(conv(x) + b); y_split != NULL additionally receives y as the hi / lo
 * split the fp16x3 dense layers consume (tal_split_f16x3_fwd's format; C % 32 == 0), fused into the store.
 * tal_gconv_s2_f16x3_fwd: y = conv(x) + b, stride 2, no padding: [B, T_in, C_in] -> [B, (T_in-21)/2+1, C_out]. */
size_t tal_gconv_f16x3_weight_bytes(int C_in, int C_out, int groups, int stride);
int tal_pack_gconv_f16x3_weight(const float* w_ref, void* w_frag, int C_in, int C_out, int groups, int stride,
                                void* stream);
int tal_gconv_res_f16x3_fwd(const float* x, const void* w_frag, const float* bias, float alpha, int B,
                            int64_t T, int C, int groups, float* y, void* y_split, void* stream);
int tal_gconv_s2_f16x3_fwd(const float* x, const void* w_frag, const float* bias, int B, int64_t T_in,
                           int C_in, int C_out, int groups, float* y, void* stream);
/* The same convs on activations that live in the hi / lo split form (tal_split_f16x3_fwd's format: the canonical
 * activation format between the kernels of a long-input TDS stage): the input slab is filled without any conversion
 * arithmetic, the TDSBlock residual is rebuilt as hi + lo * 2^-11, and only the split form of the output is written.
 * tal_gconv_s2_split_fwd also accepts an fp32 input (x_is_split == 0). */
int tal_gconv_res_split_fwd(const void* x_split, const void* w_frag, const float* bias, float alpha, int B,
                            int64_t T, int C, int groups, void* y_split, void* stream);
int tal_gconv_s2_split_fwd(const void* x, int x_is_split, const void* w_frag, const float* bias, int B,
                           int64_t T_in, int C_in, int C_out, int groups, void* y_split, void* stream);

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
    const void* fc0_w_split; /* fc.0.weight / fc.3.weight as hi/lo fp16 splits (tal_split_f16x3_fwd), or NULL: */
    const void* fc3_w_split; /* with both present the block's dense layers run in the fp16x3 form for M > 512   */
    const void* conv_w_frag; /* conv.0.weight as fp16x3 MFMA fragments (tal_pack_gconv_f16x3_weight), or NULL: with it
                              * (and the two splits above) the grouped conv runs on the matrix cores for M > 512     */
} tal_tds_block_w;

typedef struct tal_tds_desc {
    int32_t n_stages;                 /* 3 */
    int32_t groups;                   /* 80 (= input_size) */
    int32_t channels[TAL_MAX_STAGES + 1]; /* {80, 800, 1120, 1440} */
    int32_t depths[TAL_MAX_STAGES];   /* {2, 3, 6} */
    const float* down_w[TAL_MAX_STAGES];  /* packed stride-2 conv weight (blocks.i.0.weight) */
    const float* down_b[TAL_MAX_STAGES];  /* (blocks.i.0.bias) */
    tal_tds_block_w blocks[TAL_MAX_STAGES][TAL_MAX_DEPTH];
    const void* down_w_frag[TAL_MAX_STAGES]; /* stride-2 conv weights as fp16x3 MFMA fragments, or NULL (VALU kernel) */
    int32_t flags;                    /* TAL_TDS_EXACT_F32: every layer on the exact fp32-input kernels (no fp16x3 form) */
    int32_t _pad2;
} tal_tds_desc;
#define TAL_TDS_EXACT_F32 1
#define TAL_TDS_OUT_SPLIT 2   /* leave y in the hi / lo split form where the last stage runs all-split (tal_tds_out_split() says whether
                               * a call will): for tal_sd_head_split_fwd, whose embedding layer consumes that form -- the last dense layer
                               * then never writes an fp32 copy of the encoder output */

/* output length after all stride-2 stages: T' = f(f(f(T))), f(t) = (t-21)/2+1 */
int64_t tal_tds_out_len(const tal_tds_desc* d, int64_t T);
size_t tal_tds_workspace_bytes(const tal_tds_desc* d, int B, int64_t T);
/* fp16-range guard.  The fp16x3 form carries fp32 values as two fp16 halves, so it needs |x| <= 65504 for every
 * activation it converts (and for the weights, which the caller checks when it builds the splits).  Every converting
 * kernel tracks max |x|; tal_tds_fwd clears a status word in its workspace at tal_tds_status_offset() bytes and the
 * kernels raise it (non-zero int32) when a value was out of range -- the output of that call is then NOT valid and the
 * caller re-runs it with desc->flags |= TAL_TDS_EXACT_F32 (fp32-input MFMA kernels, no range limit).  Values below
 * 2^-24 in magnitude lose their low half (absolute error <= 2^-35 per product, far below fp32 resolution of the sums).
 * Stream capture: tal_logmel_*_fwd, tal_tds_fwd and tal_sd_head_fwd enqueue kernel launches on `stream` and nothing else (the
 * status word is cleared by a kernel, not a memset node), so a caller may capture them into a HIP graph once every one-off
 * build (plans, weight packs) has run eagerly; the status word is read after each replay like after each call. */
size_t tal_tds_status_offset(const tal_tds_desc* d, int B, int64_t T);
/* 1: tal_tds_fwd / tal_tds_premean_fwd with TAL_TDS_OUT_SPLIT in d->flags writes y in the split form for these shapes; 0: fp32.
 * A PREDICTION (made for a 16-byte aligned x under the options in force at the time of the query): the call itself records the
 * form it wrote in word 1 of its status block (int32 at tal_tds_status_offset() + 4: 1 = split, 0 = fp32), and a caller that
 * hands y to tal_sd_head_split_fwd on the strength of the prediction reads that word together with the range flag.
 * tal_tds_tiled_fwd always writes fp32 (the flag is ignored there). */
int tal_tds_out_split(const tal_tds_desc* d, int B, int64_t T);
/* x [B, T, channels[0]] -> y [B, T', channels[n_stages]] */
int tal_tds_fwd(const tal_tds_desc* d, const float* x, int B, int64_t T, float* y,
                void* workspace, size_t workspace_bytes, void* stream);
/* The same call on a log-mel tensor BEFORE LogMelSpec's global-mean subtraction (tal/asr/models.py:52: `x -= x.mean()`;
 * tal_logmel_fwd with subtract_mean = 0 leaves the scalar in *mean_out): x_mean is a DEVICE pointer to that scalar.  The first
 * resize conv has no padding (models.py:363-364), so conv(x - m) = conv(x) - m * sum_k w[k]: the subtraction is applied as a
 * correction of that conv's bias and the separate pass over the log-mel (two activation-sized transfers and a launch) is not run.
 * Same workspace, status word and re-run rule as tal_tds_fwd.  Only for stacks whose first resize conv is the 1 -> 10 channels
 * per group form (tal_tds_premean_ok() != 0: the reference's 80 -> 800 encoder, x 16-byte aligned); TAL_EINVAL otherwise.
 * Results differ from tal_tds_fwd on the subtracted tensor by fp32 rounding of the first conv only. */
int tal_tds_premean_ok(const tal_tds_desc* d, const float* x);
int tal_tds_premean_fwd(const tal_tds_desc* d, const float* x, const float* x_mean, int B, int64_t T, float* y,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Time-tiled form of the same call for ONE item (SURVEY.md section 8b `halo_mode`): the input is cut into tiles of
 * `out_tile` output frames; each tile runs as its own tal_tds_fwd over the slice of x that carries its receptive-field halo
 * (tal_tds_halo: output frame t reads the input frames [stride t - left, stride t + right]; 640 / 780 / 8 for the 2 / 3 / 6
 * block stack) and reproduces its frames as the whole sequence would -- slices start at multiples of the stride, and at a
 * true end of the sequence the blocks' zero padding is the same in the slice.  For items beyond the 2 GiB-per-item limit
 * of the fp16x3 kernels' 32-bit offsets (~3.7 h of audio), which would otherwise take the generic kernels, and for
 * bounding the workspace.  The fp16-range status word of the whole call (OR over the tiles) sits at
 * tal_tds_tiled_status_offset() bytes of the workspace; a caller that finds it raised re-runs with TAL_TDS_EXACT_F32. */
int tal_tds_halo(const tal_tds_desc* d, int64_t* left, int64_t* right, int64_t* stride);
size_t tal_tds_tiled_workspace_bytes(const tal_tds_desc* d, int64_t T, int64_t out_tile);
size_t tal_tds_tiled_status_offset(const tal_tds_desc* d, int64_t T, int64_t out_tile);
/* x [1, T, channels[0]] -> y [1, T', channels[n_stages]] */
int tal_tds_tiled_fwd(const tal_tds_desc* d, const float* x, int64_t T, float* y, int64_t out_tile,
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
/* The same head on an encoder output still in the hi / lo split form (tal_tds_fwd with TAL_TDS_OUT_SPLIT where tal_tds_out_split()
 * says so): x_split [M, C] split rows, w_embed_split = tal_split_f16x3_fwd of spk_embed_proj.weight [E, C]; the embedding layer
 * runs in the fp16x3 form (fp32-equivalent: max error against float64 3e-6 on the 1-hour shape, the fp32 kernel's 9e-6), everything
 * behind `feat` as above.  M > 128, C % 32 == 0. */
int tal_sd_head_split_fwd(const void* x_split, int64_t M, int C, const void* w_embed_split, const float* b_embed, int E,
                          const float* w_logit, const float* b_logit, int S, float* feat, float* logits,
                          int32_t* ids, void* workspace, size_t workspace_bytes, void* stream);
/* Row-wise argmax of a [M, N] fp32 matrix -> int32 ids (first maximum wins). */
int tal_argmax_rows(const float* x, int64_t M, int N, int32_t* ids, void* stream);

/* ------------------------------------------------------------------ *
 * Transformer decoder: ASRModel.decode / decode_spk, tal/asr/models.py:203-289,
 * ModRZTXDecoderLayer :488-528 (+ torch.nn.MultiheadAttention), PositionalEncoding
 * tal/modules.py:41-64.  Activations are batch-major [B, U, E] (rows = b*U+u).
 * ------------------------------------------------------------------ */
/* embedding -> embedding_proj (no bias; NULL when embed_size == 0) -> + pe[:U]
 * (models.py:218-223).  tokens int64 [B, U]; emb [V, E0]; proj [D, E0]; pe [max_len, D];
 * out [B, U, D].  Returns TAL_EINVAL if U > max_len (the reference asserts, system.py:341). */
int tal_embed_tokens_fwd(const int64_t* tokens, int B, int U, const float* emb, int V, int E0,
                         const float* proj, int D, const float* pe, int max_len, float* out,
                         void* stream);
/* x [B, U, D] + pe[:U] (PositionalEncoding.forward called on its own). */
int tal_add_positional_fwd(const float* x, int B, int U, int D, const float* pe, int max_len,
                           float* out, void* stream);

typedef struct tal_decoder_layer_w {
    const float* sa_in_w;   /* self_attn.in_proj_weight       [3E, E] */
    const float* sa_in_b;   /* self_attn.in_proj_bias         [3E]    */
    const float* sa_out_w;  /* self_attn.out_proj.weight      [E, E]  */
    const float* sa_out_b;  /* self_attn.out_proj.bias        [E]     */
    const float* ca_in_w;   /* multihead_attn.in_proj_weight  [3E, E] */
    const float* ca_in_b;   /* multihead_attn.in_proj_bias    [3E]    */
    const float* ca_out_w;  /* multihead_attn.out_proj.weight [E, E]  */
    const float* ca_out_b;  /* multihead_attn.out_proj.bias   [E]     */
    const float* lin1_w;    /* linear1.weight [FF, E] */
    const float* lin1_b;    /* linear1.bias   [FF]    */
    const float* lin2_w;    /* linear2.weight [E, FF] */
    const float* lin2_b;    /* linear2.bias   [E]     */
    float resweight;        /* host scalars */
    float resweight_src;
    /* FOLDED forms for the latency-oriented layer (a decode step: <= 256 prefix rows), all four or none (NULL: the layer runs its
     * eight launches).  Two pairs of dependent dense layers of ModRZTXDecoderLayer.forward (tal/asr/models.py:512-528) read the
     * same two activations, so each pair is ONE dense layer over the concatenated K axis with a pre-multiplied weight:
     *   x1 = tgt + rw (ctx Wo^T + bo)  and  q_c = s (x1 Wq^T + bq)          [s = head_dim^-0.5]
     *     = [ctx | tgt] . fold_sa_w^T + fold_sa_b,   fold_sa_w [2E, 2E] = [[rw Wo, *], [s rw Wq Wo, s Wq]]
     *   x2 = x1 + rws (ctx2 Wo2^T + bo2)  and  ff = relu(x2 W1^T + b1)
     *     = [ctx2 | x1] . fold_ca_w^T + fold_ca_b,   fold_ca_w [E + FF, 2E] = [[rws Wo2, *], [rws W1 Wo2, W1]]
     * (products formed in float64, rounded once; * = never read: the first E output columns contract over the first E inputs only
     * and the kernel adds the skip path tgt / x1 to them).  6 launches per layer instead of 8; the results differ from the unfolded layer by
     * fp32 re-association (attention rows ~1e-6). */
    const float* fold_sa_w;   /* [2E, 2E]  */
    const float* fold_sa_b;   /* [2E]      */
    const float* fold_ca_w;   /* [E + FF, 2E] */
    const float* fold_ca_b;   /* [E + FF]  */
} tal_decoder_layer_w;

/* Cross-attention key / value projections of a fixed memory window (cacheable across
 * decode steps: the memory does not change while the text prefix grows).
 * mem [B, S, E] -> k [B, S, E]; vt [B, E, S4] = V^T without bias (the bias is added after
 * P.V, exact because softmax rows sum to 1), S4 = S rounded up to a multiple of 4,
 * pad columns zero. */
int64_t tal_pad4(int64_t n);
int tal_cross_kv_fwd(const tal_decoder_layer_w* w, const float* mem, int B, int S, int E,
                     float* k, float* vt, void* stream);
size_t tal_decoder_layer_workspace_bytes(int B, int U, int S, int E, int H, int FF);
/* One ModRZTXDecoderLayer.forward:
 *   tgt += rw  * self_attn(tgt, tgt, tgt, attn_mask=tgt_mask)
 *   tgt += rws * multihead_attn(tgt, mem, mem, key_padding_mask=mem_kpm)   [weights cached]
 *   tgt += rw  * linear2(relu(linear1(tgt)))
 * tgt/out [B, U, E] (out may alias tgt); tgt_mask additive float [U, U] or NULL;
 * mem_kpm uint8 [B, S] (non-zero = ignore) or NULL; either mem [B, S, E] or a
 * (k_cache, vt_cache) pair from tal_cross_kv_fwd must be given; xattn_avg [B, U, S]
 * (head-averaged cross-attention probabilities = layer.src_attn_weights) or NULL. */
int tal_decoder_layer_fwd(const tal_decoder_layer_w* w, const float* tgt, int B, int U,
                          const float* mem, int S, int E, int H, int FF, const float* tgt_mask,
                          const uint8_t* mem_kpm, const float* k_cache, const float* vt_cache,
                          float* out, float* xattn_avg, void* workspace, size_t workspace_bytes,
                          void* stream);
/* n_layers consecutive tal_decoder_layer_fwd calls (nn.TransformerDecoder.forward, norm=None) in
 * one entry point: `layers` is an array of n_layers structs; k_cache / vt_cache are arrays of
 * n_layers device pointers (or NULL: project the memory in every layer); xattn_avg
 * [n_layers, B, U, S] or NULL; workspace as for one layer. */
int tal_decoder_stack_fwd(const tal_decoder_layer_w* layers, int n_layers, const float* tgt, int B,
                          int U, const float* mem, int S, int E, int H, int FF, const float* tgt_mask,
                          const uint8_t* mem_kpm, const float* const* k_cache,
                          const float* const* vt_cache, float* out, float* xattn_avg, void* workspace,
                          size_t workspace_bytes, void* stream);
/* Tied factorised LM head (models.py:243-246): logits = (h . P) . Emb^T with
 * proj_t = embedding_proj.weight^T stored [E0, D] (tal_transpose_fwd builds it once).
 * h rows are taken at stride ldh (ldh = U*D with h pointing at the last position gives
 * the "[:, -1]" rows system.py:124 needs without computing the other U-1).
 * proj_t may be NULL when embed_size == 0 (then D == E0).  workspace: M*E0 floats. */
int tal_lm_head_fwd(const float* h, int64_t M, int64_t ldh, int D, const float* proj_t, int E0,
                    const float* emb, int V, float* logits, void* workspace, size_t workspace_bytes,
                    void* stream);
/* y [C, R] = x [R, C]^T */
int tal_transpose_fwd(const float* x, int R, int Cc, float* y, void* stream);
/* Row-wise log_softmax of [M, N] (system.py:125,366), out may alias x. */
int tal_log_softmax_rows(const float* x, int64_t M, int N, float* out, void* stream);
/* Greedy step epilogue of System.generate_unaligned (tal/asr/system.py:355-411) in one launch and one result
 * buffer: out[0] = argmax(log_softmax(logits [V])) as int32 bits (lowest index on ties), out[1..S] = the new
 * token's cross-attention row averaged over the n_layers rows attn_rows + l * layer_stride (floats), summed in
 * layer order.  One D2H copy of 1 + S floats then carries everything the host loop steers by.  token_out
 * (may be NULL) also receives the token as int64, e.g. the next slot of a device-resident prefix. */
int tal_greedy_pick_fwd(const float* logits, int V, const float* attn_rows, int n_layers,
                        int64_t layer_stride, int S, float* out, int64_t* token_out, void* stream);
/* One whole step of System.generate_unaligned's greedy loop (tal/asr/system.py:332-411) in one call: embed the live prefix
 * tokens[history_start : n_gen] (models.py:218-223), run the n_layers decoder layers against the cached cross-attention
 * K / V^T of the current encoder window (tal_cross_kv_fwd), the tied LM head on the last position, and the pick of
 * tal_greedy_pick_fwd; the new token is appended at tokens[n_gen].  picked_dev receives {token as int32 bits, attention
 * row [S]}.  With sync != 0 the result is also delivered to picked_host (pinned host memory) and the call returns when it
 * has landed (sync == 1: copy command + stream wait; sync == 2: the pick kernel writes {token, row, sequence word} --
 * 2 + S words -- straight into the pinned buffer, which must be mapped into the device's address space as
 * hipHostMalloc memory is, and the call polls the word) -- the one entry point of this library that waits for the stream; the loop steers on the host
 * (.item() at system.py:408-411 in the reference).  Short prefixes (<= 64 tokens) run on the latency-oriented kernels
 * (8 launches per layer), longer ones on the batched-GEMM layer.  Batch 1, like the reference's loop. */
typedef struct tal_greedy_ctx {
    const tal_decoder_layer_w* layers;   /* n_layers structs */
    int32_t n_layers, E, H, FF, V, E0, S, max_len;   /* E0 = embed_size (0: no factorised embedding), S = window frames */
    const float* emb;        /* embedding.weight [V, E0 or E] (= lm_head.weight, tied) */
    const float* proj;       /* embedding_proj.weight [E, E0] or NULL */
    const float* proj_t;     /* its transpose [E0, E] (tal_transpose_fwd) or NULL */
    const float* pe;         /* pos_dec_encoder.pe [max_len, E] */
    const float* const* k_cache;    /* n_layers device pointers: K [1, S, E] of the window */
    const float* const* vt_cache;   /* n_layers device pointers: V^T [1, E, pad4(S)] */
    const uint8_t* mem_kpm;  /* [1, S] key-padding mask of the window or NULL */
    int64_t* tokens;         /* device-resident prefix buffer (capacity > n_gen) */
    void* workspace;         /* tal_greedy_step_workspace_bytes(max prefix length, ...) */
    size_t workspace_bytes;
    float* picked_dev;       /* [1 + S] */
    float* picked_host;      /* pinned host [2 + S], or NULL when sync == 0 */
    uint32_t* tickets;       /* 256 words, ZERO before the first call (the kernels leave them zero): arrival tickets of the
                              * kernels that merge partial results in-launch (key-split cross-attention, LM head + pick; words
                              * 252-254: phase counters and error word of the one-launch step, option decode_persist);
                              * NULL: the unmerged forms (more launches).  One context per stream. */
    float* picked_host_dev;  /* device alias of picked_host; NULL: resolved (hipHostGetDevicePointer) by the first sync 2 / 3 step */
    uint32_t seq;            /* library-owned: sequence value of the latest sync 2 / 3 step (start at 0) */
    uint32_t needs_reset;    /* library-owned (start at 0): a step of this context failed part-way (option decode_persist: a phase barrier gave
                              * up) and left counters in `tickets`; the next step on the context waits for the stream and zeroes the block */
    int64_t k_pitch;         /* floats between the K rows k_cache[l] points at; 0: E (a window of its own) */
    /* episode-wide K | V table (tal_greedy_set_window; all 0 / NULL: windows are projected one by one with tal_cross_kv_fwd):
     * kv_all[l] = [enc_frames, kv_pitch] floats, K in columns [0, E), V (no bias) in [E, 2E) of a frame's row -- one dense layer
     * over the whole encoder output per decoder layer, x in_proj_weight[E : 3E] with bias (in_proj_bias[E : 2E] | 0) */
    const float* const* kv_all;
    const uint8_t* kpm_all;  /* key-padding bytes of the whole episode [enc_frames] or NULL */
    int64_t enc_frames, kv_pitch;
    /* [V] floats or NULL: added to the last position's logits before the arg-max -- the LM shallow fusion of
     * System.generate_unaligned (tal/asr/system.py:368-384: logprobs[:, :n] += lm_weight * log_softmax(lm(prefix)[-1])[:n];
     * arg max (log_softmax(x) + b) = arg max (x + b)): the caller fills lm_weight * LM log-probabilities on the shared part of the
     * two vocabularies and 0 beyond it, before every step.  Honoured by tal_greedy_step_fwd and tal_greedy_step_multi_fwd; the
     * one-launch form (option decode_persist) is not taken while it is set. */
    const float* pick_bias;
    /* != 0: this session's steps never take the folded decoder layer (tal_decoder_layer_w.fold_*), whatever its prefix length.  The fold
     * pays while a step is latency-bound (a session on its own launches: -8 % per step) and costs where sessions share their launches and
     * several such chains run side by side (4 host threads x groups of 2: +5 %; a merged step of 16: +19 %: throughput-bound); a session
     * keeps ONE form for its lifetime so that its results do not depend on which other sessions happen to step beside it. */
    uint32_t no_fold;
    uint32_t _pad2;
} tal_greedy_ctx;
size_t tal_greedy_step_workspace_bytes(int U_max, int S, int E, int H, int FF, int V, int E0, int n_layers);
/* sync: 0 = enqueue only; 1 = copy {token, attention row} to picked_host and wait for the stream; 2 = the last kernel writes
 * {token, row, sequence word} into picked_host itself and the call polls the word (no copy command, no driver wake-up;
 * after 20 s without a result it waits for the stream and returns TAL_EHIP: nothing is left in flight); 3 = as 2 without
 * the polling -- the caller asks tal_greedy_step_poll, which lets ONE host thread keep several sessions (one context and
 * one stream each) in flight. */
int tal_greedy_step_fwd(tal_greedy_ctx* c, int64_t history_start, int64_t n_gen, int sync, void* stream);
/* 1: the context's latest sync 2 / 3 step has delivered; 0: not after wait_ms milliseconds (0 = one look); < 0: error -- TAL_EHIP
 * when the step delivered the failure marker (token -1: the one-launch form's phase barrier gave up); the context is then marked
 * (needs_reset) and its next step starts from a zeroed ticket block. */
int tal_greedy_step_poll(tal_greedy_ctx* c, int wait_ms);
/* The same step for G sessions (1 <= G <= 16) in SHARED launches: one chain of 34 launches (26 for sessions on the folded decoder layer) advances every session by one token
 * (the decode loop of System.generate_unaligned is batch 1 -- .item() at tal/asr/system.py:331,411,417 --, so a corpus of
 * episodes is decoded as concurrent sessions; a chain of small dependent launches per session tops out at the device's four
 * hardware queues).  Each session keeps its own context (prefix, window K / V^T, workspace, tickets, pinned result buffer);
 * every launch takes the G argument sets by value and runs the single-session kernel body per session, so a session's token,
 * attention row and hidden states are bit-identical to tal_greedy_step_fwd's.  All sessions decode with the same model and are
 * enqueued on `stream`; results are delivered as with sync == 3 (tal_greedy_step_poll per context).  A session joins a merged
 * step only while its step takes the latency-oriented kernels in the forms the merged launches use (tal_greedy_group_ok: prefix
 * <= 192 tokens, window > 64 frames, ...); step the others with tal_greedy_step_fwd. */
int tal_greedy_group_ok(const tal_greedy_ctx* c, int64_t history_start, int64_t n_gen);
/* Windows as views of an episode-wide K | V table (tal_greedy_ctx.kv_all): V^T [E, pad4(S)] of the frames [frame0, frame0 + S)
 * for n_layers (<= 8) layers in one transposing launch (pad columns zero); tal_greedy_set_window points a context's k_cache
 * (rows kv_pitch apart: the context's k_pitch must equal kv_pitch), mem_kpm and -- through that launch -- its vt_cache buffers at
 * the window [frame0, frame0 + S). */
int tal_window_vt_fwd(const float* const* kv_all, int n_layers, int64_t frame0, int S, int E, int64_t kv_pitch,
                      float* const* vt, void* stream);
int tal_greedy_set_window(tal_greedy_ctx* c, int64_t frame0, void* stream);
int tal_greedy_step_multi_fwd(tal_greedy_ctx* const* ctxs, const int64_t* history_start, const int64_t* n_gen, int G,
                              void* stream);
/* HOST side of the sliding-window loop: the decisions System.generate_unaligned takes on a step's result (tal/asr/system.py:
 * 389-521: attention centre of mass -> progress / stall counters, n-gram repetition, window shift by shift_frames with the
 * proportional cut of the text history, skip + roll-back + forced EOS on a stall or repetition, clamps, end of the episode),
 * as one call per generated token on a plain state struct -- the loop body runs ~5,700 times per hour of audio beside a 0.3 ms
 * GPU step, and in a group of sessions that share their launches the host side, not the GPU, sets the pace.
 * All pointers are HOST pointers owned by the caller. */
typedef struct tal_unaligned_state {
    int64_t* gen;              /* token stream, capacity gen_cap; gen[0 : n] are valid */
    int64_t gen_cap, n;
    int64_t history_start;     /* first token of the live prefix (the next step's input is gen[history_start : n]) */
    int64_t chunk_start;       /* first encoder frame of the next step's window (python-slice semantics: may be negative) */
    int64_t encoder_len;       /* unpadded encoder frames of the episode */
    int64_t eos, it, max_iters;
    int64_t* rec_chunk_start;  /* one record per kept token: the window start as the reference records it, ... */
    float* rec_attn;           /* ... the attention row (rec_stride floats reserved per record), ... */
    int32_t* rec_len;          /* ... and its length */
    int64_t rec_cap, n_rec;
    int32_t rec_stride;
    int32_t chunk_size, max_positions, stall_patience, rep_n;
    int32_t skip_frames;       /* int(chunk_size * skip_prct) */
    int32_t shift_frames;      /* int(chunk_size * shift_prct) */
    float del_prct;            /* float32(shift_prct / thresh_prct) */
    double thresh_prct;
    double highest_progress;
    int32_t num_no_improve, window_time;
    int32_t flags;             /* TAL_UNALIGNED_* of the latest consume, OR-ed with what the caller has not cleared */
    int32_t gen_pinned;        /* gen[] is pinned host memory: tal_unaligned_group_run uploads a rewritten prefix itself */
} tal_unaligned_state;
#define TAL_UNALIGNED_WINDOW_MOVED 1      /* chunk_start changed: re-materialise the window and its cross-attention K / V^T */
#define TAL_UNALIGNED_PREFIX_REWRITTEN 2  /* roll-back / forced EOS rewrote gen[]: upload the prefix before the next step */
#define TAL_UNALIGNED_DONE 4              /* the episode is finished */
#define TAL_UNALIGNED_GROW 8              /* gen / record capacity is used up: enlarge before the next consume */
#define TAL_UNALIGNED_ALONE 16            /* (group_run) the next step does not take the merged kernels' forms: step it alone */
/* One step's result {token, attention row [S]} -> state.  Returns the flags (>= 0) or a negative TAL_E* code. */
int tal_unaligned_consume(tal_unaligned_state* st, int64_t token, const float* attn, int S);
/* G == 1: the same loop on the session's own launches (tal_greedy_step_fwd with sync 3; any prefix length).
 * Advance G sessions (state i <-> context i) by merged steps (tal_greedy_step_multi_fwd + poll +
 * tal_unaligned_consume per session) until a session raises a flag the library cannot serve itself -- it moves a window that is
 * a view of the context's episode-wide K | V table (tal_greedy_set_window) and uploads a rewritten prefix from a pinned token
 * stream; DONE, GROW, ALONE and windows outside the table go back to the caller -- or max_steps steps are done.  Returns the number of steps
 * taken (>= 0; 0 when a session could not start: its ALONE / GROW flag says why) or a negative TAL_E* code.  dev_cap[i] =
 * capacity of ctxs[i]->tokens in tokens. */
int tal_unaligned_group_run(tal_unaligned_state* const* st, tal_greedy_ctx* const* ctxs, const int64_t* dev_cap, int G,
                            int max_steps, void* stream);
/* HOST helper (no device work, `row` is a host pointer): ngram_repeat_mask(row, n).sum() of tal/asr/util.py:5-17, the
 * repetition detector System.generate_unaligned evaluates once per generated token (system.py:418-421). */
int64_t tal_ngram_repeat_count(const int64_t* row, int64_t len, int n);
/* Beam-search candidate selection of System.generate (system.py:141-160): per batch item the
 * top-k of (logprobs[row, v] + row_score[row]) over its cur_beam rows x V tokens, rows with
 * row_done != 0 masked to -inf.  logprobs [B*cur_beam, V]; row_score / row_done [B*cur_beam]
 * (may be NULL); out_val [B, k] descending; out_idx [B, k] int64 flat index beam*V + token. */
int tal_beam_topk(const float* logprobs, const float* row_score, const uint8_t* row_done, int B,
                  int cur_beam, int V, int k, float* out_val, int64_t* out_idx, void* stream);

/* ------------------------------------------------------------------ *
 * Attention-weighted pooling of diarization features and speaker votes per generated token / word /
 * utterance: tal/utils/aligned_to_wder_format.py:150-214 (unaligned), :321-353 (aligned).  Consumes the
 * `attention` and `chunkStart` alignments of generate_unaligned together with SDModel features / ids.
 *   attn [N, S], chunk_start int64 [N], feat [T, E], ids int32 [T]
 *   window of token n = the python slice x[cs : cs + S] (negative starts wrap from the end, ends are clamped;
 *   attention is truncated to the slice length, as aw[:len(chunk)])
 *   pool: out[n] = sum_s attn[n,s] * feat[window(n)][s]; half_mode != 0 reproduces the reference's arithmetic:
 *         attention and features rounded to fp16 (`.half()`), fp32 accumulation, fp16-rounded result
 *   vote: out_id[n] = speaker id with the largest summed attention inside the window (lowest position on ties;
 *         out_weight [N] = that sum, may be NULL)
 *   vote_groups: the same vote over the tokens [group_offsets[g], group_offsets[g+1]) of each of G groups (one word,
 *         :150-196): float64 sums; half_mode rounds the attention to fp16 first (then the sums are exact and
 *         order-independent); ties go to the id whose first appearance is LAST (sorted(...)[-1] over a dict in
 *         insertion order); empty group -> -1.  ids outside [0, num_ids) are ignored; num_ids * 12 bytes of LDS.
 *   majority_vote: most frequent id in the python slice ids[ranges[2g] : ranges[2g+1]] (:330-333,
 *         Counter.most_common(1): ties go to the id that appears first); empty range -> -1.
 * ------------------------------------------------------------------ */
int tal_attn_pool_fwd(const float* attn, const int64_t* chunk_start, const float* feat, int64_t T,
                      int E, int N, int S, int half_mode, float* out, void* stream);
int tal_attn_vote_fwd(const float* attn, const int64_t* chunk_start, const int32_t* ids, int64_t T,
                      int N, int S, int32_t* out_id, float* out_weight, void* stream);
int tal_attn_vote_groups_fwd(const float* attn, const int64_t* chunk_start, const int32_t* ids, int64_t T,
                             int S, const int64_t* group_offsets, int G, int num_ids, int half_mode,
                             int32_t* out_id, double* out_weight, void* stream);
int tal_majority_vote_fwd(const int32_t* ids, int64_t T, const int64_t* ranges, int G, int num_ids,
                          int32_t* out_id, double* out_count, void* stream);

/* ------------------------------------------------------------------ *
 * GRU cell of UIS-RNN's CoreRNN, tal/diarization/uisrnn/uisrnn.py:20-39 (torch.nn.GRU,
 * gate order r, z, n).  x [B, In], h [B, H] -> h_out [B, H] (h_out != h).
 * w_ih [3H, In], w_hh [3H, H], b_ih / b_hh [3H].  The mean head (linear_mean1 -> ReLU ->
 * linear_mean2) is two tal_linear_fwd calls.
 * ------------------------------------------------------------------ */
size_t tal_gru_cell_workspace_bytes(int B, int H);
int tal_gru_cell_fwd(const float* x, const float* h, int B, int In, int H, const float* w_ih,
                     const float* w_hh, const float* b_ih, const float* b_hh, float* h_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ *
 * Measurement hooks (bench.py): when enabled, every launch of the hot kernels
 * is bracketed by two hipEvents on its launch stream.  class: 0 dense-layer
 * GEMM, 1 TDSBlock grouped conv, 2 stride-2 grouped conv, 3 log-mel, 4 other.
 * tal_prof_collect synchronises on the recorded events and returns the summed
 * duration (ms), the number of launches and the summed algorithmic work
 * (flops for 0-2, HBM bytes for 3-4) since the last tal_prof_reset.
 * A process-wide switch meant for benchmarking (slots are handed out atomically; reset / collect while launches
 * are in flight on other threads is the caller's race).
 * ------------------------------------------------------------------ */
int tal_prof_enable(int on);
int tal_prof_reset(void);
int tal_prof_collect(int cls, double* total_ms, int64_t* launches, double* total_work);

#ifdef __cplusplus
}
#endif
#endif /* TAL_ASRD_H */
