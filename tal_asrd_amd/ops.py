"""Tensor-level wrappers over the C-ABI (include/tal_asrd.h).  torch is used only
for device memory and the current HIP stream; every op below is a hand-written
HIP kernel and raises if the native library or a GPU is missing.
"""
import ctypes as C

import torch

from . import _native as N


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _f32c(t, what):
    N.require_cuda(t, what)
    if t.dtype != torch.float32:
        raise N.NativeError("%s: expected float32, got %s (only the waveform may be half precision, as the "
                            "reference's call sites hand it over; everything behind the front-end is fp32)" % (what, t.dtype))
    return t.contiguous()


# ------------------------------------------------------------------ log-mel
def logmel_plan(window, fb):
    """Device plan (Hann-folded DFT basis + sparse mel filters) for tal_logmel_fwd."""
    lib = N.lib()
    window = _f32c(window, "logmel_plan(window)")
    fb = _f32c(fb, "logmel_plan(fb)")
    if tuple(window.shape) != (400,) or tuple(fb.shape) != (201, 80):
        raise N.NativeError("logmel_plan: window/fb must be [400] and [201, 80], got %s %s"
                            % (tuple(window.shape), tuple(fb.shape)))
    plan = torch.empty(lib.tal_logmel_plan_bytes(), dtype=torch.uint8, device=window.device)
    N.check(lib.tal_logmel_plan_init(N.ptr(window), N.ptr(fb), N.ptr(plan), N.stream_handle()),
            "tal_logmel_plan_init")
    return plan


def _audio(t, what):
    """The waveform is the one tensor the reference's callers hand over in half precision (`audio_x.half()`,
    tal/asr/system.py:92,285; `x_wav.cuda().half()`, tal/baseline/reconcile.py:78): fp16 and fp32 are both taken as they
    are (the kernel widens fp16 samples while it stages them); anything else raises."""
    N.require_cuda(t, what)
    if t.dtype not in (torch.float32, torch.float16):
        raise N.NativeError("%s: the waveform must be float32 or float16, got %s" % (what, t.dtype))
    return t.contiguous()


def logmel(plan, audio, eps=1e-6, subtract_mean=True, return_stats=False):
    """audio [B, L] fp32 or fp16 -> [B, T, 80] fp32 (LogMelSpec.forward, tal/asr/models.py:35-53)."""
    lib = N.lib()
    audio = _audio(audio, "logmel")
    if audio.dim() != 2:
        raise N.NativeError("logmel: audio must be [batch, samples]")
    B, L = audio.shape
    T = lib.tal_logmel_num_frames(L)
    out = torch.empty(B, T, 80, dtype=torch.float32, device=audio.device)
    mean = torch.empty(1, dtype=torch.float32, device=audio.device)
    stats = torch.empty(2, dtype=torch.float64, device=audio.device)
    nws = lib.tal_logmel_workspace_bytes(B, L)
    ws = _ws(nws, audio.device)
    fwd = lib.tal_logmel_f16_fwd if audio.dtype == torch.float16 else lib.tal_logmel_fwd
    N.check(fwd(N.ptr(plan), N.ptr(audio), B, L, eps, 1 if subtract_mean else 0, N.ptr(out),
                N.ptr(mean), N.ptr(stats), N.ptr(ws), nws, N.stream_handle()), "tal_logmel_fwd")
    return (out, mean, stats) if return_stats else out


def subtract_scalar_(x, mean):
    lib = N.lib()
    N.check(lib.tal_subtract_scalar(N.ptr(x), x.numel(), N.ptr(mean), N.stream_handle()), "tal_subtract_scalar")
    return x


# ------------------------------------------------------------------ dense
def linear(x, weight, bias=None, mode=0, res=None, alpha=0.0, out=None):
    """y = epilogue(x . W^T + b) over the last dim; weight [N, K] (nn.Linear / 1x1 Conv1d layout)."""
    lib = N.lib()
    x = _f32c(x, "linear")
    w2 = weight.reshape(weight.shape[0], -1)
    w2 = _f32c(w2, "linear(weight)")
    K = x.shape[-1]
    Nout = w2.shape[0]
    if w2.shape[1] != K:
        raise N.NativeError("linear: x[..., %d] vs weight %s" % (K, tuple(weight.shape)))
    M = x.numel() // K
    y = out if out is not None else torch.empty(*x.shape[:-1], Nout, dtype=torch.float32, device=x.device)
    if res is not None:
        res = _f32c(res, "linear(res)")
    if bias is not None:
        bias = _f32c(bias, "linear(bias)")
    nws = lib.tal_linear_workspace_bytes(M, Nout, K)
    if nws:
        ws = _ws(nws, x.device)    # (torch's caching allocator hands the same block back call after call)
        N.check(lib.tal_linear_ws_fwd(N.ptr(x), N.ptr(w2), N.ptr(bias), N.ptr(res), float(alpha), int(mode), M, Nout, K,
                                      N.ptr(y), N.ptr(ws), nws, N.stream_handle()), "tal_linear_ws_fwd")
    else:
        N.check(lib.tal_linear_fwd(N.ptr(x), N.ptr(w2), N.ptr(bias), N.ptr(res), float(alpha), int(mode), M, Nout, K,
                                   N.ptr(y), N.stream_handle()), "tal_linear_fwd")
    return y


def split_f16x3(x2d):
    """fp32 [rows, K] (K % 32 == 0) -> the hi / lo fp16 split the fp16x3 dense layers consume (same number of bytes,
    returned as an opaque uint8 tensor)."""
    lib = N.lib()
    x2d = _f32c(x2d, "split_f16x3")
    rows, K = x2d.shape
    out = torch.empty(rows * K * 4, dtype=torch.uint8, device=x2d.device)
    N.check(lib.tal_split_f16x3_fwd(N.ptr(x2d), N.ptr(out), rows, K, N.stream_handle()), "tal_split_f16x3_fwd")
    return out


# ------------------------------------------------------------------ grouped conv
def pack_gconv_weight(weight, groups):
    """reference Conv1d weight [C_out, C_in/G, 21] -> packed [G][C_in/G][21][C_out/G]."""
    lib = N.lib()
    w = _f32c(weight, "pack_gconv_weight")
    c_out, cig, ks = w.shape
    packed = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    N.check(lib.tal_pack_gconv_weight(N.ptr(w), N.ptr(packed), c_out, cig, ks, groups, N.stream_handle()),
            "tal_pack_gconv_weight")
    return packed


def gconv_s2(x, w_packed, bias, c_out, groups):
    """x [B, T, C_in] -> [B, (T-21)//2+1, C_out] (tal/asr/models.py:363-364)."""
    lib = N.lib()
    x = _f32c(x, "gconv_s2")
    B, T, c_in = x.shape
    y = torch.empty(B, (T - 21) // 2 + 1, c_out, dtype=torch.float32, device=x.device)
    N.check(lib.tal_gconv_s2_fwd(N.ptr(x), N.ptr(w_packed), N.ptr(bias), B, T, c_in, c_out, groups, N.ptr(y),
                                 N.stream_handle()), "tal_gconv_s2_fwd")
    return y


def gconv_res(x, w_packed, bias, alpha, groups):
    """x + alpha * relu(gconv21(x)) on [B, T, C] (tal/asr/models.py:304-308,329)."""
    lib = N.lib()
    x = _f32c(x, "gconv_res")
    B, T, c = x.shape
    y = torch.empty_like(x)
    N.check(lib.tal_gconv_res_fwd(N.ptr(x), N.ptr(w_packed), N.ptr(bias), float(alpha), B, T, c, groups, N.ptr(y),
                                  N.stream_handle()), "tal_gconv_res_fwd")
    return y


def pack_gconv_f16x3_weight(weight, groups, stride=1):
    """reference Conv1d weight [C_out, C_in/G, 21] -> fp16x3 MFMA operand fragments (opaque uint8 tensor), or None when
    the shape has no matrix-core kernel (tal_gconv_f16x3_weight_bytes == 0)."""
    lib = N.lib()
    w = _f32c(weight, "pack_gconv_f16x3_weight")
    c_out, cig, ks = w.shape
    c_in = cig * groups
    nbytes = lib.tal_gconv_f16x3_weight_bytes(c_in, c_out, groups, stride) if ks == 21 else 0
    if nbytes == 0:
        return None
    frag = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    N.check(lib.tal_pack_gconv_f16x3_weight(N.ptr(w), N.ptr(frag), c_in, c_out, groups, stride, N.stream_handle()),
            "tal_pack_gconv_f16x3_weight")
    return frag


def gconv_s2_f16x3(x, w_frag, bias, c_out, groups):
    """x [B, T, C_in] -> [B, (T-21)//2+1, C_out] (tal/asr/models.py:363-364) on the matrix cores (fp16x3 form)."""
    lib = N.lib()
    x = _f32c(x, "gconv_s2_f16x3")
    B, T, c_in = x.shape
    y = torch.empty(B, (T - 21) // 2 + 1, c_out, dtype=torch.float32, device=x.device)
    N.check(lib.tal_gconv_s2_f16x3_fwd(N.ptr(x), N.ptr(w_frag), N.ptr(bias), B, T, c_in, c_out, groups, N.ptr(y),
                                       N.stream_handle()), "tal_gconv_s2_f16x3_fwd")
    return y


def gconv_res_f16x3(x, w_frag, bias, alpha, groups, want_split=False):
    """x + alpha * relu(gconv21(x)) on [B, T, C] on the matrix cores (fp16x3 form); with want_split also returns the
    result as the hi / lo split the fp16x3 dense layers consume (split_f16x3's format)."""
    lib = N.lib()
    x = _f32c(x, "gconv_res_f16x3")
    B, T, c = x.shape
    y = torch.empty_like(x)
    ys = torch.empty(B * T * c * 4, dtype=torch.uint8, device=x.device) if want_split else None
    N.check(lib.tal_gconv_res_f16x3_fwd(N.ptr(x), N.ptr(w_frag), N.ptr(bias), float(alpha), B, T, c, groups, N.ptr(y),
                                        N.ptr(ys) if want_split else None, N.stream_handle()), "tal_gconv_res_f16x3_fwd")
    return (y, ys) if want_split else y


def gconv_res_split(x_split, shape, w_frag, bias, alpha, groups):
    """TDSBlock conv on activations in the hi / lo split form: x_split (opaque uint8 tensor, split_f16x3's format) of
    logical shape [B, T, C] -> the split form of x + alpha * relu(gconv21(x))."""
    lib = N.lib()
    B, T, c = shape
    ys = torch.empty(B * T * c * 4, dtype=torch.uint8, device=x_split.device)
    N.check(lib.tal_gconv_res_split_fwd(N.ptr(x_split), N.ptr(w_frag), N.ptr(bias), float(alpha), B, T, c, groups, N.ptr(ys),
                                        N.stream_handle()), "tal_gconv_res_split_fwd")
    return ys


def gconv_s2_split(x, shape, x_is_split, w_frag, bias, c_out, groups):
    """Stride-2 resize conv writing the split form of its output; x is fp32 [B, T, C_in] or the split form of it."""
    lib = N.lib()
    B, T, c_in = shape
    t_out = (T - 21) // 2 + 1
    ys = torch.empty(B * t_out * c_out * 4, dtype=torch.uint8, device=x.device)
    N.check(lib.tal_gconv_s2_split_fwd(N.ptr(x), 1 if x_is_split else 0, N.ptr(w_frag), N.ptr(bias), B, T, c_in, c_out, groups,
                                       N.ptr(ys), N.stream_handle()), "tal_gconv_s2_split_fwd")
    return ys


# ------------------------------------------------------------------ TDS driver
range_fallbacks = 0      # calls that were re-run on the exact fp32 kernels because an activation left the fp16 range


class RangeCheck:
    """Pending fp16-range check of one tal_tds_fwd call: `flagged()` reads the call's status word (a 4-byte device-to-host
    copy, i.e. a wait for the stream), `rerun_exact()` repeats the call on the exact fp32-input kernels into the same
    output tensor.  Deferring the read lets the caller enqueue the kernels that consume the encoder output first."""

    def __init__(self, desc, x, y, ws, nws, off, x_mean=None, y_split=False):
        self.desc, self.x, self.y, self.ws, self.nws, self.off, self.x_mean = desc, x, y, ws, nws, off, x_mean
        self.y_split = y_split          # y holds the hi / lo split form (tds_forward(out_split=True)); an exact re-run writes fp32

    def flagged(self):
        """One 8-byte read of the call's status block: word 0 = an activation left the fp16 range; word 1 = the form the call
        wrote y in, which must be the form `y_split` promised the consumer (tal_tds_out_split is a prediction: another thread's
        tal_set_option between the query and the call, or a misaligned x, changes what the call does -- never silently)."""
        if self.desc.flags & N.TAL_TDS_EXACT_F32 and not self.y_split:
            return False
        flag, form = self.ws[self.off:self.off + 8].view(torch.int32).tolist()
        if bool(form) != bool(self.y_split):
            raise N.NativeError("tal_tds_fwd wrote its output in the %s form but its consumer was enqueued for the %s form "
                                "(a kernel-selection option changed between tal_tds_out_split and the call?)"
                                % ("split" if form else "fp32", "split" if self.y_split else "fp32"))
        return flag != 0 and not (self.desc.flags & N.TAL_TDS_EXACT_F32)

    def rerun_exact(self):
        global range_fallbacks
        range_fallbacks += 1
        lib = N.lib()
        B, T, _ = self.x.shape
        exact = N.TdsDesc.from_buffer_copy(self.desc)       # (a copy: the cached descriptor may be in use by another thread's call)
        exact.flags |= N.TAL_TDS_EXACT_F32
        N.check(_tds_call(lib, exact, self.x, self.x_mean, B, T, self.y, self.ws, self.nws), "tal_tds_fwd (exact fp32 re-run)")
        self.y_split = False
        return self.y


def _tds_call(lib, desc, x, x_mean, B, T, y, ws, nws):
    if x_mean is None:
        return lib.tal_tds_fwd(C.byref(desc), N.ptr(x), B, T, N.ptr(y), N.ptr(ws), nws, N.stream_handle())
    return lib.tal_tds_premean_fwd(C.byref(desc), N.ptr(x), N.ptr(x_mean), B, T, N.ptr(y), N.ptr(ws), nws, N.stream_handle())


def tds_premean_ok(desc, x):
    """May `x` be handed to tds_forward as a log-mel BEFORE its mean subtraction (x_mean=...)?  (tal_tds_premean_ok)"""
    return bool(N.lib().tal_tds_premean_ok(C.byref(desc), N.ptr(x)))


def tds_forward(desc, x, c_out, check_range=True, defer=False, x_mean=None, out_split=False):
    """x [B, T, C0] -> [B, T', C_last] through tal_tds_fwd (whole encoder, one C call).

    x_mean (device tensor [1]): x is the log-mel before LogMelSpec's global-mean subtraction and x_mean the scalar to subtract
    (logmel(..., subtract_mean=False, return_stats=True)); the subtraction is folded into the first resize conv's bias
    (tal_tds_premean_fwd).  Only where tds_premean_ok(desc, x).

    out_split=True (with defer=True): the caller's consumer takes the hi / lo split form (sd_head(x_split=True)); where the last
    stage runs all-split the output tensor then holds that form -- same shape and bytes, NOT fp32 values -- and the returned
    RangeCheck says so (`.y_split`).

    fp16-range guard: the long-input layers run in the fp16x3 form (fp32 values as two fp16 halves), which needs
    |activation| <= 65504.  The kernels raise a status word when a value was out of range; the call is then repeated
    on the exact fp32-input kernels (one small device-to-host read per call; check_range=False skips it).
    defer=True returns (y, RangeCheck) and leaves the read to the caller (after it has enqueued the consumers of y)."""
    lib = N.lib()
    x = _f32c(x, "tds_forward")
    B, T, _ = x.shape
    t_out = lib.tal_tds_out_len(C.byref(desc), T)
    if t_out <= 0:
        raise N.NativeError("tds_forward: %d frames are too few for the stride-2 k=21 stages" % T)
    y = torch.empty(B, t_out, c_out, dtype=torch.float32, device=x.device)
    nws = lib.tal_tds_workspace_bytes(C.byref(desc), B, T)
    ws = _ws(nws, x.device)
    y_split = False
    if out_split:
        if not defer:
            raise N.NativeError("tds_forward: out_split needs defer=True (the caller must look at RangeCheck.y_split)")
        asked = N.TdsDesc.from_buffer_copy(desc)
        asked.flags |= N.TAL_TDS_OUT_SPLIT
        if lib.tal_tds_out_split(C.byref(asked), B, T):
            desc, y_split = asked, True
    N.check(_tds_call(lib, desc, x, x_mean, B, T, y, ws, nws), "tal_tds_fwd")
    chk = RangeCheck(desc, x, y, ws, nws, lib.tal_tds_status_offset(C.byref(desc), B, T), x_mean, y_split)
    if defer:
        return y, chk
    if check_range and chk.flagged():
        chk.rerun_exact()
    return y


def tds_forward_tiled(desc, x, c_out, out_tile, check_range=True):
    """x [1, T, C0] -> [1, T', C_last] through tal_tds_tiled_fwd: tiles of `out_tile` output frames, each with its
    receptive-field halo (include/tal_asrd.h); same range guard as tds_forward."""
    lib = N.lib()
    x = _f32c(x, "tds_forward_tiled")
    if x.dim() != 3 or x.shape[0] != 1:
        raise N.NativeError("tds_forward_tiled: x must be [1, T, C]")
    T = x.shape[1]
    t_out = lib.tal_tds_out_len(C.byref(desc), T)
    if t_out <= 0:
        raise N.NativeError("tds_forward_tiled: %d frames are too few for the stride-2 k=21 stages" % T)
    y = torch.empty(1, t_out, c_out, dtype=torch.float32, device=x.device)
    nws = lib.tal_tds_tiled_workspace_bytes(C.byref(desc), T, int(out_tile))
    ws = _ws(nws, x.device)

    def run(d):
        N.check(lib.tal_tds_tiled_fwd(C.byref(d), N.ptr(x), T, N.ptr(y), int(out_tile), N.ptr(ws), nws, N.stream_handle()),
                "tal_tds_tiled_fwd")
    run(desc)
    if check_range and not (desc.flags & N.TAL_TDS_EXACT_F32):
        off = lib.tal_tds_tiled_status_offset(C.byref(desc), T, int(out_tile))
        if int(ws[off:off + 4].view(torch.int32)[0]) != 0:
            global range_fallbacks
            range_fallbacks += 1
            exact = N.TdsDesc.from_buffer_copy(desc)
            exact.flags |= N.TAL_TDS_EXACT_F32
            run(exact)
    return y


# ------------------------------------------------------------------ diarization head
def sd_head(x, w_embed, b_embed, w_logit, b_logit, want_logits=True, want_ids=True, x_split=False, w_embed_split=None):
    """x [..., C] -> (feat [..., E], logits [..., S] | None, ids [...] int32 | None).
    x_split=True: x holds the hi / lo split form (tds_forward(out_split=True) with RangeCheck.y_split) and w_embed_split the
    split of w_embed (split_f16x3): the embedding layer runs in the fp16x3 form (tal_sd_head_split_fwd)."""
    lib = N.lib()
    x = _f32c(x, "sd_head")
    Cc = x.shape[-1]
    M = x.numel() // Cc
    E, S = w_embed.shape[0], w_logit.shape[0]
    dev = x.device
    feat = torch.empty(*x.shape[:-1], E, dtype=torch.float32, device=dev)
    logits = torch.empty(*x.shape[:-1], S, dtype=torch.float32, device=dev) if want_logits else None
    ids = torch.empty(x.shape[:-1], dtype=torch.int32, device=dev) if want_ids else None
    nws = lib.tal_sd_head_workspace_bytes(M, S) if (want_ids and not want_logits) else 0
    ws = _ws(nws, dev)
    if x_split:
        if w_embed_split is None:
            raise N.NativeError("sd_head: x_split needs w_embed_split (ops.split_f16x3 of the embedding weight)")
        if not nws:
            nws = lib.tal_sd_head_workspace_bytes(M, S)
            ws = _ws(nws, dev)
        N.check(lib.tal_sd_head_split_fwd(N.ptr(x), M, Cc, N.ptr(w_embed_split), N.ptr(b_embed), E,
                                          N.ptr(_f32c(w_logit, "w")), N.ptr(b_logit), S, N.ptr(feat), N.ptr(logits),
                                          N.ptr(ids), N.ptr(ws), nws, N.stream_handle()), "tal_sd_head_split_fwd")
        return feat, logits, ids
    N.check(lib.tal_sd_head_fwd(N.ptr(x), M, Cc, N.ptr(_f32c(w_embed, "w")), N.ptr(b_embed), E,
                                N.ptr(_f32c(w_logit, "w")), N.ptr(b_logit), S, N.ptr(feat), N.ptr(logits),
                                N.ptr(ids), N.ptr(ws), nws, N.stream_handle()), "tal_sd_head_fwd")
    return feat, logits, ids


def argmax_rows(x):
    lib = N.lib()
    x = _f32c(x, "argmax_rows")
    Nn = x.shape[-1]
    M = x.numel() // Nn
    ids = torch.empty(x.shape[:-1], dtype=torch.int32, device=x.device)
    N.check(lib.tal_argmax_rows(N.ptr(x), M, Nn, N.ptr(ids), N.stream_handle()), "tal_argmax_rows")
    return ids


def add_positional(x, pe):
    """x [B, U, D] + pe[:U] (PositionalEncoding.forward, tal/modules.py:63)."""
    lib = N.lib()
    x = _f32c(x, "add_positional")
    B, U, D = x.shape
    out = torch.empty_like(x)
    N.check(lib.tal_add_positional_fwd(N.ptr(x), B, U, D, N.ptr(_f32c(pe, "pe")), pe.shape[0], N.ptr(out),
                                       N.stream_handle()), "tal_add_positional_fwd")
    return out
