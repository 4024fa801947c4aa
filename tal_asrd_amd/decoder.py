"""Host side of the transformer decoder path: ASRModel.decode / decode_spk
(tal/asr/models.py:203-289) and ModRZTXDecoderLayer.forward (:512-528) on the
C-ABI kernels (tal_embed_tokens_fwd, tal_decoder_layer_fwd, tal_lm_head_fwd).

Layout: the reference runs the stack sequence-major ([U, B, E], torch's native
transformer layout) and permutes in and out (models.py:212,227,241); the kernels
run batch-major [B, U, E], which is what decode() receives and returns, so those
permutes do not exist here.  `ModRZTXDecoderLayer.forward` keeps the [U, B, E]
contract for callers that use a layer directly.

Caches (pure functions of parameters / memory tensors, invalidated by
data_ptr + version):
  * embedding_proj.weight^T for the tied factorised LM head;
  * per-layer cross-attention K and V^T of a memory window -- the decode loops
    call decode() hundreds of times against the same encoder window while only
    the text prefix grows (system.py:113,350), the memory projections
    (S x 2 x E^2 MACs per layer) are computed once per window.
"""
import ctypes as C

import torch

from . import _native as N
from . import ops


def _layer_struct(layer):
    w = N.DecoderLayerW()
    sa, ca = layer.self_attn, layer.multihead_attn
    w.sa_in_w, w.sa_in_b = sa.in_proj_weight.data_ptr(), sa.in_proj_bias.data_ptr()
    w.sa_out_w, w.sa_out_b = sa.out_proj.weight.data_ptr(), sa.out_proj.bias.data_ptr()
    w.ca_in_w, w.ca_in_b = ca.in_proj_weight.data_ptr(), ca.in_proj_bias.data_ptr()
    w.ca_out_w, w.ca_out_b = ca.out_proj.weight.data_ptr(), ca.out_proj.bias.data_ptr()
    w.lin1_w, w.lin1_b = layer.linear1.weight.data_ptr(), layer.linear1.bias.data_ptr()
    w.lin2_w, w.lin2_b = layer.linear2.weight.data_ptr(), layer.linear2.bias.data_ptr()
    w.resweight = float(layer.resweight.detach())
    w.resweight_src = float(layer.resweight_src.detach())
    folds = _folded_weights(layer)
    if folds is not None:
        w.fold_sa_w, w.fold_sa_b, w.fold_ca_w, w.fold_ca_b = (t.data_ptr() for t in folds)
    layer.__dict__["_tal_folds"] = folds          # (kept alive beside the struct that points at them)
    return w


FOLD_DECODER_LAYERS = True      # False: structs without the folded forms (every decode step on the eight-launch layer)


def _folded_weights(layer):
    """The two folded dense layers of include/tal_asrd.h (tal_decoder_layer_w.fold_*): products in float64, rounded once."""
    if not FOLD_DECODER_LAYERS:
        return None
    sa, ca = layer.self_attn, layer.multihead_attn
    E = sa.embed_dim
    if E % 256 != 0 or layer.linear1.out_features % 64 != 0:      # (the two-segment dense kernels: K slices of up to 256 per wave)
        return None
    d = torch.float64
    s = float(E // sa.num_heads) ** -0.5
    rw, rws = float(layer.resweight.detach()), float(layer.resweight_src.detach())
    Wo, bo = sa.out_proj.weight.detach().to(d), sa.out_proj.bias.detach().to(d)
    Wq, bq = ca.in_proj_weight.detach()[:E].to(d), ca.in_proj_bias.detach()[:E].to(d)
    Wo2, bo2 = ca.out_proj.weight.detach().to(d), ca.out_proj.bias.detach().to(d)
    W1, b1 = layer.linear1.weight.detach().to(d), layer.linear1.bias.detach().to(d)
    eye = torch.zeros(E, E, dtype=d, device=Wo.device)      # (the skip path's half is never read: the kernel adds tgt / x1 itself)
    # (the layer scales q with a single fp32 multiply by s after the bias; here s rides in the weights)
    s32 = float(torch.tensor(s, dtype=torch.float32))
    f_sa_w = torch.cat([torch.cat([rw * Wo, eye], 1), torch.cat([s32 * rw * (Wq @ Wo), s32 * Wq], 1)], 0)
    f_sa_b = torch.cat([rw * bo, s32 * (rw * (Wq @ bo) + bq)])
    f_ca_w = torch.cat([torch.cat([rws * Wo2, eye], 1), torch.cat([rws * (W1 @ Wo2), W1], 1)], 0)
    f_ca_b = torch.cat([rws * bo2, rws * (W1 @ bo2) + b1])
    return tuple(t.to(torch.float32).contiguous() for t in (f_sa_w, f_sa_b, f_ca_w, f_ca_b))


def layer_weights(layer):
    """Cached tal_decoder_layer_w for a ModRZTXDecoderLayer (rebuilt when a parameter changes)."""
    # (walking the module tree costs ~60 us per layer and call -- 8 % of an episode's decode at one call per layer and encoder
    #  window; the cached list is rebuilt whenever a module of this layer registered a Parameter since, models.param_epoch)
    from .models import param_epoch
    epoch = param_epoch(layer)
    plist = layer.__dict__.get("_tal_plist")
    # (a conversion in torch's overwrite-parameters mode replaces the Parameter objects without registering them: the first one
    #  is compared by identity)
    if plist is None or plist[0] != epoch or plist[1][0] is not next(iter(layer.parameters()), None):
        plist = (epoch, list(layer.parameters()))
        layer.__dict__["_tal_plist"] = plist
    key = tuple((p.data_ptr(), p._version) for p in plist[1])
    cached = getattr(layer, "_tal_w", None)
    if cached is None or cached[0] != key:
        for p in layer.parameters():
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise N.NativeError("decoder layer parameters must be contiguous fp32 CUDA tensors")
        cached = (key, _layer_struct(layer))
        layer._tal_w = cached
    return cached[1]


def _kpm_u8(mask, B, S, device):
    if mask is None:
        return None
    m = mask.to(device=device)
    if tuple(m.shape) != (B, S):
        raise N.NativeError("memory_key_padding_mask must be [batch, src_len] = %s, got %s" % ((B, S), tuple(m.shape)))
    return m.to(torch.uint8).contiguous()


def _check_memory(memory, B, E, what):
    """The kernels stride the memory / its K, V^T caches per batch item of `tgt`: a memory with another batch size
    (e.g. `speaker_out`, which the reference's beam search does not repeat, system.py:168-171) would be read out
    of bounds, where the reference raises a shape error inside nn.MultiheadAttention."""
    if memory.dim() != 3 or memory.shape[0] != B or memory.shape[2] != E:
        raise N.NativeError("%s: memory must be [batch=%d, src_len, d_model=%d], got %s"
                            % (what, B, E, tuple(memory.shape)))


def cross_kv(layer, memory):
    """(K [B,S,E], V^T [B,E,S4]) of `memory` for this layer's cross-attention, cached per memory tensor."""
    lib = N.lib()
    key = (memory.data_ptr(), memory._version, tuple(memory.shape),
           layer.multihead_attn.in_proj_weight.data_ptr(), layer.multihead_attn.in_proj_weight._version,
           layer.multihead_attn.in_proj_bias.data_ptr(), layer.multihead_attn.in_proj_bias._version)
    cached = getattr(layer, "_tal_kv", None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    B, S, E = memory.shape
    k = torch.empty(B, S, E, dtype=torch.float32, device=memory.device)
    vt = torch.empty(B, E, lib.tal_pad4(S), dtype=torch.float32, device=memory.device)
    N.check(lib.tal_cross_kv_fwd(C.byref(layer_weights(layer)), N.ptr(memory), B, S, E, N.ptr(k), N.ptr(vt),
                                 N.stream_handle()), "tal_cross_kv_fwd")
    layer._tal_kv = (key, k, vt, memory)  # keep `memory` alive so its data_ptr cannot be recycled
    return k, vt


def run_layer(layer, tgt, memory, tgt_mask=None, kpm=None, want_weights=True, cache_kv=True):
    """One decoder layer on batch-major tensors: tgt [B,U,E], memory [B,S,E] -> (out [B,U,E], weights [B,U,S])."""
    lib = N.lib()
    tgt = ops._f32c(tgt, "decoder layer (tgt)")
    memory = ops._f32c(memory, "decoder layer (memory)")
    B, U, E = tgt.shape
    _check_memory(memory, B, E, "decoder layer")
    S = memory.shape[1]
    H = layer.nhead
    FF = layer.linear1.out_features
    w = layer_weights(layer)
    out = torch.empty_like(tgt)
    avg = torch.empty(B, U, S, dtype=torch.float32, device=tgt.device) if want_weights else None
    k = vt = None
    if cache_kv:
        k, vt = cross_kv(layer, memory)
    if tgt_mask is not None:
        tgt_mask = ops._f32c(tgt_mask.to(tgt.device), "tgt_mask")
        if tuple(tgt_mask.shape) != (U, U):
            raise N.NativeError("tgt_mask must be [tgt_len, tgt_len]")
    nws = lib.tal_decoder_layer_workspace_bytes(B, U, S, E, H, FF)
    ws = ops._ws(nws, tgt.device)
    N.check(lib.tal_decoder_layer_fwd(C.byref(w), N.ptr(tgt), B, U, N.ptr(memory), S, E, H, FF, N.ptr(tgt_mask),
                                      N.ptr(kpm), N.ptr(k), N.ptr(vt), N.ptr(out), N.ptr(avg), N.ptr(ws), nws,
                                      N.stream_handle()), "tal_decoder_layer_fwd")
    return out, avg


def decoder_layer_forward(layer, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                          memory_key_padding_mask=None):
    """ModRZTXDecoderLayer.forward with the reference's sequence-major contract:
    tgt [U,B,E], memory [S,B,E] -> [U,B,E]; sets layer.src_attn_weights [B,U,S]."""
    if memory_mask is not None or tgt_key_padding_mask is not None:
        raise N.NativeError("memory_mask / tgt_key_padding_mask are never used by the reference (models.py:237-238) "
                            "and are not built")
    N.require_cuda(tgt, "ModRZTXDecoderLayer.forward")
    t = tgt.permute(1, 0, 2).contiguous()
    m = memory.permute(1, 0, 2).contiguous()
    kpm = _kpm_u8(memory_key_padding_mask, m.shape[0], m.shape[1], t.device)
    out, avg = run_layer(layer, t, m, tgt_mask, kpm, want_weights=True, cache_kv=False)
    layer.src_attn_weights = avg.detach()
    return out.permute(1, 0, 2).contiguous()


def _embed(model, y_prev, check_tokens=True):
    """embedding -> embedding_proj -> + pe (models.py:218-223) in one kernel."""
    lib = N.lib()
    N.require_cuda(y_prev, "ASRModel.decode(y_prev)")
    y = y_prev.to(torch.int64).contiguous()
    B, U = y.shape
    emb = model.embedding.weight
    V, E0 = emb.shape
    proj = model.embedding_proj.weight if model.embed_size else None
    D = proj.shape[0] if proj is not None else E0
    pe = model.pos_dec_encoder.pe
    # nn.Embedding raises on out-of-range ids; the kernel clamps (never reads out of bounds) and the
    # range is checked here unless the caller already knows it (decode loops: ids come from an arg-max)
    if check_tokens and (int(y.min()) < 0 or int(y.max()) >= V):
        raise IndexError("token id out of range [0, %d)" % V)
    out = torch.empty(B, U, D, dtype=torch.float32, device=y.device)
    N.check(lib.tal_embed_tokens_fwd(N.ptr(y), B, U, N.ptr(emb), V, E0, N.ptr(proj), D, N.ptr(pe), pe.shape[0],
                                     N.ptr(out), N.stream_handle()), "tal_embed_tokens_fwd")
    return out


def causal_mask(n, device):
    """triu(ones, 1) -> -inf (models.py:229-235)."""
    m = torch.triu(torch.ones(n, n), 1)
    return m.masked_fill(m == 1, float("-inf")).to(device)


def _stack_structs(stack):
    """Contiguous array of tal_decoder_layer_w for a whole stack (cached; rebuilt when any parameter changes)."""
    key = tuple((p.data_ptr(), p._version) for p in stack.parameters())
    cached = getattr(stack, "_tal_stack", None)
    if cached is None or cached[0] != key:
        arr = (N.DecoderLayerW * len(stack.layers))()
        for i, layer in enumerate(stack.layers):
            arr[i] = layer_weights(layer)
        cached = (key, arr)
        stack._tal_stack = cached
    return cached[1]


def _stack_kv(stack, memory):
    """Per-layer cached cross-attention K / V^T pointer arrays for `memory`."""
    key = (memory.data_ptr(), memory._version, tuple(memory.shape)) + tuple(
        (l.multihead_attn.in_proj_weight.data_ptr(), l.multihead_attn.in_proj_weight._version,
         l.multihead_attn.in_proj_bias.data_ptr(), l.multihead_attn.in_proj_bias._version) for l in stack.layers)
    cached = getattr(stack, "_tal_stack_kv", None)
    if cached is None or cached[0] != key:
        ks, vts = [], []
        for layer in stack.layers:
            k, vt = cross_kv(layer, memory)
            ks.append(k)
            vts.append(vt)
        n = len(ks)
        karr = (C.c_void_p * n)(*[k.data_ptr() for k in ks])
        varr = (C.c_void_p * n)(*[v.data_ptr() for v in vts])
        cached = (key, karr, varr, ks, vts, memory)
        stack._tal_stack_kv = cached
    return cached[1], cached[2]


def stack_kv_private(stack, memory):
    """The same K / V^T pointer arrays, owned by the caller: (karr, varr, keep-alive list).  The cached form above keeps ONE
    window per stack; several decode sessions over the same weights (System.transcribe_unaligned_many) each hold their own."""
    lib = N.lib()
    B, S, E = memory.shape
    ks, vts = [], []
    for layer in stack.layers:
        k = torch.empty(B, S, E, dtype=torch.float32, device=memory.device)
        vt = torch.empty(B, E, lib.tal_pad4(S), dtype=torch.float32, device=memory.device)
        N.check(lib.tal_cross_kv_fwd(C.byref(layer_weights(layer)), N.ptr(memory), B, S, E, N.ptr(k), N.ptr(vt),
                                     N.stream_handle()), "tal_cross_kv_fwd")
        ks.append(k)
        vts.append(vt)
    n = len(ks)
    karr = (C.c_void_p * n)(*[k.data_ptr() for k in ks])
    varr = (C.c_void_p * n)(*[v.data_ptr() for v in vts])
    return karr, varr, (ks, vts, memory)


def _run_stack(model, stack, y_prev, memory, mask, causal, check_tokens=True):
    """embedding -> all decoder layers (one C call) -> hidden [B,U,E]; sets layer.src_attn_weights."""
    lib = N.lib()
    h = _embed(model, y_prev, check_tokens)
    memory = ops._f32c(memory, "decode(memory)")
    B, U, E = h.shape
    _check_memory(memory, B, E, "decode")
    S = memory.shape[1]
    kpm = _kpm_u8(mask, B, S, h.device)
    tm = causal_mask(U, h.device) if causal else None
    layers = stack.layers
    n = len(layers)
    H, FF = layers[0].nhead, layers[0].linear1.out_features
    arr = _stack_structs(stack)
    karr, varr = _stack_kv(stack, memory)
    out = torch.empty_like(h)
    avg = torch.empty(n, B, U, S, dtype=torch.float32, device=h.device)
    nws = lib.tal_decoder_layer_workspace_bytes(B, U, S, E, H, FF)
    ws = ops._ws(nws, h.device)
    N.check(lib.tal_decoder_stack_fwd(arr, n, N.ptr(h), B, U, N.ptr(memory), S, E, H, FF, N.ptr(tm), N.ptr(kpm),
                                      karr, varr, N.ptr(out), N.ptr(avg), N.ptr(ws), nws, N.stream_handle()),
            "tal_decoder_stack_fwd")
    for i, layer in enumerate(layers):
        layer.src_attn_weights = avg[i]
    stack.src_attn_weights_all = avg   # [n_layers, B, U, S], one tensor for the decode loops
    return out


def _proj_t(model):
    """embedding_proj.weight^T [E0, D], cached (F.linear(h, W.t()) at models.py:244)."""
    lib = N.lib()
    wgt = model.embedding_proj.weight
    key = (wgt.data_ptr(), wgt._version)
    cached = getattr(model, "_tal_proj_t", None)
    if cached is None or cached[0] != key:
        D, E0 = wgt.shape
        t = torch.empty(E0, D, dtype=torch.float32, device=wgt.device)
        N.check(lib.tal_transpose_fwd(N.ptr(wgt), D, E0, N.ptr(t), N.stream_handle()), "tal_transpose_fwd")
        cached = (key, t)
        model._tal_proj_t = cached
    return cached[1]


def lm_head(model, h, last_only=False):
    """h [B,U,D] -> logits [B,U,V] (or [B,V] for the last position only)."""
    lib = N.lib()
    B, U, D = h.shape
    emb = model.embedding.weight
    V, E0 = emb.shape
    if last_only:
        src, M, ldh = h[:, U - 1], B, U * D
        logits = torch.empty(B, V, dtype=torch.float32, device=h.device)
    else:
        src, M, ldh = h, B * U, D
        logits = torch.empty(B, U, V, dtype=torch.float32, device=h.device)
    pt = _proj_t(model) if model.embed_size else None
    nws = M * E0 * 4
    ws = ops._ws(nws, h.device)
    N.check(lib.tal_lm_head_fwd(C.c_void_p(src.data_ptr()), M, ldh, D, N.ptr(pt), E0, N.ptr(emb), V, N.ptr(logits),
                                N.ptr(ws), nws, N.stream_handle()), "tal_lm_head_fwd")
    return logits


@torch.no_grad()
def asr_decode(model, y_prev, encoder_out, causal=True, last_only=False, check_tokens=True):
    """ASRModel.decode (models.py:203-247).  last_only=True computes the LM head for the
    final position only (all the reference's decode loops read, system.py:124,355-361)."""
    h = _run_stack(model, model.decoder, y_prev, encoder_out["encoder_out"], encoder_out["encoder_padding_mask"],
                   causal, check_tokens)
    return lm_head(model, h, last_only)


@torch.no_grad()
def asr_decode_spk(model, y_prev, encoder_out, causal=True, last_only=False):
    """ASRModel.decode_spk (models.py:249-289)."""
    h = _run_stack(model, model.spk_decoder, y_prev, encoder_out["speaker_out"],
                   encoder_out["encoder_padding_mask"], causal)
    if last_only:
        h = h[:, -1].contiguous()
    a, b = model.speaker_head[0], model.speaker_head[1]
    return ops.linear(ops.linear(h, a.weight, a.bias), b.weight, b.bias)


def log_softmax(x):
    """Row-wise log_softmax over the last dim (system.py:125,366)."""
    lib = N.lib()
    x = ops._f32c(x, "log_softmax")
    out = torch.empty_like(x)
    Nn = x.shape[-1]
    N.check(lib.tal_log_softmax_rows(N.ptr(x), x.numel() // Nn, Nn, N.ptr(out), N.stream_handle()),
            "tal_log_softmax_rows")
    return out
