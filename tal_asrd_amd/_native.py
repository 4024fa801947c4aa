"""ctypes binding of the C-ABI library (include/tal_asrd.h -> libtal_asrd_hip.so).

The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950).  There
is NO fallback: if the library is missing or an entry point fails, the product
path raises.  This module never imports anything from oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAL_ASRD_LIB", os.path.join(_HERE, "libtal_asrd_hip.so"))  # override: kernel ablation builds only

TAL_MAX_STAGES = 4
TAL_MAX_DEPTH = 8
TAL_TDS_EXACT_F32 = 1
TAL_TDS_OUT_SPLIT = 2
TAL_GROUP_MAX = 16      # sessions per merged decode step (csrc/common.h)

c_float_p = C.c_void_p  # device pointers travel as integers


class TdsBlockW(C.Structure):
    _fields_ = [("conv_w", C.c_void_p), ("conv_b", C.c_void_p), ("fc0_w", C.c_void_p), ("fc0_b", C.c_void_p),
                ("fc3_w", C.c_void_p), ("fc3_b", C.c_void_p), ("resweight", C.c_float), ("_pad", C.c_int32),
                ("fc0_w_split", C.c_void_p), ("fc3_w_split", C.c_void_p), ("conv_w_frag", C.c_void_p)]


class TdsDesc(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("groups", C.c_int32),
                ("channels", C.c_int32 * (TAL_MAX_STAGES + 1)), ("depths", C.c_int32 * TAL_MAX_STAGES),
                ("down_w", C.c_void_p * TAL_MAX_STAGES), ("down_b", C.c_void_p * TAL_MAX_STAGES),
                ("blocks", (TdsBlockW * TAL_MAX_DEPTH) * TAL_MAX_STAGES),
                ("down_w_frag", C.c_void_p * TAL_MAX_STAGES), ("flags", C.c_int32), ("_pad2", C.c_int32)]


class DecoderLayerW(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ca_in_w", "ca_in_b",
                                          "ca_out_w", "ca_out_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b")] + \
               [("resweight", C.c_float), ("resweight_src", C.c_float)] + \
               [(n, C.c_void_p) for n in ("fold_sa_w", "fold_sa_b", "fold_ca_w", "fold_ca_b")]


class GreedyCtx(C.Structure):
    _fields_ = [("layers", C.c_void_p), ("n_layers", C.c_int32), ("E", C.c_int32), ("H", C.c_int32), ("FF", C.c_int32),
                ("V", C.c_int32), ("E0", C.c_int32), ("S", C.c_int32), ("max_len", C.c_int32),
                ("emb", C.c_void_p), ("proj", C.c_void_p), ("proj_t", C.c_void_p), ("pe", C.c_void_p),
                ("k_cache", C.c_void_p), ("vt_cache", C.c_void_p), ("mem_kpm", C.c_void_p), ("tokens", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("picked_dev", C.c_void_p),
                ("picked_host", C.c_void_p), ("tickets", C.c_void_p), ("picked_host_dev", C.c_void_p),
                ("seq", C.c_uint32), ("needs_reset", C.c_uint32), ("k_pitch", C.c_int64), ("kv_all", C.c_void_p), ("kpm_all", C.c_void_p),
                ("enc_frames", C.c_int64), ("kv_pitch", C.c_int64), ("pick_bias", C.c_void_p),
                ("no_fold", C.c_uint32), ("_pad2", C.c_uint32)]


class UnalignedState(C.Structure):
    _fields_ = [("gen", C.c_void_p), ("gen_cap", C.c_int64), ("n", C.c_int64), ("history_start", C.c_int64), ("chunk_start", C.c_int64),
                ("encoder_len", C.c_int64), ("eos", C.c_int64), ("it", C.c_int64), ("max_iters", C.c_int64),
                ("rec_chunk_start", C.c_void_p), ("rec_attn", C.c_void_p), ("rec_len", C.c_void_p), ("rec_cap", C.c_int64),
                ("n_rec", C.c_int64), ("rec_stride", C.c_int32), ("chunk_size", C.c_int32), ("max_positions", C.c_int32),
                ("stall_patience", C.c_int32), ("rep_n", C.c_int32), ("skip_frames", C.c_int32), ("shift_frames", C.c_int32),
                ("del_prct", C.c_float), ("thresh_prct", C.c_double), ("highest_progress", C.c_double),
                ("num_no_improve", C.c_int32), ("window_time", C.c_int32), ("flags", C.c_int32), ("gen_pinned", C.c_int32)]


UNALIGNED_WINDOW_MOVED, UNALIGNED_PREFIX_REWRITTEN, UNALIGNED_DONE, UNALIGNED_GROW, UNALIGNED_ALONE = 1, 2, 4, 8, 16

# name -> (restype, argtypes); must list every symbol include/tal_asrd.h declares
# (tests/test_abi.py checks header <-> table <-> library).
_i, _i64, _sz, _f, _p = C.c_int, C.c_int64, C.c_size_t, C.c_float, C.c_void_p
SIGNATURES = {
    "tal_version": (_i, []),
    "tal_last_error": (C.c_char_p, []),
    "tal_set_option": (_i, [C.c_char_p, _i]),
    "tal_get_option": (_i, [C.c_char_p, C.POINTER(C.c_int)]),
    "tal_option_name": (C.c_char_p, [_i]),
    "tal_logmel_num_frames": (_i64, [_i64]),
    "tal_logmel_plan_bytes": (_sz, []),
    "tal_logmel_plan_init": (_i, [_p, _p, _p, _p]),
    "tal_logmel_workspace_bytes": (_sz, [_i, _i64]),
    "tal_logmel_fwd": (_i, [_p, _p, _i, _i64, _f, _i, _p, _p, _p, _p, _sz, _p]),
    "tal_logmel_f16_fwd": (_i, [_p, _p, _i, _i64, _f, _i, _p, _p, _p, _p, _sz, _p]),
    "tal_subtract_scalar": (_i, [_p, _i64, _p, _p]),
    "tal_linear_fwd": (_i, [_p, _p, _p, _p, _f, _i, _i64, _i, _i, _p, _p]),
    "tal_linear_workspace_bytes": (_sz, [_i64, _i, _i]),
    "tal_split_f16x3_fwd": (_i, [_p, _p, _i64, _i, _p]),
    "tal_linear_f16x3_fwd": (_i, [_p, _p, _p, _p, _f, _i, _i64, _i, _i, _p, _i, _p, _sz, _p]),
    "tal_linear_ws_fwd": (_i, [_p, _p, _p, _p, _f, _i, _i64, _i, _i, _p, _p, _sz, _p]),
    "tal_linear_f16x3_guarded_fwd": (_i, [_p, _p, _p, _p, _i, _f, _i, _i64, _i, _i, _p, _i, _p, _p, _sz, _p]),
    "tal_pack_gconv_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "tal_gconv_s2_fwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _p, _p]),
    "tal_gconv_res_fwd": (_i, [_p, _p, _p, _f, _i, _i64, _i, _i, _p, _p]),
    "tal_gconv_f16x3_weight_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "tal_pack_gconv_f16x3_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "tal_gconv_s2_f16x3_fwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _p, _p]),
    "tal_gconv_res_f16x3_fwd": (_i, [_p, _p, _p, _f, _i, _i64, _i, _i, _p, _p, _p]),
    "tal_gconv_res_split_fwd": (_i, [_p, _p, _p, _f, _i, _i64, _i, _i, _p, _p]),
    "tal_gconv_s2_split_fwd": (_i, [_p, _i, _p, _p, _i, _i64, _i, _i, _i, _p, _p]),
    "tal_tds_out_len": (_i64, [C.POINTER(TdsDesc), _i64]),
    "tal_tds_workspace_bytes": (_sz, [C.POINTER(TdsDesc), _i, _i64]),
    "tal_tds_status_offset": (_sz, [C.POINTER(TdsDesc), _i, _i64]),
    "tal_tds_fwd": (_i, [C.POINTER(TdsDesc), _p, _i, _i64, _p, _p, _sz, _p]),
    "tal_tds_out_split": (_i, [C.POINTER(TdsDesc), _i, _i64]),
    "tal_tds_premean_ok": (_i, [C.POINTER(TdsDesc), _p]),
    "tal_tds_premean_fwd": (_i, [C.POINTER(TdsDesc), _p, _p, _i, _i64, _p, _p, _sz, _p]),
    "tal_tds_halo": (_i, [C.POINTER(TdsDesc), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tal_tds_tiled_workspace_bytes": (_sz, [C.POINTER(TdsDesc), _i64, _i64]),
    "tal_tds_tiled_status_offset": (_sz, [C.POINTER(TdsDesc), _i64, _i64]),
    "tal_tds_tiled_fwd": (_i, [C.POINTER(TdsDesc), _p, _i64, _p, _i64, _p, _sz, _p]),
    "tal_sd_head_workspace_bytes": (_sz, [_i64, _i]),
    "tal_sd_head_fwd": (_i, [_p, _i64, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _sz, _p]),
    "tal_sd_head_split_fwd": (_i, [_p, _i64, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _sz, _p]),
    "tal_argmax_rows": (_i, [_p, _i64, _i, _p, _p]),
    "tal_embed_tokens_fwd": (_i, [_p, _i, _i, _p, _i, _i, _p, _i, _p, _i, _p, _p]),
    "tal_add_positional_fwd": (_i, [_p, _i, _i, _i, _p, _i, _p, _p]),
    "tal_pad4": (_i64, [_i64]),
    "tal_cross_kv_fwd": (_i, [C.POINTER(DecoderLayerW), _p, _i, _i, _i, _p, _p, _p]),
    "tal_decoder_layer_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "tal_decoder_layer_fwd": (_i, [C.POINTER(DecoderLayerW), _p, _i, _i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p,
                                   _p, _sz, _p]),
    "tal_decoder_stack_fwd": (_i, [_p, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "tal_lm_head_fwd": (_i, [_p, _i64, _i64, _i, _p, _i, _p, _i, _p, _p, _sz, _p]),
    "tal_transpose_fwd": (_i, [_p, _i, _i, _p, _p]),
    "tal_greedy_step_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "tal_greedy_step_fwd": (_i, [C.POINTER(GreedyCtx), _i64, _i64, _i, _p]),
    "tal_greedy_step_poll": (_i, [C.POINTER(GreedyCtx), _i]),
    "tal_greedy_group_ok": (_i, [C.POINTER(GreedyCtx), _i64, _i64]),
    "tal_window_vt_fwd": (_i, [_p, _i, _i64, _i, _i, _i64, _p, _p]),
    "tal_greedy_set_window": (_i, [C.POINTER(GreedyCtx), _i64, _p]),
    "tal_greedy_step_multi_fwd": (_i, [C.POINTER(C.POINTER(GreedyCtx)), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i, _p]),
    "tal_unaligned_consume": (_i, [C.POINTER(UnalignedState), _i64, _p, _i]),
    "tal_unaligned_group_run": (_i, [C.POINTER(C.POINTER(UnalignedState)), C.POINTER(C.POINTER(GreedyCtx)), C.POINTER(C.c_int64), _i, _i, _p]),
    "tal_ngram_repeat_count": (_i64, [_p, _i64, _i]),
    "tal_greedy_pick_fwd": (_i, [_p, _i, _p, _i, _i64, _i, _p, _p, _p]),
    "tal_log_softmax_rows": (_i, [_p, _i64, _i, _p, _p]),
    "tal_beam_topk": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "tal_attn_pool_fwd": (_i, [_p, _p, _p, _i64, _i, _i, _i, _i, _p, _p]),
    "tal_attn_vote_groups_fwd": (_i, [_p, _p, _p, _i64, _i, _p, _i, _i, _i, _p, _p, _p]),
    "tal_majority_vote_fwd": (_i, [_p, _i64, _p, _i, _i, _p, _p, _p]),
    "tal_attn_vote_fwd": (_i, [_p, _p, _p, _i64, _i, _i, _p, _p, _p]),
    "tal_gru_cell_workspace_bytes": (_sz, [_i, _i]),
    "tal_gru_cell_fwd": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "tal_prof_enable": (_i, [_i]),
    "tal_prof_reset": (_i, []),
    "tal_prof_collect": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "tal_asrd_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for the hot path." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        # measurement scripts select kernels with TAL_OPTIONS="name=value,name,..." (a bare name means 1); the library
        # itself never reads the environment -- a C caller uses tal_set_option.  The whole string is parsed and checked against
        # the library's option list BEFORE anything is applied or the library is published: a malformed value raises here and the
        # next lib() call raises again, instead of running on half of the switches.
        pending = []
        for item in filter(None, (x.strip() for x in os.environ.get("TAL_OPTIONS", "").split(","))):
            name, _, val = item.partition("=")
            name = name.strip()
            try:
                value = int(val) if val.strip() else 1
            except ValueError:
                raise NativeError("TAL_OPTIONS: %r is not name=integer" % item) from None
            probe = C.c_int()
            if l.tal_get_option(name.encode(), C.byref(probe)) != 0:
                raise NativeError("TAL_OPTIONS: unknown option %r" % name)
            pending.append((name, value))
        for name, value in pending:
            if l.tal_set_option(name.encode(), value) != 0:
                raise NativeError("TAL_OPTIONS: tal_set_option(%s, %d) failed: %s" % (name, value, l.tal_last_error().decode()))
        _lib = l
    return _lib


def set_option(name, value=1):
    """tal_set_option: a process-wide kernel-selection switch (include/tal_asrd.h lists the names)."""
    check(lib().tal_set_option(name.encode(), int(value)), "tal_set_option(%s)" % name)


def get_option(name):
    v = C.c_int()
    check(lib().tal_get_option(name.encode(), C.byref(v)), "tal_get_option(%s)" % name)
    return v.value


def check(rc, what=""):
    if rc != 0:
        msg = lib().tal_last_error()
        raise NativeError("%s failed (rc=%d): %s" % (what or "tal call", rc, msg.decode() if msg else ""))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_handle():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, what):
    if not t.is_cuda:
        raise NativeError("%s: the hot path only runs on the GPU (HIP kernels); got a %s tensor and there is "
                          "deliberately no CPU fallback" % (what, t.device))
