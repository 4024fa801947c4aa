"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol
include/tal_asrd.h declares, the Python mirror keeps the reference's state_dict keys,
host logic (padding mask, frame counts) matches the oracle, and the product path
refuses to run without a GPU instead of silently falling back."""
import json
import os
import re

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN, ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "tal_asrd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tal_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding_table():
    from tal_asrd_amd import _native
    assert _header_symbols() == sorted(_native.SIGNATURES)


def test_library_exports_every_symbol():
    import __graft_entry__ as g
    g.build()
    from tal_asrd_amd import _native
    lib = _native.lib()
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert lib.tal_version() >= 100
    assert lib.tal_logmel_num_frames(480000) == 3001
    assert lib.tal_logmel_num_frames(15999) == 100
    assert lib.tal_logmel_plan_bytes() > 0


def test_struct_layout_matches_header():
    """ctypes mirror of tal_tds_desc must have the C layout (8-byte pointers, natural alignment)."""
    import ctypes as C
    from tal_asrd_amd import _native
    assert C.sizeof(_native.TdsBlockW) == 6 * 8 + 8 + 3 * 8
    expect = 4 + 4 + 5 * 4 + 4 * 4          # ints
    expect = (expect + 7) // 8 * 8           # align for pointers
    expect += 4 * 8 * 2 + 4 * 8 * C.sizeof(_native.TdsBlockW) + 4 * 8
    expect += 8                              # flags + pad
    assert C.sizeof(_native.TdsDesc) == expect
    assert _native.TdsDesc.flags.offset == expect - 8
    assert C.sizeof(_native.GreedyCtx) == 8 + 8 * 4 + 8 * 8 + 8 + 8 + 3 * 8 + 8 + 8 + 5 * 8 + 8 + 8    # pointer, 8 ints, 8 pointers, workspace + size, 3 pointers, device alias, seq + pad, pitch + episode table, pick_bias, no_fold + pad


def test_struct_layouts_against_the_c_compiler(tmp_path):
    """Every struct of include/tal_asrd.h as gcc lays it out (sizeof + the offset of every field) against its ctypes mirror."""
    import ctypes as C
    import subprocess
    from tal_asrd_amd import _native
    pairs = {"tal_tds_block_w": _native.TdsBlockW, "tal_tds_desc": _native.TdsDesc, "tal_decoder_layer_w": _native.DecoderLayerW,
             "tal_greedy_ctx": _native.GreedyCtx, "tal_unaligned_state": _native.UnalignedState}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "tal_asrd.h"', 'int main(void) {']
    for cname, mirror in pairs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in mirror._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        cname, fname, val = ln.split()
        got[(cname, fname)] = int(val)
    for cname, mirror in pairs.items():
        assert got[(cname, "sizeof")] == C.sizeof(mirror), cname
        for fname, _ in mirror._fields_:
            assert got[(cname, fname)] == getattr(mirror, fname).offset, (cname, fname)


def test_error_path_no_gpu_needed():
    """Argument validation happens before any launch, so it is testable on CPU."""
    import ctypes as C
    from tal_asrd_amd import _native
    lib = _native.lib()
    rc = lib.tal_linear_fwd(None, None, None, None, 0.0, 0, 4, 4, 4, None, None)
    assert rc == -1
    assert b"null pointer" in lib.tal_last_error()
    d = _native.TdsDesc()
    d.n_stages = 9
    assert lib.tal_tds_workspace_bytes(C.byref(d), 1, 100) >= 0
    rc = lib.tal_tds_fwd(C.byref(d), None, 1, 100, None, None, 0, None)
    assert rc == -1 and b"n_stages" in lib.tal_last_error()


@pytest.mark.parametrize("name,ctor", [
    ("SDModel", lambda m: m.SDModel()),
    ("ASRModel_2x_spk", lambda m: m.ASRModel("2x", num_speakers=6008, use_speaker_head=True)),
    ("ASRModel_1x_tok", lambda m: m.ASRModel("1x", num_speakers=40, use_speaker_head=False)),
    ("ASRModel_1x_e0", lambda m: m.ASRModel("1x", num_speakers=6008, vocab_size=10000, use_speaker_head=True, embed_size=0)),
])
def test_state_dict_keys_match_reference(name, ctor):
    from tal_asrd_amd import models
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))[name]
    own = [[k, list(v.shape)] for k, v in ctor(models).state_dict().items()]
    assert own == ref


def test_buffers_match_oracle():
    from oracle import tal_oracle as O
    from tal_asrd_amd import models, modules
    m = models.LogMelSpec()
    np.testing.assert_array_equal(m.mel_transform.mel_scale.fb.numpy(), O.mel_filterbank().numpy())
    np.testing.assert_allclose(m.mel_transform.spectrogram.window.numpy(), O.hann_window().numpy(), atol=1e-7)
    np.testing.assert_array_equal(modules.sinusoid_table(32, 64).numpy(), O.positional_encoding(32, 64).numpy())


def test_padding_mask_host_logic():
    from oracle import tal_oracle as O
    from tal_asrd_amd import padding_mask
    for lens, t in (([480000, 400000], 358), ([4800000, 123457, 4800000], 3733), ([16000], 8)):
        got = padding_mask(torch.tensor(lens), t, "cpu").numpy()
        np.testing.assert_array_equal(got, O.padding_mask(lens, t))


def test_no_cpu_fallback():
    from tal_asrd_amd import NativeError, SDModel
    m = SDModel()
    with pytest.raises(NativeError):
        m.encode(torch.zeros(1, 16000))
    with pytest.raises(NativeError):
        m.spk_embed_proj(torch.zeros(4, 1440))


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "tal_asrd_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_synth_is_deterministic():
    from tal_asrd_amd import synth
    a = synth.synth_audio(16000, 7)
    b = synth.synth_audio(16000, 7)
    np.testing.assert_array_equal(a, b)
    assert abs(a).max() <= 0.5 and (a == 0).sum() > 1000
    w = synth.fill_state_dict({"x.weight": (4, 8), "x.bias": (4,), "b.resweight": (1,)})
    assert w["x.weight"].shape == (4, 8) and 0.2 <= float(w["b.resweight"][0]) <= 0.32
    assert abs(w["x.weight"]).max() <= 1 / np.sqrt(8)


def test_options_are_explicit_calls_not_environment():
    """Kernel-selection switches are tal_set_option calls (an environment variable must not change which
    kernels a caller of the C ABI gets): names enumerate, unknown names are errors, and no source of the shipped library
    reads the environment."""
    import ctypes as C
    import glob
    from tal_asrd_amd import _native as N
    lib = N.lib()
    names = []
    i = 0
    while True:
        nm = lib.tal_option_name(i)
        if not nm:
            break
        names.append(nm.decode())
        i += 1
    assert "tds_exact_f32" in names and "decode_small_rows" in names and len(names) == len(set(names))
    assert N.get_option("tds_exact_f32") == 0 and N.get_option("decode_small_rows") == 256 and N.get_option("gconv_short_below") == 4 and N.get_option("gemm_s64_below") == 2
    N.set_option("tds_exact_f32", 1)
    assert N.get_option("tds_exact_f32") == 1
    N.set_option("tds_exact_f32", 0)
    assert lib.tal_set_option(b"no_such_option", 1) != 0
    assert b"unknown option" in lib.tal_last_error()
    assert lib.tal_version() >= 300
    for src in glob.glob(os.path.join(ROOT, "tal_asrd_amd", "csrc", "*")):
        assert "getenv" not in open(src).read(), src


def test_tiled_entry_points_agree_with_the_python_plan():
    """tal_tds_halo (host arithmetic of tal_tds_tiled_fwd) against tiling.receptive_halo: [640, 780] frames at stride 8 for
    the reference's 2 / 3 / 6 block stack (output frame c reads mel frames [8c - 640, 8c + 780])."""
    import ctypes as C
    from tal_asrd_amd import _native as N, tiling
    lib = N.lib()
    for depths in ((2, 3, 6), (1, 1, 2), (0, 4, 1), (3,)):
        d = N.TdsDesc()
        d.n_stages, d.groups = len(depths), 8
        for i in range(len(depths) + 1):
            d.channels[i] = 8 * (i + 1)
        for i, v in enumerate(depths):
            d.depths[i] = v
        left, right, stride = C.c_int64(), C.c_int64(), C.c_int64()
        assert lib.tal_tds_halo(C.byref(d), C.byref(left), C.byref(right), C.byref(stride)) == 0
        assert (left.value, right.value, stride.value) == tiling.receptive_halo(depths), depths
        if depths == (2, 3, 6):
            assert (left.value, right.value, stride.value) == (640, 780, 8)
        # the tiled workspace holds the longest slice's own workspace, its output and the status block
        T, tile = 30001, 512
        plan = tiling.plan_tiles(T, tile, depths)
        longest = max(t.in_stop - t.in_start for t in plan)
        need = lib.tal_tds_tiled_workspace_bytes(C.byref(d), T, tile)
        assert need >= lib.tal_tds_workspace_bytes(C.byref(d), 1, longest) + 64
        assert lib.tal_tds_tiled_status_offset(C.byref(d), T, tile) == need - 64


def test_malformed_tal_options_never_half_applies():
    """TAL_OPTIONS is parsed and validated as a whole before the library is published: a bad entry raises on every lib() call
    of that process and none of the good entries has been applied."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from tal_asrd_amd import _native as N\n"
            "for _ in range(2):\n"
            "    try:\n"
            "        N.lib(); print('loaded')\n"
            "    except N.NativeError as e:\n"
            "        print('raised', e)\n"
            "import ctypes as C\n"
            "l = C.CDLL(N.LIB_PATH); v = C.c_int(); l.tal_get_option(b'gemm_no_w64', C.byref(v)); print('gemm_no_w64', v.value)\n" % ROOT)
    for bad in ("gemm_no_w64=1,decode_small_rows=abc", "gemm_no_w64,no_such_switch=1"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TAL_OPTIONS=bad), capture_output=True, text=True, timeout=120).stdout
        assert out.count("raised") == 2 and "loaded" not in out and "gemm_no_w64 0" in out, out
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TAL_OPTIONS="gemm_no_w64, decode_small_rows=128"), capture_output=True, text=True, timeout=120).stdout
    assert out.count("loaded") == 2 and "gemm_no_w64 1" in out, out
