"""CPU tests of the host-side decode helpers and of the multi-GPU sharding logic."""
import os

import numpy as np

from oracle import tal_oracle as O
from tal_asrd_amd.util import ngram_repeat_mask, split_speaker_turns


def test_ngram_repeat_mask_matches_oracle():
    rng = np.random.RandomState(0)
    for _ in range(20):
        xs = rng.randint(0, 4, size=(2, rng.randint(3, 40)))
        np.testing.assert_array_equal(ngram_repeat_mask(xs, 3), O.ngram_repeat_mask(xs, 3))
    xs = np.array([[1, 2, 3, 1, 2, 3, 1, 2, 3, 9]])
    np.testing.assert_array_equal(ngram_repeat_mask(xs, 3), [[0, 0, 0, 1, 1, 1, 1, 1, 1, 0]])
    assert ngram_repeat_mask(np.zeros((1, 4), dtype=np.int64), 5).sum() == 0


def test_split_speaker_turns():
    V = 100
    toks = [0, 5, 6, 1, 1, 103, 7, 8, 1, 9]
    turns, splits = split_speaker_turns(toks, V)
    assert turns == [([5, 6], None), ([7, 8], 3), ([9], None)]
    assert splits == [3, 8, 9]
    turns, splits = split_speaker_turns(toks, V, add_last=False)
    assert splits == [3, 8]
    assert split_speaker_turns([], V) == ([], [])


def test_native_ngram_repeat_count_matches_mask_sum():
    """The decode loop's C helper (tal_ngram_repeat_count, host code) against ngram_repeat_mask(...).sum()."""
    import numpy as np
    from tal_asrd_amd import _native as N
    from tal_asrd_amd.util import ngram_repeat_mask
    lib = N.lib()
    rng = np.random.RandomState(0)
    for _ in range(400):
        L, n = int(rng.randint(0, 60)), int(rng.randint(1, 7))
        row = rng.randint(0, 3, size=L).astype(np.int64)
        want = int(ngram_repeat_mask([row.tolist()], n).sum()) if L else 0
        assert lib.tal_ngram_repeat_count(row.ctypes.data, L, n) == want


def _unaligned_state(prime, encoder_len, chunk_size=357, max_iters=1000000, max_positions=512, thresh_prct=0.5, shift_prct=0.25,
                     stall_patience=25, rep_n=5, skip_prct=0.1, eos=1, cap=4096):
    import numpy as np
    from tal_asrd_amd import _native as N
    st = N.UnalignedState()
    bufs = {"gen": np.zeros(cap, np.int64), "cs": np.zeros(cap, np.int64), "attn": np.zeros((cap, chunk_size), np.float32), "len": np.zeros(cap, np.int32)}
    bufs["gen"][:len(prime)] = prime
    st.gen, st.gen_cap, st.n = bufs["gen"].ctypes.data, cap, len(prime)
    st.rec_chunk_start, st.rec_attn, st.rec_len = bufs["cs"].ctypes.data, bufs["attn"].ctypes.data, bufs["len"].ctypes.data
    st.rec_cap, st.rec_stride = cap, chunk_size
    st.encoder_len, st.eos, st.max_iters = encoder_len, eos, max_iters
    st.chunk_size, st.max_positions, st.stall_patience, st.rep_n = chunk_size, max_positions, stall_patience, rep_n
    st.skip_frames, st.shift_frames = int(chunk_size * skip_prct), int(chunk_size * shift_prct)
    st.del_prct, st.thresh_prct = float(np.float32(shift_prct / thresh_prct)), thresh_prct
    return st, bufs


def test_unaligned_consume_follows_the_pinned_control_flow():
    """tal_unaligned_consume -- the product's ONE implementation of System.generate_unaligned's per-token decisions
    (tal/asr/system.py:389-521), a C host helper -- driven with the raw (token, attention row) of every step of the oracle's
    run on the two recorded episodes (the oracle reproduces the trajectories recorded from the reference,
    tests/test_oracle_golden.py): after EVERY step the window start, history start and stream length agree, the flags say
    exactly when the window moved / the prefix was rewritten / the episode ended, and the final token stream, recorded window
    starts and attention rows equal the fixture's."""
    import ctypes as C
    import json
    import os
    import numpy as np
    from oracle import tal_oracle as O
    from tal_asrd_amd import synth, _native as N
    from tests.conftest import GOLDEN, golden
    lib = N.lib()
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["ASRModel_2x_spk"]
    sd = synth.fill_state_dict({k: tuple(s) for k, s in keys})
    for name in ("flow_unaligned_short", "flow_unaligned"):
        g = golden(name)
        L = int(g["audio_len"])
        enc_len = O.tds_total_out_len(O.num_frames(L))
        st, bufs = _unaligned_state([1], enc_len, max_iters=int(g["max_iters"]))
        seen = {"steps": 0, "prev_win": 0}

        def on_step(token, row, win, hist, ntok, finished):
            row = np.ascontiguousarray(row, dtype=np.float32)
            st.flags = 0
            flags = lib.tal_unaligned_consume(C.byref(st), token, row.ctypes.data, row.shape[0])
            assert flags >= 0, lib.tal_last_error()
            assert (st.chunk_start, st.history_start, st.n) == (win, hist, ntok), (seen["steps"], st.chunk_start, win, st.history_start, hist, st.n, ntok)
            assert bool(flags & N.UNALIGNED_WINDOW_MOVED) == (win != seen["prev_win"])
            assert bool(flags & N.UNALIGNED_DONE) == (bool(finished) or seen["steps"] + 1 >= int(g["max_iters"]))
            seen["prev_win"] = win
            seen["steps"] += 1
        toks, starts, rows = O.generate_unaligned(synth.synth_audio_batch(1, L, int(g["audio_seed"])), [[1]], [L], sd,
                                                  max_iters=int(g["max_iters"]), stall_patience=25, on_step=on_step)
        assert seen["steps"] > 30
        np.testing.assert_array_equal(bufs["gen"][:st.n], g["generated"][0])
        np.testing.assert_array_equal(bufs["cs"][:st.n_rec], g["chunk_start"])
        lens = bufs["len"][:st.n_rec]
        if "attn" in g.files:
            np.testing.assert_allclose(bufs["attn"][:st.n_rec], g["attn"], atol=1e-5, rtol=0)
        else:
            assert lens.tolist() == g["attn_len"].tolist()
            np.testing.assert_allclose(np.concatenate([bufs["attn"][i, :lens[i]] for i in range(st.n_rec)]), g["attn_flat"], atol=1e-5, rtol=0)


def test_unaligned_consume_progress_next_to_the_threshold():
    """The attention centre of mass (tal/asr/system.py:405-408) decides `progress > thresh_prct` (window shift) and
    `progress > highest_progress` (stall counter).  The reference leaves the order of its float32 sum to torch (a vectorised
    reduction on the CPU, a tree on CUDA: its own two devices differ in the last place); tal_unaligned_consume takes the float32
    products, adds them in double and rounds ONCE to float32.  Pinned here on rows whose exact sum lies within a few ulps of the
    threshold: the decision is the one exact arithmetic makes -- no row within an ulp flips by summation order."""
    import ctypes as C
    import numpy as np
    from tal_asrd_amd import _native as N
    lib = N.lib()
    S, thresh = 357, 0.5
    rng = np.random.RandomState(7)
    ramp = (np.arange(S, dtype=np.float32) / np.float32(S)).astype(np.float32)
    checked = {True: 0, False: 0}
    for trial in range(200):
        # two spikes either side of the window centre + a little mass everywhere; the weight of the upper spike is bisected until the
        # exact centre of mass sits a few float32 steps from the threshold, on the side `want_above`
        a, b = int(rng.randint(120, 170)), int(rng.randint(190, 240))
        base = (rng.rand(S) * 1e-4).astype(np.float32)
        want_above = bool(trial & 1)
        lo, hi = 0.0, 1.0
        for _ in range(int(rng.randint(19, 25))):          # stop a few float32 steps (6e-8 at 0.5) away, not on the threshold
            w = 0.5 * (lo + hi)
            row = base.copy()
            row[a] += np.float32(1.0 - w)
            row[b] += np.float32(w)
            exact = float(np.sum((row * ramp).astype(np.float64)))        # the float32 products, summed exactly enough (53 bits)
            if exact > thresh:
                hi = w
            else:
                lo = w
        w = hi if want_above else lo
        row = base.copy()
        row[a] += np.float32(1.0 - w)
        row[b] += np.float32(w)
        exact = float(np.sum((row * ramp).astype(np.float64)))
        assert abs(exact - thresh) < 2e-6                              # a handful of float32 steps at 0.5
        decision = float(np.float32(exact)) > thresh                    # rounded once
        st, bufs = _unaligned_state([1, 5, 6], encoder_len=4000)
        st.window_time, st.n_rec = 3, 2
        row = np.ascontiguousarray(row, dtype=np.float32)
        flags = lib.tal_unaligned_consume(C.byref(st), 7, row.ctypes.data, S)
        assert flags >= 0, lib.tal_last_error()
        moved = bool(flags & N.UNALIGNED_WINDOW_MOVED)
        assert moved == decision, (trial, exact, moved)
        assert (st.chunk_start == st.shift_frames) == decision
        checked[decision] += 1
    assert checked[True] > 20 and checked[False] > 20


def test_parameter_registration_counter_is_scoped_to_this_packages_modules():
    """A drop-in library must not watch the host program's modules: the counter that invalidates cached
    parameter lists is per module tree of this package; a Parameter registered by any other module leaves it alone."""
    import torch
    from torch import nn
    from tal_asrd_amd import models as M
    tds = M.TDS(input_size=8, sizes=[8, 16], depths=[1])
    layer = M.ModRZTXDecoderLayer(64, 4, 128)
    e_tds, e_layer = M.param_epoch(tds), M.param_epoch(layer)
    nn.Linear(4, 4)                                         # the host program builds a module
    host = nn.Module()
    host.w = nn.Parameter(torch.zeros(3))
    assert (M.param_epoch(tds), M.param_epoch(layer)) == (e_tds, e_layer)
    conv = tds.blocks[0][0]
    conv.weight = nn.Parameter(conv.weight.detach().clone())                # replaced inside the tree: noticed by that tree only
    assert M.param_epoch(tds) == e_tds + 1 and M.param_epoch(layer) == e_layer
    layer.linear1.bias = nn.Parameter(layer.linear1.bias.detach().clone())
    assert M.param_epoch(layer) == e_layer + 1 and M.param_epoch(tds) == e_tds + 1
    k0 = tds._param_key()
    conv.weight = nn.Parameter(conv.weight.detach().clone())
    assert tds._param_key() != k0                           # the cached list was rebuilt (new storage)


def test_param_epoch_notices_late_submodules_and_survives_pickle(tmp_path):
    """The counter behind the cached parameter lists (models.param_epoch): a submodule attached after construction joins the tree
    and moves the counter; a Parameter replaced inside it moves it again; a tree restored by pickle in a FRESH process -- no
    constructor of this package has run there -- still notices a replaced Parameter (the hooks are installed wherever the
    counter is read)."""
    import pickle
    import subprocess
    import sys
    import textwrap
    import torch
    from tal_asrd_amd import models as M
    m = M.TDSBlock(32, 21, 8)
    e0 = M.param_epoch(m)
    m.extra = torch.nn.Linear(3, 3)
    assert M.param_epoch(m) == e0 + 1
    m.extra.weight = torch.nn.Parameter(torch.zeros(3, 3))
    assert M.param_epoch(m) == e0 + 2
    p = tmp_path / "m.pkl"
    with open(p, "wb") as f:
        pickle.dump(m, f)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import sys, pickle
        sys.path.insert(0, %r)
        import torch
        m = pickle.load(open(%r, "rb"))
        from tal_asrd_amd import models as M
        e = M.param_epoch(m)
        conv = next(x for x in m.modules() if isinstance(x, torch.nn.Conv1d))
        conv.weight = torch.nn.Parameter(conv.weight.detach().clone())
        assert M.param_epoch(m) == e + 1, (e, M.param_epoch(m))
        print("ok")
    ''') % (root, str(p))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_bench_clock_sampler_assigns_samples_to_passes():
    """bench.ClockSampler.window: samples stamped by the sampler child are averaged over the wall-clock window of a pass; an unavailable
    sampler says so instead of reporting zeros."""
    import bench
    s = bench.ClockSampler.__new__(bench.ClockSampler)
    s.proc, s.why = None, None
    s.samples = [(10.0, 1900.0, 1100.0), (10.5, 1800.0, 1300.0), (11.0, None, 1350.0), (12.0, 1700.0, 1400.0)]
    w = s.window(10.4, 11.5)
    assert w["samples"] == 1 and w["sclk_mhz"] == 1800.0 and w["sclk_mhz_min"] == 1800.0 and abs(w["power_w"] - 1325.0) < 1e-9
    assert s.window(20.0, 21.0) == {"sclk_mhz": None, "sclk_mhz_min": None, "power_w": None, "samples": 0}
    s.why = "ERR no device"
    assert s.window(10.0, 12.0)["unavailable"] == "ERR no device"


def test_the_committed_traffic_figure_describes_this_trees_dense_kernels():
    """`roofline.traffic` of the bench line is quoted from profiles/rN_pmc_traffic.json only while that file's recorded hash of the
    dense-kernel sources equals this tree's (bench.load_traffic): an edit of those sources after the round's profile run would leave
    the driver's line with `traffic: null` -- this test fails first, here."""
    import bench
    traffic, source = bench.load_traffic()
    assert traffic is not None, source
    assert 0.8e9 < traffic < 3e9
