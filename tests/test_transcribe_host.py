"""CPU tests of the windowed-transcription host logic against vectors recorded from the reference's own
splice helpers (tal/asr/transcribe.py:29-76; tests/golden/make_golden.py, section `transcribe`)."""
import json
import os

import pytest

from tal_asrd_amd import transcribe as T

HERE = os.path.dirname(os.path.abspath(__file__))


def test_splice_helpers_match_reference():
    with open(os.path.join(HERE, "golden", "splice_strings.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 6
    for c in cases:
        strs, wo = c["strs"], c["word_overlap"]
        for i in range(len(strs) - 1):
            assert list(T.overlap_ix(strs[i], strs[i + 1], wo)) == c["overlap_ix"][i]
            assert list(T.splice_ix(strs[i], strs[i + 1], wo)) == c["splice_ix"][i]
        assert T.splice_strings(strs, wo) == c["spliced"]
    with pytest.raises(IndexError):
        T.splice_strings(["only one"], 5)


def test_window_bounds():
    # tal/asr/transcribe.py:124: n = ceil((len - window) / stride) + 1
    assert T.window_bounds(400000, 160000, 120000) == [(0, 160000), (120000, 280000), (240000, 400000)]
    assert T.window_bounds(400001, 160000, 120000)[-1] == (360000, 520000)       # a short last window
    assert T.window_bounds(160000, 160000, 120000) == [(0, 160000)]
    assert T.window_bounds(100000, 160000, 120000) == [(0, 160000)]              # shorter than one window
    with open(os.path.join(HERE, "golden", "flow_transcribe.json")) as f:
        g = json.load(f)
    assert len(T.window_bounds(g["audio_len"], g["window"], g["stride"])) == g["n_windows"]
