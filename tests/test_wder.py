"""CPU tests of the restated WER / WDER scorer (tal/wder.py:41-73,150-234)."""
import numpy as np

from tal_asrd_amd import wder as W
from tests.conftest import golden


def test_levenshtein_and_opcodes():
    assert W.levenshtein("kitten", "sitting") == 3
    assert W.levenshtein([], [1, 2]) == 2
    ops = W.align_opcodes("abcd", "abxd")
    assert [t for t, *_ in ops] == ["equal", "equal", "replace", "equal"]
    ops = W.align_opcodes("abc", "ac")
    assert sum(t == "delete" for t, *_ in ops) == 1 and sum(t == "equal" for t, *_ in ops) == 2
    for a, b in (("abcde", "xbdye"), ("", "ab"), ("abc", "")):
        ops = W.align_opcodes(a, b)
        assert sum(t != "equal" for t, *_ in ops) == W.levenshtein(a, b)


def test_wder_examples():
    ref = [("the", "A"), ("cat", "A"), ("sat", "B"), ("down", "B")]
    # perfect words, speaker labels permuted -> WER 0, WDER 0 (labels are matched optimally)
    hyp = [("the", 7), ("cat", 7), ("sat", 3), ("down", 3)]
    wer, dist, n, wder, rl, hl = W.calculate_wder(ref, hyp)
    assert (wer, dist, n, wder) == (0.0, 0, 4, 0.0)
    assert dict(zip(hl, rl)) == {7: "A", 3: "B"}
    # one substituted word carrying the wrong speaker: WER 1/4, WDER 1/4
    hyp = [("the", 7), ("dog", 3), ("sat", 3), ("down", 3)]
    wer, dist, n, wder, _, _ = W.calculate_wder(ref, hyp)
    assert (wer, dist, n) == (0.25, 1, 4) and abs(wder - 0.25) < 1e-12
    # an inserted word does not enter WDER
    hyp = [("the", 7), ("big", 3), ("cat", 7), ("sat", 3), ("down", 3)]
    wer, dist, n, wder, _, _ = W.calculate_wder(ref, hyp)
    assert dist == 1 and wder == 0.0
    assert W.calculate_wer(ref, hyp)[:2] == (0.25, 1)


def test_identical_token_streams_give_zero_error():
    """The repository's parity statement (identical tokens and speaker-change indices as the
    reference) implies identical WER / WDER: scoring the recorded reference stream against itself."""
    g = golden("flow_unaligned")
    toks = g["generated"][0].tolist()
    words = W.tokens_to_words(toks, 10000)
    assert len(words) > 50 and len({s for _, s in words}) >= 2
    wer, dist, n, wder, _, _ = W.calculate_wder(words, list(words))
    assert dist == 0 and wder == 0.0
    # a single flipped token moves WER by exactly 1/n and leaves WDER at 0
    hyp = list(words)
    hyp[10] = (hyp[10][0] + 1, hyp[10][1])
    wer2, dist2, _, wder2, _, _ = W.calculate_wder(words, hyp)
    assert dist2 == 1 and wder2 == 0.0


def test_corpus_wder_schema_and_none_quirk():
    ref = [("hello there", "jack"), ("general kenobi", "margaret"), ("you are bold", "jack")]
    hyp = [("hello there", (np.zeros(4), 5)), ("general kenobi", (np.ones(4), 9)), ("you are old", (np.zeros(4), 5))]
    words, n = W.convert_to_wder_format(hyp)
    assert n == 2 and words[0] == ("hello", 0) and words[2] == ("general", 1) and words[-1] == ("old", 0)
    # a None speaker stays its own label (reference builds its output from the unfilled list)
    words, n = W.convert_to_wder_format([("a b", None), ("c", 3), ("d", None)])
    assert n == 2 and [s for _, s in words] == [0, 0, 1, 0]
    owder, ower, wders = W.corpus_wder([(ref, hyp), (ref, []), (ref, hyp)])
    assert len(wders) == 2 and owder == 0.0 and abs(ower - 1 / 7) < 1e-12
