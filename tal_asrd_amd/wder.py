"""WER / WDER scoring (host side), restated from
tal/wder.py:41-73 (optimal speaker matching), :150-163 (WER), :166-234 (WDER,
https://arxiv.org/pdf/1907.05337.pdf):

    WER  = Levenshtein(ref_words, hyp_words) / len(ref_words)
    WDER = 1 - (best one-to-one speaker-label matching accuracy over the word pairs the
               ASR alignment marks as correct or substituted)

The reference gets the distance from `editdistance` and the alignment opcodes from
`edit_distance.SequenceMatcher` (neither is installed here, neither is vendored): the
distance is unique, but among equally cheap alignments the opcode choice of that library is
not documented, so WDER parity on texts with ambiguous alignments is UNPINNED; the backtrace
below prefers match/substitution, then deletion, then insertion, from the end of the strings.
When hypothesis and reference token streams are identical (the parity statement of this
repository: identical tokens and speaker-change indices), every alignment is the diagonal and
WER / WDER are identical by construction.
"""
import numpy as np
from scipy import optimize


def levenshtein_table(a, b):
    n, m = len(a), len(b)
    d = np.zeros((n + 1, m + 1), dtype=np.int64)
    d[:, 0] = np.arange(n + 1)
    d[0, :] = np.arange(m + 1)
    for i in range(1, n + 1):
        ai = a[i - 1]
        row, prev = d[i], d[i - 1]
        for j in range(1, m + 1):
            cost = 0 if ai == b[j - 1] else 1
            row[j] = min(prev[j] + 1, row[j - 1] + 1, prev[j - 1] + cost)
    return d


def levenshtein(a, b):
    return int(levenshtein_table(list(a), list(b))[-1, -1])


def align_opcodes(a, b):
    """[(tag, i0, i1, j0, j1)] with single-element spans; tags: equal / replace / delete / insert."""
    a, b = list(a), list(b)
    d = levenshtein_table(a, b)
    i, j = len(a), len(b)
    ops = []
    while i > 0 or j > 0:
        if i > 0 and j > 0 and d[i, j] == d[i - 1, j - 1] + (0 if a[i - 1] == b[j - 1] else 1):
            ops.append(("equal" if a[i - 1] == b[j - 1] else "replace", i - 1, i, j - 1, j))
            i, j = i - 1, j - 1
        elif i > 0 and d[i, j] == d[i - 1, j] + 1:
            ops.append(("delete", i - 1, i, j, j))
            i -= 1
        else:
            ops.append(("insert", i, i, j - 1, j))
            j -= 1
    return ops[::-1]


def sequence_match(seq1, seq2):
    """tal/wder.py:41-73: optimal one-to-one label matching (Hungarian) between two equally long
    label sequences -> (labels1 matched, labels2 matched, accuracy)."""
    seq1, seq2 = list(seq1), list(seq2)
    if not seq1 or len(seq1) != len(seq2):
        raise ValueError("sequence1 and sequence2 must have the same non-zero length")
    u1, u2 = sorted(set(seq1)), sorted(set(seq2))
    i1 = {v: k for k, v in enumerate(u1)}
    i2 = {v: k for k, v in enumerate(u2)}
    counts = np.zeros((len(u1), len(u2)))
    for x, y in zip(seq1, seq2):
        counts[i1[x], i2[y]] += 1.0
    rows, cols = optimize.linear_sum_assignment(-counts)
    return [u1[r] for r in rows], [u2[c] for c in cols], float(counts[rows, cols].sum() / len(seq1))


def calculate_wer(ref, hyp):
    """ref / hyp: lists of (word, speaker).  tal/wder.py:150-163 -> (wer, distance, n_ref)."""
    ref_words = [w for w, _ in ref]
    hyp_words = [w for w, _ in hyp]
    dist = levenshtein(ref_words, hyp_words)
    return dist / len(ref_words), dist, len(ref_words)


def calculate_wder(ref, hyp):
    """tal/wder.py:166-234 -> (wer, distance, n_ref, wder, ref_labels, hyp_labels)."""
    ref_words, ref_spk = [w for w, _ in ref], [s for _, s in ref]
    hyp_words, hyp_spk = [w for w, _ in hyp], [s for _, s in hyp]
    ops = align_opcodes(ref_words, hyp_words)
    dist = sum(1 for t, *_ in ops if t != "equal")
    pairs = [(ref_spk[i0], hyp_spk[j0]) for t, i0, _, j0, _ in ops if t == "replace"]
    pairs += [(ref_spk[i0], hyp_spk[j0]) for t, i0, _, j0, _ in ops if t == "equal"]
    if not pairs:
        return dist / len(ref_words), dist, len(ref_words), 1.0, None, None
    r, h = zip(*pairs)
    ref_labels, hyp_labels, acc = sequence_match(r, h)
    return dist / len(ref_words), dist, len(ref_words), 1.0 - acc, ref_labels, hyp_labels


def convert_to_wder_format(speaker_utterances, tokenizer=str.split):
    """[(utterance, speaker id | (embedding, id) | None)] -> ([(word, relative speaker index)], n_speakers).
    tal/wder.py:82-147 without the HDBSCAN branch (clustering searches are out of scope, hdbscan is
    absent).  Two behaviours of the reference are kept on purpose: speakers are numbered in order of first
    appearance, and a `None` speaker stays a label of its own -- the reference computes a forward-filled
    list (`s_u_filled`, :107-122) but builds its output from the unfilled one (:137-146).  The reference
    tokenises with nltk's `word_tokenize` (absent here); the default below splits on whitespace."""
    if not speaker_utterances:
        raise ValueError("no utterances")
    if isinstance(speaker_utterances[0][-1], tuple):
        speaker_utterances = [(u, spk_id) for u, (_, spk_id) in speaker_utterances]
    seen, out = [], []
    for utt, spk in speaker_utterances:
        if spk not in seen:
            seen.append(spk)
        k = seen.index(spk)
        out.extend((w, k) for w in tokenizer(utt))
    return out, len(seen)


def wder_segment(ref_utts, hyp_utts, tokenizer=str.split):
    """tal/wder.py:236-256 -> ([distance, n_ref], [ref_labels, hyp_labels], wder)."""
    ref, _ = convert_to_wder_format(ref_utts, tokenizer)
    hyp, _ = convert_to_wder_format(hyp_utts, tokenizer)
    _, dist, n_ref, wder, rl, hl = calculate_wder(ref, hyp)
    return [dist, n_ref], [rl, hl], wder


def corpus_wder(paired_results, tokenizer=str.split):
    """tal/wder.py:259-288 over the pickle schema [(ref_utts, hyp_utts)] (:313-352): pairs with an empty
    side are skipped, overall WDER = mean of per-segment WDERs, overall WER = sum(distances) / sum(n_ref).
    -> (overall_wder, overall_wer, per-segment wders)."""
    res = [wder_segment(r, h, tokenizer) for r, h in paired_results if r and h]
    if not res:
        raise ValueError("no scorable (reference, hypothesis) pair")
    wders = [w for _, _, w in res]
    dist = sum(c[0] for c, _, _ in res)
    n = sum(c[1] for c, _, _ in res)
    return float(np.mean(wders)), dist / n, wders


def tokens_to_words(tokens, vocab_size, bos_token_id=0, eos_token_id=1):
    """Turn a generated token stream into (token, turn-index) 'words': each EOS closes a speaker
    turn (tal/asr/tokenizers/__init__.py:103-138).  With the sentencepiece model absent, token ids
    stand in for words; the turn index stands in for the clustered speaker label."""
    from .util import split_speaker_turns
    turns, _ = split_speaker_turns(tokens, vocab_size, bos_token_id, eos_token_id)
    return [(t, k if spk is None else spk) for k, (toks, spk) in enumerate(turns) for t in toks]
