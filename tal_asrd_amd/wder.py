"""WER / WDER scoring (host side), restated from
tal/wder.py:41-73 (optimal speaker matching), :150-163 (WER), :166-234 (WDER,
https://arxiv.org/pdf/1907.05337.pdf):

    WER  = Levenshtein(ref_words, hyp_words) / len(ref_words)
    WDER = 1 - (best one-to-one speaker-label matching accuracy over the word pairs the
               ASR alignment marks as correct or substituted)

The reference gets the distance from `editdistance` and the alignment opcodes from
`edit_distance.SequenceMatcher` (neither is installed here, neither is vendored): the
distance is unique, but among equally cheap alignments the opcode choice of that library is
restated here from the library's published source (from memory: the package is absent), so WDER parity
on texts with ambiguous alignments is UNPINNED with respect to the real package; `align_opcodes` documents the rule.
When hypothesis and reference token streams are identical (the parity statement of this
repository: identical tokens and speaker-change indices), every alignment is the diagonal and
WER / WDER are identical by construction.
"""
import numpy as np
from scipy import optimize


def levenshtein(a, b):
    """Edit distance with unit costs (what `editdistance.eval` returns, tal/wder.py:160,192)."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        ai = a[i - 1]
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            d = prev[j - 1] + (0 if ai == b[j - 1] else 1)
            up, left = prev[j] + 1, cur[j - 1] + 1
            cur[j] = d if d <= up and d <= left else (up if up <= left else left)
        prev = cur
    return prev[-1]


def align_opcodes(a, b):
    """[(tag, i0, i1, j0, j1)] with single-element spans; tags: equal / replace / delete / insert.

    Follows `edit_distance.SequenceMatcher(a, b).get_opcodes()` (belambert/edit-distance 1.0.x, the package
    tal/wder.py:5,201 imports; not installed here, restated from its published source): one dynamic programme
    carries (cost, number of matches) per cell and picks the step with `highest_match_action` -- the candidate with
    the most matches so far wins; the diagonal (equal / replace) is tried first, then insert, then delete -- and
    the opcodes are read back from the back-pointer table.  The alignment therefore maximises matched words, it
    is not necessarily a minimum-cost one (the WER itself comes from the exact Levenshtein distance)."""
    a, b = list(a), list(b)
    m, n = len(a), len(b)
    cost = [[0] * (n + 1) for _ in range(m + 1)]
    match = [[0] * (n + 1) for _ in range(m + 1)]
    back = [[0] * (n + 1) for _ in range(m + 1)]     # 0 diagonal, 1 insert, 2 delete
    for i in range(1, m + 1):
        cost[i][0] = i
        back[i][0] = 2
    for j in range(1, n + 1):
        cost[0][j] = j
        back[0][j] = 1
    for i in range(1, m + 1):
        ai = a[i - 1]
        ci, cp, mi, mp, bi = cost[i], cost[i - 1], match[i], match[i - 1], back[i]
        for j in range(1, n + 1):
            c = 0 if ai == b[j - 1] else 1
            sm, im, dm = mp[j - 1] + (1 - c), mi[j - 1], mp[j]
            mx = sm if sm >= im else im
            if dm > mx:
                mx = dm
            if mx == sm:
                ci[j], mi[j], bi[j] = cp[j - 1] + c, sm, 0
            elif mx == im:
                ci[j], mi[j], bi[j] = ci[j - 1] + 1, im, 1
            else:
                ci[j], mi[j], bi[j] = cp[j] + 1, dm, 2
    ops = []
    i, j = m, n
    while i > 0 or j > 0:
        step = back[i][j]
        if step == 0:
            ops.append(("equal" if a[i - 1] == b[j - 1] else "replace", i - 1, i, j - 1, j))
            i, j = i - 1, j - 1
        elif step == 1:
            ops.append(("insert", max(i - 1, 0), max(i - 1, 0), j - 1, j))    # (the library's index convention)
            j -= 1
        else:
            ops.append(("delete", i - 1, i, max(j - 1, 0), max(j - 1, 0)))
            i -= 1
    return ops[::-1]


def sequence_match(seq1, seq2):
    """compute_sequence_match, tal/wder.py:41-73: optimal one-to-one label matching (Hungarian) between two
    equally long label sequences -> (row_index, col_index, accuracy): indices into the SORTED unique labels of
    seq1 (rows) and seq2 (columns), as the reference returns them."""
    if not isinstance(seq1, list) or not isinstance(seq2, list):
        raise TypeError("sequence1 and sequence2 must be lists")
    if not seq1 or len(seq1) != len(seq2):
        raise ValueError("sequence1 and sequence2 must have the same non-zero length")
    u1, u2 = sorted(set(seq1)), sorted(set(seq2))
    i1 = {v: k for k, v in enumerate(u1)}
    i2 = {v: k for k, v in enumerate(u2)}
    counts = np.zeros((len(u1), len(u2)))
    for x, y in zip(seq1, seq2):
        counts[i1[x], i2[y]] += 1.0
    rows, cols = optimize.linear_sum_assignment(-counts)
    return rows, cols, counts[rows, cols].sum() / len(seq1)


def matched_labels(seq1, seq2):
    """{label of seq2: matched label of seq1} of the optimal matching (convenience over sequence_match)."""
    rows, cols, _ = sequence_match(list(seq1), list(seq2))
    u1, u2 = sorted(set(seq1)), sorted(set(seq2))
    return {u2[c]: u1[r] for r, c in zip(rows, cols)}


def calculate_wer(ref, hyp):
    """ref / hyp: lists of (word, speaker).  tal/wder.py:150-163 -> (wer, distance, n_ref)."""
    ref_words = [w for w, _ in ref]
    hyp_words = [w for w, _ in hyp]
    dist = levenshtein(ref_words, hyp_words)
    return dist / len(ref_words), dist, len(ref_words)


def calculate_wder(ref, hyp, wer_only=False, strict=True):
    """tal/wder.py:166-234 -> (wer, distance, n_ref, wder, ref_labels, hyp_labels); the labels are the
    (row_index, col_index) of sequence_match.  Like the reference it raises ValueError when the alignment has no
    substituted or no correct word pair (`zip(*[])`, :221-222) -- a perfect hypothesis included; strict=False
    scores those cases instead (WDER over the pairs there are, 1.0 when there is none)."""
    ref_words, ref_spk = [w for w, _ in ref], [s for _, s in ref]
    hyp_words, hyp_spk = [w for w, _ in hyp], [s for _, s in hyp]
    dist = levenshtein(ref_words, hyp_words)          # editdistance.eval (tal/wder.py:192)
    wer = dist / len(ref_words)
    if wer_only:
        return wer, dist, len(ref_words), 1e8, None, None
    ops = align_opcodes(ref_words, hyp_words)         # edit_distance.SequenceMatcher opcodes (:201-217)
    sub = [(ref_spk[i0], hyp_spk[j0]) for t, i0, _, j0, _ in ops if t == "replace"]
    cor = [(ref_spk[i0], hyp_spk[j0]) for t, i0, _, j0, _ in ops if t == "equal"]
    if (not sub or not cor) and strict:
        raise ValueError("not enough values to unpack (expected 2, got 0)")
    if not sub and not cor:
        return wer, dist, len(ref_words), 1.0, None, None
    pairs = sub + cor
    ref_labels, hyp_labels, acc = sequence_match([r for r, _ in pairs], [h for _, h in pairs])
    return wer, dist, len(ref_words), 1 - acc, ref_labels, hyp_labels


def convert_to_wder_format(speaker_utterances, wer_only=False, tokenizer=str.split, should_cluster=False):
    """[(utterance, speaker id | (embedding, id) | None)] -> ([(word, relative speaker index)], n_speakers).
    tal/wder.py:82-147 without the HDBSCAN branch (clustering searches are out of scope, hdbscan is absent).
    Behaviours of the reference kept on purpose: speakers are numbered in order of first appearance; a `None`
    speaker stays a label of its own -- the reference computes a forward-filled list (`s_u_filled`, :107-122) but
    builds its output from the unfilled one (:137-146); with wer_only the speaker objects are used as they are,
    and an (embedding, id) tuple whose comparison is ambiguous (numpy raises ValueError inside list.index, which
    the reference catches as "not seen yet", :139-143) opens a new label.  The reference tokenises with nltk's
    `word_tokenize` (absent here); the default splits on whitespace."""
    if should_cluster:
        raise NotImplementedError("speaker-embedding clustering (hdbscan) is out of scope")
    if not wer_only and isinstance(speaker_utterances[0][-1], tuple):
        speaker_utterances = [(utt, spk_i) for utt, (_, spk_i) in speaker_utterances]
    assert len(speaker_utterances) > 0
    all_speakers, out = [], []
    for utt, speaker in speaker_utterances:
        try:
            k = all_speakers.index(speaker)
        except ValueError:
            k = len(all_speakers)
            all_speakers.append(speaker)
        out.extend((w, k) for w in tokenizer(utt))
    return out, len(all_speakers)


def wder_segment(ref_utts, hyp_utts, wer_only=False, tokenizer=str.split, strict=True):
    """tal/wder.py:236-256 -> ([distance, n_ref], [ref_labels, hyp_labels], wder).  (The reference passes its
    `tokenizer` keyword to the hypothesis side only; both sides use the same one here.)"""
    ref, _ = convert_to_wder_format(ref_utts, wer_only=True, tokenizer=tokenizer)
    hyp, _ = convert_to_wder_format(hyp_utts, wer_only=wer_only, tokenizer=tokenizer)
    _, dist, n_ref, wder, rl, hl = calculate_wder(ref, hyp, wer_only, strict)
    return [dist, n_ref], [rl, hl], wder


def corpus_wder(paired_results, wer_only=False, tokenizer=str.split, strict=True):
    """tal/wder.py:259-288 over the pickle schema [(ref_utts, hyp_utts)] (:313-352): pairs with an empty
    side are skipped, overall WDER = mean of per-segment WDERs, overall WER = sum(distances) / sum(n_ref).
    -> (overall_wder, overall_wer, per-segment wders, distances, n_words)."""
    res = [wder_segment(r, h, wer_only, tokenizer, strict) for r, h in paired_results if r and h]
    if not res:
        raise ValueError("no scorable (reference, hypothesis) pair")
    wders = [w for _, _, w in res]
    dists = [c[0] for c, _, _ in res]
    ns = [c[1] for c, _, _ in res]
    return float(np.mean(wders)), sum(dists) / sum(ns), wders, dists, ns


def tokens_to_words(tokens, vocab_size, bos_token_id=0, eos_token_id=1):
    """Turn a generated token stream into (token, turn-index) 'words': each EOS closes a speaker
    turn (tal/asr/tokenizers/__init__.py:103-138).  With the sentencepiece model absent, token ids
    stand in for words; the turn index stands in for the clustered speaker label."""
    from .util import split_speaker_turns
    turns, _ = split_speaker_turns(tokens, vocab_size, bos_token_id, eos_token_id)
    return [(t, k if spk is None else spk) for k, (toks, spk) in enumerate(turns) for t in toks]
