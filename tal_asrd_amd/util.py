"""Host-side helpers of the decode loops (mirror of tal/asr/util.py and the token-level
half of tal/asr/tokenizers/__init__.py:103-138)."""
import numpy as np


def ngram_repeat_mask(xs, n):
    """tal/asr/util.py:5-17: mark every position covered by an n-gram that already occurred
    earlier in the same row.  xs: [B, U] integer array-like -> same-shape 0/1 numpy array.
    (Like the reference, the last n-gram start considered is len-n-1.)"""
    xs = np.asarray(xs)
    mask = np.zeros_like(xs)
    for i, row in enumerate(xs.tolist()):
        seen = set()
        for j in range(len(row) - n):
            gram = tuple(row[j:j + n])
            if gram in seen:
                mask[i, j:j + n] = 1
            seen.add(gram)
    return mask


def split_speaker_turns(tokens, vocab_size, bos_token_id=0, eos_token_id=1, add_last=True):
    """Token-level part of _Tokenizer.decode_speakers (tal/asr/tokenizers/__init__.py:103-138):
    EOS closes an utterance (= a speaker change point), ids >= vocab_size are speaker tokens,
    BOS is ignored.  Returns ([(token_list, speaker | None), ...], split_indices).  The text
    decoding of each token list needs the sentencepiece model, which the reference does not ship."""
    turns, buf, splits, speaker = [], [], [], None
    i = -1
    for i, x in enumerate(tokens):
        x = int(x)
        if x == bos_token_id:
            continue
        if x >= vocab_size:
            speaker = x - vocab_size
        elif x == eos_token_id:
            if buf:
                turns.append((buf, speaker))
                speaker, buf = None, []
                splits.append(i)
        else:
            buf.append(x)
    if buf and add_last:
        turns.append((buf, speaker))
        splits.append(i)
    return turns, splits
