"""Windowed batch transcription on top of `System.generate`:
the counterpart of `transcribe_file` / `transcribe_batch` / `splice_strings`
(tal/asr/transcribe.py:29-210).

A long waveform is cut into fixed windows `stride_frames` apart ("frames" are audio samples in
the reference's vocabulary), windows are zero-padded into batches of `batch_size`, each batch
is one `System.generate` call (encoder + beam search on the GPU), unfinished hypotheses
(`None`) are dropped, and -- optionally -- neighbouring window texts are spliced on their
longest common substring inside the overlap region.

Differences from the reference, all outside the arithmetic:
  * audio arrives as a tensor, not a path (audio file loading / resampling / VAD are out of
    scope: torchaudio, webrtcvad are not on the hot path);
  * the reference's `transcribe_batch` passes `beam_width=` / `lm_weight=` to a
    `System.generate` that takes `beam_size` and no LM weight (tal/asr/system.py:67-75), i.e. the
    script is stale against its own System; here `beam_width` is forwarded as `beam_size` and
    there is no language model (none ships);
  * windows are sliced on the device from the resident waveform, so a batch costs no H2D copy.
"""
import math
from difflib import SequenceMatcher

import torch


# ------------------------------------------------------------------ text splicing (host, strings)
def overlap_ix(a, b, word_overlap=5):
    """Character offsets bounding the overlap search: the last `word_overlap` words of `a` and the
    first `word_overlap + 1` words of `b` (one extra for EOS).  tal/asr/transcribe.py:29-32."""
    tail = " ".join(a.split()[-word_overlap:])
    head = " ".join(b.split()[:word_overlap + 1])
    return len(a) - len(tail), len(head)


def splice_ix(a, b, word_overlap=5):
    """(end offset in a, start offset in b) at which to join the two texts: the longest common
    substring between a's tail and b's head, ignored when shorter than 5 characters.
    tal/asr/transcribe.py:35-52 (difflib's `find_longest_match`, junk heuristics included)."""
    lo_a, hi_b = overlap_ix(a, b, word_overlap)
    m = SequenceMatcher(None, a, b).find_longest_match(lo_a, len(a), 0, hi_b)
    if m.size < 5:
        return len(a), 0
    return m.a, m.b


def splice_strings(strs, word_overlap=20):
    """Join the per-window texts.  tal/asr/transcribe.py:55-76 -- including what it does with the
    matched text itself: window i contributes `[start_i : end_i)` where end_i is the START of its
    match with window i+1 and start_{i+1} the START of the same match in window i+1, so the
    matched substring is emitted once (from window i+1)."""
    if len(strs) < 2:
        raise IndexError("splice_strings needs at least two window texts (the reference indexes strs[1])")
    end, start = splice_ix(strs[0], strs[1], word_overlap)
    out = strs[0][:end].strip()
    for i in range(1, len(strs) - 1):
        end, nxt = splice_ix(strs[i], strs[i + 1], word_overlap)
        out += " " + strs[i][start:end].strip()
        start = nxt
    return out + " " + strs[-1][start:].strip()


# ------------------------------------------------------------------ windowing
def window_bounds(n_samples, window_frames, stride_frames):
    """[(start, end)] of the windows `transcribe_file` cuts (tal/asr/transcribe.py:124,135-139):
    n = ceil((n_samples - window) / stride) + 1 windows, `end` not clipped (slicing clips)."""
    n = int(math.ceil((n_samples - window_frames) / stride_frames)) + 1
    return [(stride_frames * i, stride_frames * i + window_frames) for i in range(n)]


def transcribe_batch(batch, system, beam_width=4, length=60, use_eot=True, eot_token_id=None, force_half=True):
    """One `System.generate` call on a list of 1-D waveforms (zero-padded to the longest).
    tal/asr/transcribe.py:172-210.  Returns the list System.generate returns for the sequences
    (CPU LongTensor per window, None where no beam finished).  The reference's call leaves `force_half` at
    System.generate's default (True: the waveform is cast to half, system.py:91-92); it is a parameter here so that a
    caller can keep the fp32 samples."""
    lens = [int(w.numel()) for w in batch]
    longest = max(lens)
    dev = batch[0].device
    audio = torch.zeros(len(batch), longest, dtype=torch.float32, device=dev)
    for i, w in enumerate(batch):
        audio[i, :lens[i]] = w
    tok = system.tokenizer
    if use_eot and eot_token_id is None:
        eot_token_id = getattr(tok, "eot_token_id", None)
        if eot_token_id is None:
            raise ValueError("use_eot=True needs an end-of-transcript token id (tokenizer.eot_token_id)")
    prime = tok.bos_token_id if use_eot else tok.eos_token_id
    generated = torch.full((len(batch), 1), prime, dtype=torch.long, device=dev)
    seqs, _ = system.generate(audio_x=audio, generated=generated,
                              audio_lens=torch.tensor(lens, dtype=torch.long, device=dev), length=length,
                              beam_size=beam_width, terminate_token=eot_token_id if use_eot else None,
                              force_half=force_half)
    return seqs


def transcribe_file(x_wav, system, window_frames, stride_frames, batch_size=15, beam_width=4, length=60,
                    truncate=-1.0, splice=False, use_eot=True, eot_token_id=None, decode=None, force_half=True):
    """tal/asr/transcribe.py:79-169 for a waveform already loaded (1-D float tensor on the GPU).
    `decode` turns a token sequence into text (the reference uses its sentencepiece tokenizer, whose
    model file is not in the repository); without it the token sequences themselves are returned
    and `splice` is refused."""
    if x_wav.dim() != 1:
        raise ValueError("x_wav must be a 1-D waveform")
    if truncate > 0.0:
        x_wav = x_wav[:int(truncate * x_wav.numel())]
    if splice and decode is None:
        raise ValueError("splice=True joins texts: pass decode=")
    outputs, batch = [], []
    bounds = window_bounds(x_wav.numel(), window_frames, stride_frames)
    for i, (s, e) in enumerate(bounds):
        batch.append(x_wav[s:e])
        if len(batch) == batch_size or i == len(bounds) - 1:
            seqs = transcribe_batch(batch, system, beam_width=beam_width, length=length, use_eot=use_eot,
                                    eot_token_id=eot_token_id, force_half=force_half)
            outputs.extend(sq if decode is None else decode(sq) for sq in seqs if sq is not None)
            batch = []
    if splice:
        merge_window = 3 * ((window_frames - stride_frames) // 16000)
        return splice_strings(outputs, merge_window)
    return outputs
