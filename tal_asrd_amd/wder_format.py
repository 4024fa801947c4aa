"""Generation results -> WDER scorer input: the counterpart of tal/utils/aligned_to_wder_format.py
(SURVEY.md section 8f item 2).

Input schema (that script's docstring, :1-47; produced by `System.test_step`, tal/asr/system.py:688-707):
    [(ref_utterances, hyp_utterances)], hyp utterance = {'utterance', 'speakerId', 'attention' [n_tok, S],
    'chunkStart' [n_tok], 'utteranceTokens'}
Output schema: [(ref_examples, hyp_examples)] per episode with
    ref = (text, speaker, role), hyp = (text, (embedding [n, E] on the CPU, speaker id), role).

Device side (csrc/pool.hip): the attention-weighted pooling of the SDModel features (:203-213), the
attention-weighted speaker vote of a word (:150-196) and the majority vote of an aligned segment (:330-333)
are HIP kernels.  The reference computes the pooling in half precision on its GPU (`.half()`, :74,159,206);
`half_mode=True` (the default here) reproduces that rounding, so embeddings agree to one fp16 ulp and votes
agree exactly.  Host side: word segmentation through the tokenizer (:82-96), role maps, grouping.
"""
from collections import defaultdict

import numpy as np
import torch

from . import _native as N
from . import ops

STRIDE = 0.08   # aligned_to_wder_format.py:291
CHUNK = 357     # the decoder's cross-attention window in encoder frames (:157,204,358)


def _cs_tensor(chunk_start, device):
    cs = chunk_start if torch.is_tensor(chunk_start) else torch.as_tensor(np.asarray(chunk_start))
    return cs.to(device=device, dtype=torch.int64).contiguous()


def pool_attention_features(attention, chunk_start, features, half_mode=False):
    """attention [N, S] (rows as returned by generate_unaligned), chunk_start [N] (int),
    features [T', E] (SDModel.spk_embed_proj output of the episode) -> [N, E]:
    emb[n] = attention[n, :len] @ features[cs:cs+S]  (aligned_to_wder_format.py:203-213)."""
    lib = N.lib()
    attention = ops._f32c(attention, "pool_attention_features(attention)")
    features = ops._f32c(features, "pool_attention_features(features)")
    cs = _cs_tensor(chunk_start, attention.device)
    n, s = attention.shape
    t, e = features.shape
    out = torch.empty(n, e, dtype=torch.float32, device=attention.device)
    N.check(lib.tal_attn_pool_fwd(N.ptr(attention), N.ptr(cs), N.ptr(features), t, e, n, s, 1 if half_mode else 0,
                                  N.ptr(out), N.stream_handle()), "tal_attn_pool_fwd")
    return out


def vote_speaker_ids(attention, chunk_start, frame_ids):
    """Most heavily attended speaker id per token: returns (ids [N] int32, weight [N])."""
    lib = N.lib()
    attention = ops._f32c(attention, "vote_speaker_ids(attention)")
    cs = _cs_tensor(chunk_start, attention.device)
    ids = frame_ids.to(device=attention.device, dtype=torch.int32).contiguous()
    n, s = attention.shape
    out = torch.empty(n, dtype=torch.int32, device=attention.device)
    wgt = torch.empty(n, dtype=torch.float32, device=attention.device)
    N.check(lib.tal_attn_vote_fwd(N.ptr(attention), N.ptr(cs), N.ptr(ids), ids.numel(), n, s, N.ptr(out), N.ptr(wgt),
                                  N.stream_handle()), "tal_attn_vote_fwd")
    return out, wgt


def vote_speaker_ids_grouped(attention, chunk_start, frame_ids, group_offsets, num_ids, half_mode=True):
    """Attention-weighted speaker vote per group of consecutive tokens (one word, aligned_to_wder_format.py:150-196):
    group g covers tokens [group_offsets[g], group_offsets[g+1]).  -> (ids [G] int32, -1 for an empty group;
    weights [G] float64)."""
    lib = N.lib()
    attention = ops._f32c(attention, "vote_speaker_ids_grouped(attention)")
    dev = attention.device
    cs = _cs_tensor(chunk_start, dev)
    ids = frame_ids.to(device=dev, dtype=torch.int32).contiguous()
    off = _cs_tensor(group_offsets, dev)
    g = off.numel() - 1
    if g < 0 or int(off[-1]) > attention.shape[0]:
        raise N.NativeError("vote_speaker_ids_grouped: group offsets run past the %d attention rows" % attention.shape[0])
    out = torch.empty(max(g, 0), dtype=torch.int32, device=dev)
    wgt = torch.empty(max(g, 0), dtype=torch.float64, device=dev)
    N.check(lib.tal_attn_vote_groups_fwd(N.ptr(attention), N.ptr(cs), N.ptr(ids), ids.numel(), attention.shape[1],
                                         N.ptr(off), g, int(num_ids), 1 if half_mode else 0, N.ptr(out), N.ptr(wgt),
                                         N.stream_handle()), "tal_attn_vote_groups_fwd")
    return out, wgt


def majority_vote(frame_ids, ranges, num_ids):
    """Most frequent speaker id in ids[start:end] for every (start, end) of `ranges` [G, 2]
    (Counter(...).most_common(1), aligned_to_wder_format.py:330-333) -> (ids [G] int32, counts [G] float64)."""
    lib = N.lib()
    N.require_cuda(frame_ids, "majority_vote")
    dev = frame_ids.device
    ids = frame_ids.to(dtype=torch.int32).contiguous()
    r = _cs_tensor(ranges, dev).reshape(-1, 2).contiguous()
    g = r.shape[0]
    out = torch.empty(g, dtype=torch.int32, device=dev)
    cnt = torch.empty(g, dtype=torch.float64, device=dev)
    N.check(lib.tal_majority_vote_fwd(N.ptr(ids), ids.numel(), N.ptr(r), g, int(num_ids), N.ptr(out), N.ptr(cnt),
                                      N.stream_handle()), "tal_majority_vote_fwd")
    return out, cnt


# ----------------------------------------------------------------------------------------------
# host side: hypothesis dicts -> scorer tuples
# ----------------------------------------------------------------------------------------------
def _window_lens(chunk_start, t_frames, s=CHUNK):
    """len(features[cs : cs + S]) with python slice semantics, per token."""
    out = []
    for c in chunk_start:
        a, b = slice(int(c), int(c) + s).indices(t_frames)[:2]
        out.append(max(b - a, 0))
    return out


def _check_stackable(lens, what):
    """The reference stacks the (attention[:len], feature chunk) pairs of a group with torch.stack
    (aligned_to_wder_format.py:168-170,209-211), which raises when a window runs past the end of the episode
    for some tokens of the group only; the same inputs raise here."""
    if len(set(lens)) > 1:
        bad = next(i for i, l in enumerate(lens) if l != lens[0])
        raise RuntimeError("stack expects each tensor to be equal size, but got [%d] at entry 0 and [%d] at entry %d (%s)"
                           % (lens[0], lens[bad], bad, what))


def split_words(utterance_tokens, tok):
    """Word boundaries inside an utterance's token list, exactly as the loop at
    aligned_to_wder_format.py:82-96: a word is dumped when the tokens since the last dump decode to text containing
    a space; the trailing word of an utterance is never dumped.  -> [(start, end, text)] (token index ranges)."""
    words = []
    buffer = []
    last_dump_ix = 0
    for i_u, u in enumerate(utterance_tokens):
        buffer.append(u)
        if buffer and " " in tok.decode(buffer[last_dump_ix:i_u]):
            words.append((last_dump_ix, i_u, tok.decode(buffer[last_dump_ix:i_u])))
            last_dump_ix = i_u
    return words


def hyp_dict_to_wder(hyp_dict, role_map, tok, ep_features, ep_ids, word_level=False, num_ids=None, half_mode=True):
    """get_hyp_dict_wder (aligned_to_wder_format.py:65-224) for one hypothesis utterance.
    ep_features [T', E] and ep_ids [T'] are DEVICE tensors (SDModel.speaker_ids of the episode).
    -> list of (text, (embedding [n, E] CPU tensor, speaker id), role)."""
    speaker_id = hyp_dict["speakerId"]
    role = role_map.get(speaker_id, "subject")
    dev = ep_features.device
    attn = torch.as_tensor(np.asarray(hyp_dict["attention"], dtype=np.float32)).to(dev)
    cs = np.asarray(hyp_dict["chunkStart"]).astype(np.int64).reshape(-1)
    t_frames = ep_features.shape[0]
    lens = _window_lens(cs, t_frames, attn.shape[1])
    if not word_level:
        _check_stackable(lens, "utterance")
        emb = pool_attention_features(attn, cs, ep_features, half_mode=half_mode)
        return [(hyp_dict["utterance"], (emb.cpu(), speaker_id), role)]
    words = split_words(hyp_dict["utteranceTokens"], tok)
    if not words:
        return []
    for a, b, _ in words:
        _check_stackable(lens[a:b], "word")
    if any(b <= a for a, b, _ in words):
        raise ValueError("not enough values to unpack (expected 2, got 0)")   # zip(*[]) at :168 for an empty word
    emb = pool_attention_features(attn, cs, ep_features, half_mode=half_mode).cpu()
    offsets = np.asarray([w[0] for w in words] + [words[-1][1]], dtype=np.int64)
    # words are consecutive token ranges by construction (each dump starts where the last one ended)
    assert all(words[i][1] == words[i + 1][0] for i in range(len(words) - 1))
    num_ids = int(ep_ids.max()) + 1 if num_ids is None else num_ids
    vid, _ = vote_speaker_ids_grouped(attn, cs, ep_ids, offsets, num_ids, half_mode=half_mode)
    vid = vid.cpu().tolist()
    word_role = role_map.get(speaker_id, "subject")
    return [(text, (emb[a:b], int(vid[k])), word_role) for k, (a, b, text) in enumerate(words)]


def unaligned_to_wder(utterances, ep_sd_features, ep_sd_ids, role_map, tok, word_level=False, num_ids=None,
                      half_mode=True):
    """The --unaligned branch (aligned_to_wder_format.py:381-427): group by episode, references as
    (text, speaker, role), hypotheses through hyp_dict_to_wder (empty-text hypotheses skipped)."""
    episode_refs, episode_hyps = defaultdict(list), defaultdict(list)
    for ref_utterances, hyp_dicts in utterances:
        ep = ref_utterances[0]["episode"]
        for ref_dict in ref_utterances:
            ep = ref_dict["episode"]
            episode_refs[ep].append((ref_dict["utterance"], ref_dict["speaker"], ref_dict["role"]))
        for hyp_dict in hyp_dicts:
            if hyp_dict["utterance"]:
                episode_hyps[ep].extend(hyp_dict_to_wder(hyp_dict, role_map, tok, ep_sd_features[ep], ep_sd_ids[ep],
                                                         word_level, num_ids, half_mode))
    return [(episode_refs[e], episode_hyps[e]) for e in episode_refs]


def aligned_to_wder(utterances, ep_sd_features, ep_sd_ids, num_ids=None, half_mode=True):
    """The aligned branch (aligned_to_wder_format.py:294-379): one reference utterance per example; the hypothesis
    speaker falls back to the majority vote of the separate diarizer over the utterance's frames, the embedding to
    the frames' features (no attention) or the attention-weighted pooling."""
    episode_refs, episode_hyps = defaultdict(list), defaultdict(list)
    for ref_utterances, hyp_dicts in utterances:
        ref_dict = ref_utterances[0]
        ep = ref_dict["episode"]
        feats = ep_sd_features[ep]
        u_start, u_end = ref_dict["utterance_start"], ref_dict["utterance_end"]
        st_frame = int(u_start / 0.08)
        e_frame = max(int(max(0.0, u_end - 1.0) / 0.08), st_frame + 1)
        episode_refs[ep].append((u_start, ref_dict["utterance"], ref_dict["speaker"], ref_dict["role"]))
        valid = [h for h in hyp_dicts if h["utterance"].strip()]
        if not valid:
            continue
        if len(valid) != 1:
            raise ValueError("aligned example with %d non-empty hypotheses (the reference reuses a stale one)" % len(valid))
        hyp_dict = valid[0]
        spk = hyp_dict.get("speakerId")
        if spk is None and ep_sd_ids:
            ids = ep_sd_ids[ep]
            n = int(ids.max()) + 1 if num_ids is None else num_ids
            vid, _ = majority_vote(ids, [[st_frame, e_frame]], n)
            spk = int(vid[0])
            if spk < 0:      # the reference: Counter([]).most_common(1)[0] -> IndexError (:332-333)
                raise IndexError("aligned utterance [%.2f s, %.2f s) of episode %r lies outside the diarizer's %d frames: "
                                 "no speaker id to vote on (the reference raises 'list index out of range' here)"
                                 % (u_start, u_end, ep, int(ids.numel())))
        if hyp_dict.get("attention") is None and ep_sd_features:
            emb = feats[st_frame:e_frame]            # the reference's features are `.half()` (:313)
            emb = emb.half().float() if half_mode else emb
        else:
            attn = torch.as_tensor(np.asarray(hyp_dict["attention"], dtype=np.float32)).to(feats.device)
            cs = np.asarray(hyp_dict["chunkStart"]).astype(np.int64).reshape(-1)
            _check_stackable(_window_lens(cs, feats.shape[0], attn.shape[1]), "utterance")
            emb = pool_attention_features(attn, cs, feats, half_mode=half_mode)
        episode_hyps[ep].append((u_start, hyp_dict["utterance"], (emb.cpu(), spk), ref_dict["role"]))
    out = []
    for e in episode_refs:
        refs = [(u, s, r) for _, u, s, r in sorted(episode_refs[e], key=lambda x: x[0])]
        hyps = [(u, s, r) for _, u, s, r in sorted(episode_hyps[e], key=lambda x: x[0])]
        out.append((refs, hyps))
    return out


def strip_roles(wder_input):
    """tal/wder.py scores (utterance, speaker) pairs (:313-352); the role element is for the role-naming tools."""
    return [([(u, s) for u, s, _ in refs], [(u, s) for u, s, _ in hyps]) for refs, hyps in wder_input]
