"""GPU parity of the attention-weighted speaker pooling / voting kernels (csrc/pool.hip) and of the host logic
around them (tal_asrd_amd/wder_format.py) against outputs of the reference's OWN get_hyp_dict_wder
(tal/utils/aligned_to_wder_format.py:65-224) recorded by tests/golden/make_golden_episode.py --unit
(tests/golden/pool_unit.*), plus a float64 numpy cross-check of the fp32 mode."""
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN, golden, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]
HALF_ULP = 2.0 ** -10     # relative spacing of fp16


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("case", ["long", "short_wrap", "short_zero"])
def test_hyp_dict_to_wder_matches_reference(case):
    """Utterance-level and word-level outputs for one hypothesis: texts, word boundaries, voted speaker ids
    identical; embeddings within one fp16 ulp (the reference pools in half precision)."""
    from tal_asrd_amd.tokenizer import SynthTokenizer
    from tal_asrd_amd.wder_format import hyp_dict_to_wder
    g = golden("pool_unit")
    with open(os.path.join(GOLDEN, "pool_unit.json")) as f:
        meta = {m["name"]: m for m in json.load(f)}[case]
    tok = SynthTokenizer()
    feat = torch.from_numpy(g[case + "_feat"]).to(_dev())
    ids = torch.from_numpy(g[case + "_ids"]).to(_dev())
    toks = g[case + "_tokens"].tolist()
    hyp = {"utterance": tok.decode(toks), "speakerId": None, "attention": torch.from_numpy(g[case + "_attn"]),
           "chunkStart": torch.from_numpy(g[case + "_cs"]), "utteranceTokens": toks}
    assert hyp["utterance"] == meta["utterance"]
    (text, (emb, spk), role), = hyp_dict_to_wder(hyp, {}, tok, feat, ids, word_level=False)
    assert (text, spk, role) == (meta["utterance"], None, meta["role"])
    want = g[case + "_utt_emb"]
    np.testing.assert_allclose(emb.numpy(), want, rtol=HALF_ULP, atol=2.0 ** -24)
    words = hyp_dict_to_wder(hyp, {}, tok, feat, ids, word_level=True, num_ids=9)
    assert [w[0] for w in words] == meta["words"]
    assert [w[1][1] for w in words] == g[case + "_word_spk"].tolist()
    assert [w[1][0].shape[0] for w in words] == g[case + "_word_ntok"].tolist()
    if words:
        np.testing.assert_allclose(np.concatenate([w[1][0].numpy() for w in words]), g[case + "_word_emb"], rtol=HALF_ULP,
                                   atol=2.0 ** -24)


def test_ragged_windows_raise_like_the_reference():
    """torch.stack of unequal slices (aligned_to_wder_format.py:209-211) raises in the reference when only some
    tokens of an utterance have windows that run past the episode end; so does the counterpart."""
    from tal_asrd_amd.tokenizer import SynthTokenizer
    from tal_asrd_amd.wder_format import hyp_dict_to_wder
    rng = np.random.RandomState(0)
    feat = torch.from_numpy(rng.randn(500, 128).astype(np.float32)).to(_dev())
    ids = torch.zeros(500, dtype=torch.int32, device=_dev())
    hyp = {"utterance": "x", "speakerId": None, "attention": torch.rand(3, 357), "chunkStart": torch.tensor([0, 100, 200]),
           "utteranceTokens": [1, 5, 6]}
    with pytest.raises(RuntimeError, match="stack expects each tensor to be equal size"):
        hyp_dict_to_wder(hyp, {}, SynthTokenizer(), feat, ids)


def test_majority_vote_matches_counter():
    from tal_asrd_amd.wder_format import majority_vote
    g = golden("pool_unit")
    vid, cnt = majority_vote(torch.from_numpy(g["major_ids"]).to(_dev()), g["major_ranges"], 5)
    assert vid.cpu().tolist() == g["major_votes"].tolist()
    ids = g["major_ids"].tolist()
    for (a, b), v, c in zip(g["major_ranges"].tolist(), vid.cpu().tolist(), cnt.cpu().tolist()):
        assert c == (ids[a:b].count(v) if v >= 0 else 0)


def test_pool_and_vote_fp32_match_numpy():
    from tal_asrd_amd.wder_format import pool_attention_features, vote_speaker_ids, vote_speaker_ids_grouped
    rng = np.random.RandomState(0)
    T, E, S, N = 1000, 128, 357, 40
    feat = rng.randn(T, E).astype(np.float32)
    ids = rng.randint(0, 7, size=T).astype(np.int32)
    attn = rng.rand(N, S).astype(np.float32)
    attn /= attn.sum(-1, keepdims=True)
    cs = rng.randint(0, T - 100, size=N).astype(np.int64)   # some windows run past the end of the episode
    cs[0], cs[1], cs[2] = 0, T - 5, -200                    # ... and one starts before it (python slices wrap)
    dev = _dev()
    emb = pool_attention_features(torch.from_numpy(attn).to(dev), torch.from_numpy(cs), torch.from_numpy(feat).to(dev))
    vid, vw = vote_speaker_ids(torch.from_numpy(attn).to(dev), torch.from_numpy(cs), torch.from_numpy(ids).to(dev))
    offs = np.asarray([0, 1, 4, 4, 17, N])
    gid, gw = vote_speaker_ids_grouped(torch.from_numpy(attn).to(dev), cs, torch.from_numpy(ids).to(dev), offs, 7, half_mode=False)
    per_tok = []
    for n in range(N):
        chunk = feat[cs[n]:cs[n] + S]
        want = attn[n, :len(chunk)].astype(np.float64) @ chunk.astype(np.float64)
        np.testing.assert_allclose(emb[n].cpu().numpy(), want, atol=1e-5, rtol=1e-5)
        w = {}
        for a, sid in zip(attn[n, :len(chunk)], ids[cs[n]:cs[n] + S]):
            w[int(sid)] = w.get(int(sid), 0.0) + float(a)
        per_tok.append(w)
        if not w:                      # empty python slice (n == 2): nothing to vote on
            assert int(vid[n]) == -1
            continue
        best = max(w.items(), key=lambda kv: kv[1])
        assert int(vid[n]) == best[0]
        assert abs(float(vw[n]) - best[1]) < 1e-4
    for k in range(len(offs) - 1):
        tot = {}
        for n in range(offs[k], offs[k + 1]):
            for sid, v in per_tok[n].items():
                tot[sid] = tot.get(sid, 0.0) + v
        if not tot:
            assert int(gid[k]) == -1
            continue
        best = max(tot.items(), key=lambda kv: kv[1])
        assert int(gid[k]) == best[0] and abs(float(gw[k]) - best[1]) < 1e-9


def test_aligned_to_wder_matches_reference():
    """The ALIGNED branch (tal/utils/aligned_to_wder_format.py:294-379) against the pickle the reference script itself
    wrote for the same test_result.pkl (tests/golden/aligned_unit.*, make_golden_episode.py --unit): per episode the
    references and hypotheses in the script's order (sorted by utterance_start), texts / roles / speaker ids identical --
    given ids and Counter(...).most_common(1) votes over ids[st_frame:e_frame] alike -- and the embeddings (attention
    pooling or the plain feature slice, both in the reference's half precision) within one fp16 ulp."""
    from tal_asrd_amd.wder_format import aligned_to_wder
    g = golden("aligned_unit")
    with open(os.path.join(GOLDEN, "aligned_unit.json")) as f:
        meta = json.load(f)
    feats = {e: torch.from_numpy(g["feat_" + e]).to(_dev()) for e in meta["episodes"]}
    ids = {e: torch.from_numpy(g["ids_" + e]).to(_dev()) for e in meta["episodes"]}
    utterances = []
    for k, ex in enumerate(meta["examples"]):
        ref = {"episode": ex["episode"], "utterance": ex["ref_utterance"], "speaker": ex["ref_speaker"], "role": ex["role"],
               "utterance_start": ex["utterance_start"], "utterance_end": ex["utterance_end"]}
        hyps = []
        for h in ex["hyps"]:
            d = {"utterance": h["utterance"], "speakerId": h["speakerId"]}
            if h["has_attention"]:
                d["attention"] = torch.from_numpy(g["attn_%d" % k])
                d["chunkStart"] = torch.from_numpy(g["cs_%d" % k])
            hyps.append(d)
        utterances.append(([ref], hyps))
    got = aligned_to_wder(utterances, feats, ids, num_ids=6)
    assert len(got) == len(meta["wder_input"])
    n_votes = 0
    for gi, ((refs, hyps), want) in enumerate(zip(got, meta["wder_input"])):
        assert [[u, s, r] for u, s, r in refs] == want["refs"]
        assert len(hyps) == len(want["hyps"])
        for hi, ((u, (emb, spk), r), w) in enumerate(zip(hyps, want["hyps"])):
            assert (u, int(spk), r) == (w["utterance"], w["speaker"], w["role"]), (gi, hi)
            n_votes += w["speaker"] != 7
            ref_emb = g["out_emb_%d_%d" % (gi, hi)]
            assert tuple(emb.shape) == ref_emb.shape, (gi, hi)
            np.testing.assert_allclose(emb.float().numpy(), ref_emb, rtol=HALF_ULP, atol=2.0 ** -24)
    assert n_votes >= 3          # the fixture exercises the majority-vote fallback, not only given ids


def test_aligned_vote_outside_the_episode_raises_like_the_reference():
    """ids[st_frame:e_frame] empty -> the reference's Counter([]).most_common(1)[0] raises IndexError; the counterpart must
    not hand the scorer a speaker id of -1."""
    from tal_asrd_amd.wder_format import aligned_to_wder
    feats = {"e": torch.zeros(100, 128, device=_dev())}
    ids = {"e": torch.zeros(100, dtype=torch.int32, device=_dev())}
    ref = {"episode": "e", "utterance": "a b", "speaker": 1, "role": "host", "utterance_start": 50.0, "utterance_end": 55.0}
    with pytest.raises(IndexError):
        aligned_to_wder([([ref], [{"utterance": "a b", "speakerId": None}])], feats, ids, num_ids=4)
