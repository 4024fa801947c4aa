"""GPU check of the attention-weighted speaker pooling ops against a numpy restatement of
tal/utils/aligned_to_wder_format.py:150-214 (no reference-run fixture: that script needs CUDA,
the sentencepiece model and pickled episodes; the arithmetic is a masked matmul and a weighted vote)."""
import numpy as np
import pytest
import torch

from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def test_pool_and_vote_match_numpy():
    from tal_asrd_amd.wder_format import pool_attention_features, vote_speaker_ids
    rng = np.random.RandomState(0)
    T, E, S, N = 1000, 128, 357, 40
    feat = rng.randn(T, E).astype(np.float32)
    ids = rng.randint(0, 7, size=T).astype(np.int32)
    attn = rng.rand(N, S).astype(np.float32)
    attn /= attn.sum(-1, keepdims=True)
    cs = rng.randint(0, T - 100, size=N).astype(np.int64)   # some windows run past the end of the episode
    cs[0], cs[1] = 0, T - 5
    dev = torch.device("cuda:0")
    emb = pool_attention_features(torch.from_numpy(attn).to(dev), torch.from_numpy(cs), torch.from_numpy(feat).to(dev))
    vid, vw = vote_speaker_ids(torch.from_numpy(attn).to(dev), torch.from_numpy(cs), torch.from_numpy(ids).to(dev))
    for n in range(N):
        chunk = feat[cs[n]:cs[n] + S]
        want = attn[n, :len(chunk)].astype(np.float64) @ chunk.astype(np.float64)
        np.testing.assert_allclose(emb[n].cpu().numpy(), want, atol=1e-5, rtol=1e-5)
        w = {}
        for a, sid in zip(attn[n, :len(chunk)], ids[cs[n]:cs[n] + S]):
            w[int(sid)] = w.get(int(sid), 0.0) + float(a)
        best = max(w.items(), key=lambda kv: kv[1])
        assert int(vid[n]) == best[0]
        assert abs(float(vw[n]) - best[1]) < 1e-4
