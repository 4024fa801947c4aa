"""UIS-RNN inference (tal_asrd_amd.uisrnn.UISRNN.predict_single) against cluster traces recorded from
the reference's own UISRNN.predict_single (tal/diarization/uisrnn/uisrnn.py:470-554; make_golden.py,
section `uisrnn`).  The CPU test checks the restated beam-search bookkeeping with the oracle's CoreRNN
standing in for the HIP one; the GPU test runs the product path (batched tal_gru_cell_fwd)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tal_asrd_amd import synth
from tests.conftest import has_gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = {"gru.weight_ih_l0": (1536, 256), "gru.weight_hh_l0": (1536, 512), "gru.bias_ih_l0": (1536,),
          "gru.bias_hh_l0": (1536,), "linear_mean1.weight": (512, 512), "linear_mean1.bias": (512,),
          "linear_mean2.weight": (256, 512), "linear_mean2.bias": (256,)}


def cases():
    with open(os.path.join(HERE, "golden", "uisrnn_predict.json")) as f:
        return json.load(f)


def weights(case):
    if case["weights"] == "echo":
        return synth.uisrnn_echo_state_dict()
    sh = dict(SHAPES)
    if case["depth"] == 2:
        sh.update({"gru.weight_ih_l1": (1536, 512), "gru.weight_hh_l1": (1536, 512),
                   "gru.bias_ih_l1": (1536,), "gru.bias_hh_l1": (1536,)})
    pre = case["weights"]
    sd = synth.fill_state_dict({pre + k: v for k, v in sh.items()})
    return {k[len(pre):]: v for k, v in sd.items()}


def model_args(case):
    return SimpleNamespace(observation_dim=256, rnn_hidden_size=512, rnn_depth=case["depth"], rnn_dropout=0,
                           sigma2=case["sigma2"], transition_bias=case["transition_bias"], crp_alpha=case["crp_alpha"])


def infer_args(case):
    return SimpleNamespace(beam_size=case["beam_size"], look_ahead=case["look_ahead"],
                           test_iteration=case["test_iteration"])


def sequence(case):
    x, labels = synth.uisrnn_sequence(case["n_obs"], 256, case["n_speakers"], case["seed"], case["noise"])
    assert labels == case["truth"]
    return x


@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_beam_search_bookkeeping_with_oracle_cell(case):
    from oracle import tal_oracle as O
    from tal_asrd_amd.uisrnn import UISRNN
    sd = weights(case)
    calls = []

    def cell(x, h):
        calls.append(x.shape[1])
        return O.core_rnn(x, h, sd, depth=case["depth"])
    m = UISRNN(model_args(case), rnn_model=cell, device="cpu")
    pred = m.predict_single(sequence(case), infer_args(case))
    assert [int(c) for c in pred] == case["pred"]
    if case["look_ahead"] == 1:
        # one batched GRU call per observation (+1 for the new-cluster prior), never more rows than beams
        assert len(calls) == 1 + case["n_obs"] * case["test_iteration"]
        assert max(calls) <= case["beam_size"]


def test_argument_checks():
    from tal_asrd_amd.uisrnn import UISRNN
    c = cases()[0]
    m = UISRNN(model_args(c), rnn_model=lambda x, h: None, device="cpu")
    with pytest.raises(TypeError):
        m.predict_single(np.zeros((4, 256), dtype=np.float32), infer_args(c))
    with pytest.raises(ValueError):
        m.predict_single(np.zeros(256), infer_args(c))
    with pytest.raises(ValueError):
        m.predict_single(np.zeros((4, 8)), infer_args(c))
    with pytest.raises(TypeError):
        m.predict("nope", infer_args(c))


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_predict_single_on_gpu(case):
    from tal_asrd_amd.uisrnn import UISRNN
    m = UISRNN(model_args(case), device="cuda:0")
    own = m.rnn_model.state_dict()
    for k, v in weights(case).items():
        own[k] = torch.from_numpy(np.array(v, copy=True))
    m.rnn_model.load_state_dict(own)
    m.rnn_model.to("cuda:0")
    pred = m.predict(sequence(case), infer_args(case))
    assert [int(c) for c in pred] == case["pred"]
