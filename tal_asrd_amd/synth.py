"""Deterministic synthetic weights and audio for the acoustic hot path.

There is no trained checkpoint and no This-American-Life audio in this
environment, so every parity fixture, test and bench
run uses procedurally generated inputs.  Everything here is a pure function of
(name, index) built from 64-bit integer hashing, so the container that
generates the golden vectors from the reference modules and the GPU box that
re-generates the same tensors for the HIP path agree bit for bit without
shipping a 274 MB weight blob.

ReZero scalars (`resweight`, `resweight_src`) are initialised to 0 by the
reference (tal/asr/models.py:321,504-505), which would make every residual
branch dead and parity vacuous; the generator forces them non-zero.
"""
import math
import zlib

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 finalisation round on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + _GOLD).astype(np.uint64)
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        x = x ^ (x >> np.uint64(31))
    return x


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def hash_uniform(name: str, n: int, offset: int = 0) -> np.ndarray:
    """n float32 values uniform in [-1, 1), a pure function of (name, index).

    Uses the top 24 bits of the hash so every value is exactly representable
    in float32 (no platform-dependent rounding)."""
    seed = np.uint64(name_seed(name)) << np.uint64(32)
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    h = _splitmix64(idx ^ seed)
    u24 = (h >> np.uint64(40)).astype(np.int64)  # [0, 2^24)
    return ((u24 - (1 << 23)).astype(np.float32)) * np.float32(1.0 / (1 << 23))


def synth_tensor(name: str, shape, bound: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return (hash_uniform(name, n) * np.float32(bound)).reshape(shape)


def rezero_value(name: str) -> float:
    """Non-zero ReZero scalar, a pure function of the key.

    Encoder blocks: [0.20, 0.32].  Decoder layers get O(1) scalars (2.0-2.4 for
    `resweight`, 3.0-3.4 for `resweight_src`): with small ones the tied embedding
    makes greedy decoding echo its input token forever and the cross-attention is
    uniform, so the decode-loop fixtures would only ever exercise one branch."""
    if "decoder.layers." in name:
        base = 2.5 if name.endswith("resweight_src") else 1.5
        return base + 0.05 * (name_seed(name) % 5)
    return 0.20 + 0.02 * (name_seed(name) % 7)


# Output gain applied to the final classifier matrices so the synthetic logits
# have a spread comparable to a trained model's (keeps argmax parity tests from
# being dominated by exact near-ties of tiny random logits).
_HEAD_GAIN = {
    "spk_logit_proj.weight": 24.0,
    "speaker_head.1.weight": 24.0,
    # sharper cross-attention: the window-advance logic of generate_unaligned steers by the
    # centre of mass of these weights (tal/asr/system.py:392-408)
    "multihead_attn.in_proj_weight": 3.0,
    # tied embedding / LM head: small enough that greedy decoding does not just echo its input
    # token (the e.W^T.W.e' self-term scales with gain^2), large enough for a clear arg-max.
    # These gains were chosen so that fp32 and fp64 evaluations of the decoder agree to ~3e-5
    # on the logits (a better-conditioned problem than gain 4 / resweight 2-3: 7e-4).
    "embedding.weight": 0.5,
}


def fill_state_dict(shapes: dict) -> dict:
    """Build a full deterministic state dict.

    `shapes` maps reference state_dict keys -> shape tuples
    (keys as in tal/asr/models.py; see tal_asrd_amd.models.*.state_dict()).
    Rule per key (fan_in = prod(shape[1:])):
      *resweight*            -> rezero_value(key)
      *.pe / window / fb     -> skipped (buffers are computed, not synthesised)
      embedding.weight       -> U(-1/sqrt(embed), 1/sqrt(embed)) (tal/modules.py:18-20) * gain
      *.weight, *.bias, in_proj_* -> U(-1/sqrt(fan_in), 1/sqrt(fan_in)) with the
                                fan_in of the matching weight for biases
    """
    out = {}
    for key, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        leaf = key.split(".")[-1]
        if "resweight" in leaf:
            out[key] = np.full(shape, rezero_value(key), dtype=np.float32)
            continue
        if leaf in ("pe", "window", "fb"):
            continue
        if key.endswith("lm_head.weight") and (key[: -len("lm_head.weight")] + "embedding.weight") in shapes:
            continue  # tied to embedding.weight (tal/asr/models.py:117); aliased below
        if leaf in ("bias", "in_proj_bias"):
            wkey = key[: -len(leaf)] + ("in_proj_weight" if leaf == "in_proj_bias" else "weight")
            wshape = shapes.get(wkey, None)
            fan_in = int(np.prod(wshape[1:])) if wshape is not None and len(wshape) > 1 else shape[0]
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        for suffix, gain in _HEAD_GAIN.items():
            if key.endswith(suffix):
                bound *= gain
        out[key] = synth_tensor(key, shape, bound)
    for key in shapes:
        if key.endswith("lm_head.weight"):
            ekey = key[: -len("lm_head.weight")] + "embedding.weight"
            if ekey in out:
                out[key] = out[ekey]
    return out


def synth_audio(num_samples: int, seed: int = 1234, sr: int = 16000) -> np.ndarray:
    """Deterministic 16 kHz mono float32 waveform in [-0.5, 0.5].

    Three amplitude-modulated sinusoids whose frequencies switch every ~7 s
    ("speaker change"), 0.5 s silences every ~11 s (exact zeros, so the
    log(eps) floor and the global-mean coupling are exercised), plus hashed
    uniform noise with the variance of 0.05*N(0,1)."""
    n = int(num_samples)
    out = np.empty(n, dtype=np.float32)
    chunk = 1 << 22
    two_pi = 2.0 * math.pi
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        t = np.arange(s, e, dtype=np.float64) / sr
        spk = np.floor(t / 7.0 + (seed % 5) * 0.2).astype(np.int64) % 3
        f0 = np.choose(spk, [180.0, 240.0, 130.0])
        f1 = np.choose(spk, [1100.0, 900.0, 1500.0])
        f2 = np.choose(spk, [3300.0, 2700.0, 3900.0])
        am0 = 0.6 + 0.4 * np.sin(two_pi * 2.0 * t)
        am1 = 0.5 + 0.5 * np.sin(two_pi * 3.1 * t + 0.3)
        am2 = 0.5 + 0.5 * np.sin(two_pi * 4.7 * t + 1.1)
        x = (0.50 * am0 * np.sin(two_pi * f0 * t)
             + 0.30 * am1 * np.sin(two_pi * f1 * t + 0.5)
             + 0.15 * am2 * np.sin(two_pi * f2 * t + 1.0))
        noise = hash_uniform("audio/%d" % seed, e - s, offset=s).astype(np.float64)
        x = x + (0.05 * math.sqrt(3.0)) * noise
        silent = np.mod(t + 0.37 * (seed % 7), 11.0) < 0.5
        x[silent] = 0.0
        out[s:e] = (x * (0.5 / 1.1)).astype(np.float32)
    np.clip(out, -0.5, 0.5, out=out)
    return out


def synth_audio_batch(batch: int, num_samples: int, seed: int = 1234, lens=None) -> np.ndarray:
    """[batch, num_samples]; item i uses seed+i; optional ragged `lens` are
    right-padded with zeros (tal/asr/data/aligned.py:250-257 padding convention)."""
    x = np.zeros((batch, num_samples), dtype=np.float32)
    for i in range(batch):
        li = num_samples if lens is None else int(lens[i])
        x[i, :li] = synth_audio(li, seed + i)
    return x


def uisrnn_echo_state_dict(obs_dim: int = 256, hidden: int = 512, noise: float = 0.01, tag: str = "uis-echo") -> dict:
    """CoreRNN(obs_dim, hidden, 1, obs_dim) weights (keys of tal/diarization/uisrnn/uisrnn.py:20-32) for
    which the predicted mean roughly echoes the last observation of a cluster, so that the CRP beam
    search revisits clusters the way a trained model does -- random weights only ever open new
    clusters.  Construction: update gate held near 0 (bias -4), candidate state tanh(0.5 x) in the
    first obs_dim hidden units, mean head 2 * (relu(+h) - relu(-h)); every tensor also carries
    deterministic noise so no term of the cell is exercised with exact zeros."""
    assert hidden >= 2 * obs_dim
    D, H = obs_dim, hidden
    sd = {}
    for name, shape in (("gru.weight_ih_l0", (3 * H, D)), ("gru.weight_hh_l0", (3 * H, H)), ("gru.bias_ih_l0", (3 * H,)),
                        ("gru.bias_hh_l0", (3 * H,)), ("linear_mean1.weight", (H, H)), ("linear_mean1.bias", (H,)),
                        ("linear_mean2.weight", (D, H)), ("linear_mean2.bias", (D,))):
        sd[name] = synth_tensor(tag + "/" + name, shape, noise)
    eye = np.eye(D, dtype=np.float32)
    sd["gru.bias_ih_l0"][H:2 * H] -= 4.0                      # z ~ 0.02: h' ~ n
    sd["gru.weight_ih_l0"][2 * H:2 * H + D, :] += 0.5 * eye   # n[:D] = tanh(0.5 x + ...)
    sd["linear_mean1.weight"][:D, :D] += eye
    sd["linear_mean1.weight"][D:2 * D, :D] -= eye
    sd["linear_mean2.weight"][:, :D] += 2.0 * eye
    sd["linear_mean2.weight"][:, D:2 * D] -= 2.0 * eye
    return sd


def uisrnn_sequence(n_obs: int, obs_dim: int, n_speakers: int, seed: int, noise: float = 0.05, scale: float = 0.5):
    """Speaker-embedding-like test sequence: `n_speakers` fixed centroids, turns of 2-5 observations,
    additive Gaussian noise.  -> (float64 [n_obs, obs_dim], list of true speaker labels)."""
    cent = scale * synth_tensor("uis/centroid/%d" % seed, (n_speakers, obs_dim), 1.0).astype(np.float64)
    rng = np.random.RandomState(seed)
    labels, cur = [], 0
    while len(labels) < n_obs:
        labels += [cur] * int(rng.randint(2, 6))
        cur = (cur + int(rng.randint(1, n_speakers))) % n_speakers
    labels = labels[:n_obs]
    return cent[labels] + noise * rng.randn(n_obs, obs_dim), labels


# ----------------------------------------------------------------------------------------------
# Synthetic text side.  The reference's sentencepiece model (`taltoken-cased.model`) does not ship,
# so token ids are turned into text by a deterministic piece table with sentencepiece's conventions:
# a piece either starts a word (leading U+2581, decoded as a space) or continues one.
# ----------------------------------------------------------------------------------------------
def token_piece(t: int) -> str:
    """Piece of token id t: about two thirds of the ids start a word."""
    t = int(t)
    starts_word = ((t * 2654435761) >> 7) % 3 != 0
    return ("▁" if starts_word else "") + "t%d" % t


def decode_pieces(tokens) -> str:
    """sentencepiece-style DecodeIds over the synthetic piece table (leading space stripped)."""
    s = "".join(token_piece(t) for t in tokens).replace("▁", " ")
    return s[1:] if s.startswith(" ") else s
