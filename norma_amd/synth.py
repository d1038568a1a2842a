"""Synthetic inputs for tests and bench (there are no checkpoints or audio offline).

Protocol of SURVEY.md 8(d): seeded PCM clips and seeded random weights in HF tensor names/shapes,
rounded to fp16 so that the CPU oracle (f32 arithmetic on the fp16-rounded values) and the HIP path
(fp16 storage) see bit-identical parameters.
"""
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from .config import Config

N_SAMPLES = 480000  # candle m::N_SAMPLES, used at src/models/whisper/model.rs:69
SAMPLE_RATE = 16000


def synth_pcm(k: int, n: int = N_SAMPLES) -> np.ndarray:
    """Chunk k: 0.3 sin(2pi 440 t) + 0.2 sin(2pi (200+37k) t) + 0.05 N(0,1), clipped to [-1,1]."""
    rng = np.random.default_rng(1234 + k)
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    x = 0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * (200.0 + 37.0 * k) * t)
    x = x + 0.05 * rng.standard_normal(n)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def tensor_specs(cfg: Config) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(HF tensor name, shape, kind) for every tensor candle reads.
    kind: 'w' matrix, 'b' bias, 'lnw' / 'lnb' LayerNorm affine."""
    d, nm = cfg.d_model, cfg.num_mel_bins
    out: List[Tuple[str, Tuple[int, ...], str]] = []

    def lin(prefix, n_out, n_in, bias=True):
        out.append((prefix + ".weight", (n_out, n_in), "w"))
        if bias:
            out.append((prefix + ".bias", (n_out,), "b"))

    def ln(prefix):
        out.append((prefix + ".weight", (d,), "lnw"))
        out.append((prefix + ".bias", (d,), "lnb"))

    def attn(prefix):
        lin(prefix + ".q_proj", d, d)
        lin(prefix + ".k_proj", d, d, bias=False)
        lin(prefix + ".v_proj", d, d)
        lin(prefix + ".out_proj", d, d)

    out.append(("model.encoder.conv1.weight", (d, nm, 3), "w"))
    out.append(("model.encoder.conv1.bias", (d,), "b"))
    out.append(("model.encoder.conv2.weight", (d, d, 3), "w"))
    out.append(("model.encoder.conv2.bias", (d,), "b"))
    for i in range(cfg.encoder_layers):
        p = f"model.encoder.layers.{i}"
        attn(p + ".self_attn"); ln(p + ".self_attn_layer_norm")
        lin(p + ".fc1", 4 * d, d); lin(p + ".fc2", d, 4 * d); ln(p + ".final_layer_norm")
    ln("model.encoder.layer_norm")
    out.append(("model.decoder.embed_tokens.weight", (cfg.vocab_size, d), "w"))
    out.append(("model.decoder.embed_positions.weight", (cfg.max_target_positions, d), "w"))
    for i in range(cfg.decoder_layers):
        p = f"model.decoder.layers.{i}"
        attn(p + ".self_attn"); ln(p + ".self_attn_layer_norm")
        attn(p + ".encoder_attn"); ln(p + ".encoder_attn_layer_norm")
        lin(p + ".fc1", 4 * d, d); lin(p + ".fc2", d, 4 * d); ln(p + ".final_layer_norm")
    ln("model.decoder.layer_norm")
    return out


def synth_tensor(seed: int, idx: int, shape: Tuple[int, ...], kind: str) -> np.ndarray:
    """One tensor, f32 holding fp16-representable values.  Matrices/biases ~ N(0, 0.02^2);
    LayerNorm affine is made non-trivial (1 + 0.1 N, 0.1 N) so that it is actually exercised."""
    rng = np.random.default_rng([seed, idx])
    x = rng.standard_normal(shape, dtype=np.float32)
    if kind in ("w", "b"):
        x *= np.float32(0.02)
    elif kind == "lnw":
        x = np.float32(1.0) + np.float32(0.1) * x
    else:
        x *= np.float32(0.1)
    return x.astype(np.float16).astype(np.float32)


def script_positions(cfg: Config, script: List[int], embed_tokens: np.ndarray, gain: float,
                     prompt_len: int = 3) -> np.ndarray:
    """'Scripted decoder' fixture: a learned-positional-embedding table that steers greedy
    decoding towards `script` (token emitted after position prompt_len-1+i is script[i]) with a
    comfortable margin, so that long free-running decodes have a well-separated argmax while the
    rest of the network still shapes the probabilities.  Pure fixture construction; the tensor is
    an ordinary `model.decoder.embed_positions.weight`."""
    rng = np.random.default_rng(991)
    pos = (0.02 * rng.standard_normal((cfg.max_target_positions, cfg.d_model))).astype(np.float32)
    for i, tok in enumerate(script):
        p = prompt_len - 1 + i
        if p >= cfg.max_target_positions:
            break
        pos[p] += np.float32(gain) * embed_tokens[tok]
    return pos.astype(np.float16).astype(np.float32)


def synth_weights(cfg: Config, seed: int = 0,
                  overrides: Optional[Dict[str, np.ndarray]] = None) -> Iterator[Tuple[str, np.ndarray]]:
    """Yield (name, f32 array) one tensor at a time (distil-large-v3 is 3 GB in f32)."""
    for idx, (name, shape, kind) in enumerate(tensor_specs(cfg)):
        if overrides and name in overrides:
            yield name, np.ascontiguousarray(overrides[name], dtype=np.float32)
        else:
            yield name, synth_tensor(seed, idx, shape, kind)


def synth_tensor_by_name(cfg: Config, name: str, seed: int = 0) -> np.ndarray:
    for idx, (n, shape, kind) in enumerate(tensor_specs(cfg)):
        if n == name:
            return synth_tensor(seed, idx, shape, kind)
    raise KeyError(name)
