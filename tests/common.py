"""Shared helpers for the parity tests: build the CPU oracle and the HIP model from the same
seeded synthetic weights (fp16-representable values, see norma_amd/synth.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from norma_amd import assets_io, config, synth, vocab  # noqa: E402


def make_config(name, **over):
    cfg = config.preset(name)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def tokens_for(name):
    return vocab.VOCABS[config.preset_vocab(name)]


def oracle_module():
    from oracle import oracle as O
    return O


def build_oracle(cfg, tk, seed=0, overrides=None, lang=None):
    from oracle import oracle as O
    om = O.OracleModel(cfg, tk, tk.en if lang is None else lang, tk.transcribe)
    for n, a in synth.synth_weights(cfg, seed, overrides):
        om.set_tensor(n, a)
    return om


def build_hip(cfg, tk, seed=0, overrides=None, max_batch=1, device=0, lang=None):
    from norma_amd import hip
    hm = hip.HipWhisper(cfg, device=device, max_batch=max_batch)
    hm.load_weights((n, a.astype(np.float16)) for n, a in synth.synth_weights(cfg, seed, overrides))
    hm.set_mel_filters(assets_io.mel_filters(cfg.num_mel_bins))
    hm.set_tokens(tk, tk.en if lang is None else lang, tk.transcribe)
    return hm


def build_together(cfg, tk, seed=0, overrides=None, batches=(1,), device=0, lang=None, with_oracle=True):
    """One pass over the synthetic tensors feeding an oracle and several HIP contexts (large-v3 is 6.2 GB in f32: the
    tensors are generated once).  Returns (oracle or None, [HipWhisper per entry of `batches`])."""
    from norma_amd import hip
    om = None
    if with_oracle:
        from oracle import oracle as O
        om = O.OracleModel(cfg, tk, tk.en if lang is None else lang, tk.transcribe)
    hms = [hip.HipWhisper(cfg, device=device, max_batch=b) for b in batches]
    for n, a in synth.synth_weights(cfg, seed, overrides):
        if om is not None:
            om.set_tensor(n, a)
        a16 = a.astype(np.float16)
        for h in hms:
            h.load_tensor(n, a16)
    for h in hms:
        h.set_mel_filters(assets_io.mel_filters(cfg.num_mel_bins))
        h.set_tokens(tk, tk.en if lang is None else lang, tk.transcribe)
    return om, hms


def scripted_overrides(cfg, tk, script, pos_rms=1.2, seed=0, peak_logit=14.0):
    """Weights that steer the greedy decode along `script` with a wide margin at every step
    (see synth.script_positions): the tied embedding is scaled so that a hidden state aligned with a
    token row reaches a logit of ~peak_logit (a peaked softmax, like a trained model; with the
    0.02-std init the softmax over 51k tokens is flat and text can never beat the summed timestamp
    mass), and embed_positions carries gain * E[script[i]] with the gain chosen so that the row has
    rms `pos_rms`, i.e. dominates the (LayerNorm-bounded) residual branches of the random network."""
    emb = synth.synth_tensor_by_name(cfg, "model.decoder.embed_tokens.weight", seed)
    scale = max(1.0, peak_logit / (cfg.d_model * 0.02))
    emb = (emb * np.float32(scale)).astype(np.float16).astype(np.float32)
    gain = pos_rms / (0.02 * scale)
    return {"model.decoder.embed_tokens.weight": emb,
            "model.decoder.embed_positions.weight": synth.script_positions(cfg, script, emb, gain)}


def transcript_script(tk, n_segments=6, words_per_segment=9, seed=5):
    """A well-formed norma token script: <|t0|> text.. <|t1|> <|t2|> text.. ... text.. eot."""
    rng = np.random.default_rng(seed)
    sup = set(vocab.default_suppress_tokens("EnV1" if tk.n_vocab == 51864 else "V1" if tk.n_vocab == 51865 else "V2"))
    out, ts = [], tk.zero_sec
    for s in range(n_segments):
        out.append(ts)
        for _ in range(words_per_segment):
            t = int(rng.integers(300, 40000))
            while t in sup:
                t += 1
            out.append(t)
        ts += int(rng.integers(60, 200))
        if s + 1 < n_segments:
            out.append(ts)   # closing timestamp
            ts += int(rng.integers(1, 20))
    out.append(tk.eot)
    return out
