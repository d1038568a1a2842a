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


# --------------------------------------------------------------------------------------------------------------------
# audio-dependent transcripts: weights whose greedy tokens are decided by the AUDIO (encoder -> cross K/V -> cross-attention
# -> logits), with trained-model-like margins.  Built by tests/golden/make_audio_golden.py, replayed by
# tests/test_gpu_audio.py from the parameters frozen in tests/golden/audio_golden.json.
#
# With the plain seed-0 weights the conv stem is so weak next to the sinusoidal positions that two different clips give
# encoder outputs 1 % apart, and the scripted fixtures above carry the argmax in embed_positions alone.  Here
#   * conv1 / conv2 weights are scaled by `conv_amp`, so that encoder outputs of different clips differ by ~30 % rms;
#   * at a "pair" step the positional table steers towards TWO tokens (a, b) equally; fixed steps (timestamps, eot) keep
#     a single target, so the transcript keeps norma's timestamp grammar;
#   * the cross-attention out_proj of the LAST decoder layer gets gamma * sum_p e_p (x) R_p on top of its random
#     weights, e_p = unit(E[a_p] - E[b_p]), R_p a seeded random unit vector, and its bias removes the response to a reference
#     attention output (the mean over the fixture's clips): what is left votes for a or b in proportion to
#     R_p . (att(clip) - att_ref), i.e. by what the cross-attention read out of THIS clip's encoder output.
# --------------------------------------------------------------------------------------------------------------------
def audio_step_layout(tk, n_pairs, segment=10):
    """[(kind, token)]: 'ts' fixed timestamp, 'pair' audio-decided text, 'eot'."""
    kinds, ts, n = [("ts", tk.zero_sec)], tk.zero_sec, 0
    while n < n_pairs:
        for _ in range(min(segment, n_pairs - n)):
            kinds.append(("pair", None)); n += 1
        if n < n_pairs:
            ts += 60; kinds.append(("ts", ts)); ts += 4; kinds.append(("ts", ts))
    kinds.append(("eot", tk.eot))
    return kinds


def audio_pair_direction(seed, d):
    r = np.random.default_rng([77, int(seed)]).standard_normal(d)
    return (r / np.linalg.norm(r)).astype(np.float32)


def audio_overrides(cfg, tk, spec, seed=0):
    """spec: dict(conv_amp, pos_rms, peak_logit, gamma, segment, pairs=[[a, b, r_seed], ...] (distinct pairs, one vote
    direction each), seq=[pair index of every pair step] (a pair may recur), att_ref=[d floats])."""
    d = cfg.d_model
    kinds = audio_step_layout(tk, len(spec["seq"]), spec["segment"])
    emb = synth.synth_tensor_by_name(cfg, "model.decoder.embed_tokens.weight", seed)
    scale = max(1.0, spec["peak_logit"] / (d * 0.02))
    emb = (emb * np.float32(scale)).astype(np.float16).astype(np.float32)
    gain = spec["pos_rms"] / (0.02 * scale)
    pos = (0.02 * np.random.default_rng(991).standard_normal((cfg.max_target_positions, d))).astype(np.float32)
    j = 0
    for i, (k, t) in enumerate(kinds):
        if k == "pair":
            a, b, _ = spec["pairs"][spec["seq"][j]]; j += 1
            pos[2 + i] += np.float32(gain / np.sqrt(2.0)) * (emb[a] + emb[b])
        else:
            pos[2 + i] += np.float32(gain) * emb[t]
    over = {"model.decoder.embed_tokens.weight": emb,
            "model.decoder.embed_positions.weight": pos.astype(np.float16).astype(np.float32)}
    for n in ("model.encoder.conv1.weight", "model.encoder.conv2.weight"):
        over[n] = (synth.synth_tensor_by_name(cfg, n, seed) * np.float32(spec["conv_amp"])).astype(np.float16).astype(np.float32)
    last = f"model.decoder.layers.{cfg.decoder_layers - 1}.encoder_attn.out_proj"
    wo = synth.synth_tensor_by_name(cfg, last + ".weight", seed).astype(np.float64)
    bo = synth.synth_tensor_by_name(cfg, last + ".bias", seed).astype(np.float64)
    att_ref = np.asarray(spec["att_ref"], dtype=np.float64)
    for a, b, rs in spec["pairs"]:
        e = (emb[a] - emb[b]).astype(np.float64)
        e /= np.linalg.norm(e)
        wo += spec["gamma"] * np.outer(e, audio_pair_direction(rs, d).astype(np.float64))
    wo16 = wo.astype(np.float16).astype(np.float32)
    # the bias cancels the response of the ROUNDED vote rows to the reference read-out
    vote = wo16.astype(np.float64) - synth.synth_tensor_by_name(cfg, last + ".weight", seed).astype(np.float64)
    over[last + ".weight"] = wo16
    over[last + ".bias"] = (bo - vote @ att_ref).astype(np.float16).astype(np.float32)
    return over, kinds


def audio_calibrate(cfg, xas, spec, vote, seed=0):
    """Fill spec['att_ref'] and spec['gamma'] from the encoder outputs `xas` of the fixture's clips: att_ref = mean over the
    clips of the last decoder layer's (near-uniform) cross-attention read-out W_v . mean_s(xa) + b_v; gamma such that a
    typical vote moves logit(a) - logit(b) by ~`vote`."""
    d = cfg.d_model
    last = f"model.decoder.layers.{cfg.decoder_layers - 1}.encoder_attn"
    wv = synth.synth_tensor_by_name(cfg, last + ".v_proj.weight", seed)
    bv = synth.synth_tensor_by_name(cfg, last + ".v_proj.bias", seed)
    atts = np.stack([wv @ xa.mean(0) + bv for xa in xas])
    att_ref = atts.mean(0)
    dev = atts - att_ref
    emb = synth.synth_tensor_by_name(cfg, "model.decoder.embed_tokens.weight", seed) * np.float32(max(1.0, spec["peak_logit"] / (d * 0.02)))
    eab = float(np.mean([np.linalg.norm(emb[a] - emb[b]) for a, b, _ in spec["pairs"][:20]]))
    s_rms = float(np.sqrt(np.mean([(dev @ audio_pair_direction(i, d)) ** 2 for i in range(50)])))
    spec["att_ref"] = [float(v) for v in att_ref.astype(np.float32)]
    spec["gamma"] = float(vote * spec["pos_rms"] / (eab * s_rms))
    return float(np.sqrt((dev ** 2).mean())), s_rms, eab
