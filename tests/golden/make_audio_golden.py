#!/usr/bin/env python3
"""tests/golden/audio_golden.json: audio-DEPENDENT greedy transcripts with the oracle's per-step margins.

For each model (tiny.en, distil-large-v3) and each of CLIPS synthetic clips the oracle (oracle/, the CPU restatement of the
reference path) decodes weights built by tests/common.py:audio_overrides -- the next text token is decided by what the
cross-attention read out of that clip's encoder output -- and this script
  1. measures the reference read-out att_ref (mean over the clips of W_v . mean_s(xa) + b_v of the last decoder layer),
  2. draws the token pairs and vote directions, decodes, and re-draws the direction of every pair whose top-2 relative
     margin (p1 - p2) / p1 is below MIN_MARGIN for any clip, until all steps of all clips clear it,
  3. freezes parameters, expected tokens and margins.
Parity stays "unpinned" (the oracle is a restatement, not the reference binary); what this pins is that token identity
HIP-vs-oracle is asserted where the argmax depends on encoder, cross K/V and decoder numerics.
Run in the build container:  python tests/golden/make_audio_golden.py  (~10 min, dominated by the d = 1280 encoder)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402
from norma_amd import assets_io, config, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

CLIPS = [0, 1, 2, 3]
MIN_MARGIN = 0.03
MAX_ITERATIONS = 30
# 55 distinct pairs, each met twice (110 audio-decided steps + 22 fixed ones): the votes of ALL pairs are added at every
# step, so their number, not the transcript length, is what dilutes the steering
MODELS = {"tiny.en": dict(n_pairs=55, repeats=2, pos_rms=4.0, vote=1.0), "distil-large-v3": dict(n_pairs=55, repeats=2, pos_rms=4.0, vote=1.0),
          # 32 random decoder layers dilute the steering (see tests/test_gpu_configs.py): a stronger positional table
          "large-v3": dict(n_pairs=55, repeats=2, pos_rms=8.0, vote=1.0)}


def build(name, n_pairs, repeats, pos_rms, vote):
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    d = cfg.d_model
    spec = dict(conv_amp=10.0, pos_rms=pos_rms, peak_logit=14.0, gamma=0.0, segment=10, pairs=[], att_ref=[0.0] * d,
                seq=[j for _ in range(repeats) for j in range(n_pairs)])
    rng = np.random.default_rng(2024)
    sup = set(cfg.suppress_tokens)
    while len(spec["pairs"]) < n_pairs:
        a, b = (int(x) for x in rng.integers(300, 40000, size=2))
        if a != b and a not in sup and b not in sup:
            spec["pairs"].append([a, b, len(spec["pairs"])])
    over, kinds = common.audio_overrides(cfg, tk, spec)
    om = common.build_oracle(cfg, tk, overrides=over)
    filt = assets_io.mel_filters(cfg.num_mel_bins)
    t0 = time.time()
    xas = [om.encoder_forward(O.pcm_to_mel(synth.synth_pcm(k), filt)) for k in CLIPS]
    print(name, "encoder x", len(CLIPS), "%.0f s" % (time.time() - t0), flush=True)
    dev_rms, s_rms, eab = common.audio_calibrate(cfg, xas, spec, vote)
    print(name, "att deviation rms %.4f, vote rms %.4f, |Ea-Eb| %.3f -> gamma %.1f" % (dev_rms, s_rms, eab, spec["gamma"]), flush=True)
    next_seed = n_pairs
    for it in range(MAX_ITERATIONS):
        over, kinds = common.audio_overrides(cfg, tk, spec)
        lastp = f"model.decoder.layers.{cfg.decoder_layers - 1}.encoder_attn.out_proj"
        om.set_tensor(lastp + ".weight", over[lastp + ".weight"])
        om.set_tensor(lastp + ".bias", over[lastp + ".bias"])
        res = [om.decode(xa, want_steps=True) for xa in xas]
        pair_steps = [i for i, (k, _) in enumerate(kinds) if k == "pair"]
        worst = {}
        ok = True
        for r in res:
            toks = r["tokens"][3:]
            if len(toks) != len(kinds):
                ok = False
            st = r["steps"][:len(toks)]
            for n, i in enumerate(pair_steps):
                j = spec["seq"][n]
                if i >= len(toks) or toks[i] not in spec["pairs"][j][:2]:
                    worst[j] = -1.0
                    continue
                m = float((st[i, 0] - st[i, 1]) / st[i, 0])
                worst[j] = min(worst.get(j, 9.0), m)
            for i, (k, t) in enumerate(kinds):
                if k != "pair" and (i >= len(toks) or toks[i] != t):
                    ok = False
        bad = [j for j, m in worst.items() if m < MIN_MARGIN]
        print(name, "iteration", it, "transcripts complete" if ok else "transcripts INCOMPLETE", "pairs below margin:", len(bad), flush=True)
        if ok and not bad:
            break
        for j in bad:
            spec["pairs"][j][2] = next_seed
            next_seed += 1
    else:
        raise SystemExit("did not converge")
    out = dict(spec=spec, clips=CLIPS, tokens=[r["tokens"] for r in res], avg_logprob=[r["avg_logprob"] for r in res],
               no_speech_prob=[r["no_speech_prob"] for r in res],
               margins=[[round(float((r["steps"][i, 0] - r["steps"][i, 1]) / r["steps"][i, 0]), 4) for i in range(len(r["tokens"]) - 3)] for r in res],
               p_next=[[round(float(r["steps"][i, 0]), 4) for i in range(len(r["tokens"]) - 3)] for r in res])
    n = len(kinds)
    diff = [sum(x != y for x, y in zip(out["tokens"][a], out["tokens"][b])) for a in range(len(CLIPS)) for b in range(a + 1, len(CLIPS))]
    mm = np.array([m for row in out["margins"] for m in row])
    print(name, "steps", n, "pairwise differing tokens min/median", min(diff), int(np.median(diff)), "margins min %.3f p10 %.3f median %.3f" % (mm.min(), np.percentile(mm, 10), np.median(mm)), flush=True)
    om.close()
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or list(MODELS)
    path = os.path.join(HERE, "audio_golden.json")
    gold = json.load(open(path)) if os.path.exists(path) else {}
    for name in names:
        gold[name] = build(name, **MODELS[name])
        with open(path, "w") as f:
            json.dump(gold, f)
    print("wrote", path)
