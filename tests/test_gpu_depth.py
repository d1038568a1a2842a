"""GPU: decoder numerics at FULL decoder depth, with stated tolerances.

`north_star` asks for "logits within a stated fp tolerance"; the reference's decoder is Type::decoder_forward +
Type::decoder_final_linear (src/models/whisper/model.rs:466-483).  tests/test_gpu_parity.py bounds hidden / logit error at
<= 4 decoder layers; this file does it where the error is largest:

  large-v3 (32 + 32 layers)      a teacher-forced 16-token prefix through nh_decoder_forward / nh_final_linear against the
                                 oracle, per position, plus a DEPTH PROFILE (the first n = 1, 2, 4, 8, 16, 32 decoder blocks,
                                 NH_OPT_DECODER_LAYER_LIMIT, against oracles built with n decoder layers from the same tensors),
                                 so that a per-layer error that grows faster than fp16 rounding can be seen and located;
  the `pos_rms = 1.2` fixture    the weights of round 2's failed run (gpurun_out/r02_pytest1.log: "At index 5 diff: 50687 != 5404").
                                 That log is the second half of the chained assertion `got == ref == prompt + script`: HIP and the
                                 oracle BOTH emit timestamp 50687 there and both leave the script (5404) -- measured here: the
                                 two token lists are identical, and at that step `sum p[timestamps] >= max p[text]`
                                 (model.rs:263-272) holds by 0.0287 vs 0.0043 on either side, so the rule forces a timestamp.
                                 The fixture was wrong (32 random decoder layers dilute a pos_rms = 1.2 steer), not the kernel.
                                 The test keeps the evidence: both sides of the rule at every position for the oracle and for the
                                 HIP logits, the in-loop tokens (fused logit step, kind == 3 branch) against the rule evaluated on
                                 the HIP logits, and token identity wherever the oracle's margin exceeds what the measured
                                 logit error can move;
  distil-large-v3 (32 + 2)       the same hidden / logit bars at its full depth.

The numbers of the last run are written to gpurun_out/r03_depth_report.json (copied to profiles/ by hand) and quoted in
DESIGN.md's parity table.
"""
import json
import os

import numpy as np
import pytest

import common
from norma_amd import assets_io, config, hip, synth

pytestmark = pytest.mark.gpu

T_PREFIX = 16
# Stated tolerances (hidden = output of the final LayerNorm, O(1) per element; logits relative to their own spread).
# Measured values are in DESIGN.md section 2; the bars leave ~2x headroom over the worst position seen.
# measured (r03, profiles/r03_depth_report.json): hidden max 1.3e-3 (1 block) ... 1.8e-3 (32 blocks), rms 1.8e-4 ... 3.7e-4;
# logits <= 2.2e-3 sigma at 32 blocks; distil-large-v3 1.9e-3 / 1.9e-3 sigma
HID_BAR = {1: 3e-3, 2: 3e-3, 4: 3e-3, 8: 3.5e-3, 16: 4e-3, 32: 4e-3}
LOGIT_BAR_SIGMA = 5e-3        # max |logit error| / std(logits of that position), 32 decoder layers
DISTIL_HID_BAR, DISTIL_LOGIT_BAR_SIGMA = 4e-3, 4e-3


def _report(name, payload):
    out = os.path.join(common.ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "r03_depth_report.json")
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data[name] = payload
        with open(path, "w") as f:
            json.dump(data, f, indent=1)
    except OSError:
        pass
    print(name, json.dumps(payload))


def _rule_sides(logits, tk, suppress):
    """model.rs:263-270 on one row of logits: (sum of timestamp probabilities, best allowed text probability, its index)."""
    z = logits.astype(np.float64)
    p = np.exp(z - z.max())
    p /= p.sum()
    nt = tk.no_timestamps
    sum_ts = float(p[nt + 1:].sum())
    text = p[:nt].copy()
    sup = np.asarray([s for s in suppress if s < nt], dtype=np.int64)
    text[sup] = -np.inf
    return sum_ts, float(text.max()), int(text.argmax())


def _feed(cfg_full, over, om_full, partial, hms):
    """One pass over the synthetic tensors: the full oracle, the decoder-only oracles of `partial` {n: OracleModel} (they
    take the embedding, the positions, the first n decoder blocks and the final LayerNorm) and the HIP contexts."""
    for name, arr in synth.synth_weights(cfg_full, 0, over):
        if om_full is not None:
            om_full.set_tensor(name, arr)
        if name.startswith("model.decoder."):
            for n, om in partial.items():
                if name.startswith("model.decoder.layers.") and int(name.split(".")[3]) >= n:
                    continue
                om.set_tensor(name, arr)
        a16 = arr.astype(np.float16)
        for h in hms:
            h.load_tensor(name, a16)


def test_large_v3_all_32_decoder_layers_hidden_and_logits_and_the_r02_divergence():
    from oracle import oracle as O
    import test_gpu_configs as tc
    name = "large-v3"
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    assert (cfg.encoder_layers, cfg.decoder_layers) == (32, 32)
    script = common.transcript_script(tk, n_segments=4, words_per_segment=7, seed=11)
    want = tk.en + 23
    over = tc._multilingual_overrides(cfg, tk, script, want, pos_rms=1.2)      # round 2's ORIGINAL fixture
    depths = [1, 2, 4, 8, 16]
    om = O.OracleModel(cfg, tk, -1, tk.transcribe)
    partial = {}
    for n in depths:
        c = config.preset(name)
        c.encoder_layers, c.decoder_layers = 0, n
        partial[n] = O.OracleModel(c, tk, -1, tk.transcribe)
    hm = hip.HipWhisper(cfg, device=0, max_batch=1)
    _feed(cfg, over, om, partial, [hm])
    hm.set_mel_filters(assets_io.mel_filters(cfg.num_mel_bins))
    hm.set_tokens(tk, -1, tk.transcribe)

    clip = synth.synth_pcm(9)
    filt = assets_io.mel_filters(cfg.num_mel_bins)
    xa = om.encoder_forward(O.pcm_to_mel(clip, filt))
    hm.logmel([clip]); hm.encode()
    enc_err = float(np.abs(hm.encoder_output(0) - xa).max())
    assert enc_err <= 4e-3, enc_err

    om.set_language(want)
    ref = om.decode(xa)                                     # the oracle's own greedy transcript under this fixture
    hm.set_languages([want])
    got = hm.decode_greedy()[0]                             # the HIP path's (fused logit step, hipGraph replay)
    prefix = np.array([ref["tokens"][:T_PREFIX]], dtype=np.int32)
    T = prefix.shape[1]
    assert T == T_PREFIX and ref["tokens"][:3] == [tk.sot, want, tk.transcribe]

    # ---- depth profile: the first n decoder blocks ----
    profile = []
    for n in depths + [32]:
        hm.set_option(hip.NH_OPT_DECODER_LAYER_LIMIT, 0 if n == 32 else n)
        hid = hm.decoder_forward(prefix)[0]
        ref_h = (om if n == 32 else partial[n]).decoder_forward(prefix[0], xa, True)
        err = np.abs(hid - ref_h)
        profile.append(dict(layers=n, max=float(err.max()), rms=float(np.sqrt((err ** 2).mean())),
                            per_pos_max=[float(v) for v in err.max(1)]))
        if n == 32:
            hid32, ref32 = hid, ref_h
    hm.set_option(hip.NH_OPT_DECODER_LAYER_LIMIT, 0)
    for om_n in partial.values():
        om_n.close()
    checks = []          # evaluated after the report is written, so a failing run still leaves its numbers behind
    for row in profile:
        checks.append((row["max"] <= HID_BAR[row["layers"]], ("hidden error over the bar", row["layers"], row["max"])))
    # fp16 operand rounding adds up like a random walk over the blocks (3 residual updates each): the rms error may grow
    # with depth, but not faster than linearly in the number of blocks
    r1, r32 = profile[0]["rms"], profile[-1]["rms"]
    checks.append((r32 <= 32 * r1, ("rms error grows faster than linearly in depth", r1, r32)))

    # ---- logits at full depth, and both sides of `sum p[ts] >= max p[text]` ----
    got_l = hm.final_linear(hid32)
    ref_l = om.final_linear(ref32)
    sig = ref_l.std(1)
    lerr = np.abs(got_l - ref_l).max(1)
    checks.append((bool((lerr <= LOGIT_BAR_SIGMA * sig).all()), ("logit error over the bar", (lerr / sig).tolist())))
    sup = cfg.suppress_tokens
    sides = []
    for p in range(T):
        o_ts, o_tx, o_i = _rule_sides(ref_l[p], tk, sup)
        h_ts, h_tx, h_i = _rule_sides(got_l[p], tk, sup)
        sides.append(dict(pos=p, oracle_sum_ts=o_ts, oracle_max_text=o_tx, hip_sum_ts=h_ts, hip_max_text=h_tx,
                          oracle_text=o_i, hip_text=h_i, logit_err_over_sigma=float(lerr[p] / sig[p])))

    # ---- gpurun_out/r02_pytest1.log: HIP vs oracle under the pos_rms = 1.2 fixture ----
    first = next((i for i, (a, b) in enumerate(zip(got["tokens"], ref["tokens"])) if a != b), None)
    if first is None and len(got["tokens"]) != len(ref["tokens"]):
        first = min(len(got["tokens"]), len(ref["tokens"]))
    prompt_script = [tk.sot, want, tk.transcribe] + script
    off_script = next((i for i, (a, b) in enumerate(zip(ref["tokens"], prompt_script)) if a != b), None)
    div = dict(hip_tokens=got["tokens"][:T], oracle_tokens=ref["tokens"][:T], script_tokens=prompt_script[:T],
               first_hip_vs_oracle_divergence=first, oracle_leaves_the_script_at=off_script,
               n_tokens=dict(hip=len(got["tokens"]), oracle=len(ref["tokens"])))
    # every text-vs-timestamp decision inside the prefix (token p + 1 is decided from the logits of position p when
    # token p is text): the in-loop choice must be the rule evaluated on the HIP logits, and must equal the oracle's
    # wherever the oracle's log-margin exceeds what the measured logit error can move (one logit up, the mass down)
    decisions = []
    for p in range(3, T - 1):
        if not ref["tokens"][p] < tk.no_timestamps or got["tokens"][:p + 1] != ref["tokens"][:p + 1]:
            continue
        s = sides[p]
        margin = abs(float(np.log(s["oracle_max_text"] / s["oracle_sum_ts"])))
        reach = 2.0 * float(lerr[p])
        hip_says_ts = s["hip_sum_ts"] >= s["hip_max_text"]
        decisions.append(dict(pos=p, oracle_log_margin=margin, logit_error_reach=reach, hip_rule_says_timestamp=bool(hip_says_ts),
                              hip_token=got["tokens"][p + 1], oracle_token=ref["tokens"][p + 1]))
        checks.append(((got["tokens"][p + 1] > tk.no_timestamps) == hip_says_ts, ("in-loop decision differs from the rule on the HIP logits", s)))
        if not hip_says_ts:
            checks.append((got["tokens"][p + 1] == s["hip_text"], ("in-loop text token is not the HIP logits' best text", s)))
        if margin > 10.0 * reach:
            checks.append((got["tokens"][p + 1] == ref["tokens"][p + 1], ("tokens differ at a decision with a clear oracle margin", p, margin, reach)))
    div["decisions"] = decisions
    checks.append((len(decisions) >= 4, ("too few text-vs-timestamp decisions in the prefix", len(decisions))))
    checks.append((off_script is not None and off_script == 5, ("the oracle was expected to leave the script at token 5 (r02 log)", off_script)))
    checks.append((first is None or first >= T, ("HIP and the oracle diverge inside the prefix", first)))
    _report("large-v3", dict(encoder_max_err=enc_err, depth_profile=profile, logit_err_over_sigma=[float(v) for v in lerr / sig],
                             logit_sigma=[float(v) for v in sig], rule_sides=sides, divergence=div))
    for ok, msg in checks:
        assert ok, msg
    hm.close(); om.close()


def test_distil_large_v3_full_depth_hidden_and_logits():
    from oracle import oracle as O
    name = "distil-large-v3"
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    script = common.transcript_script(tk, n_segments=3, words_per_segment=6, seed=4)
    over = common.scripted_overrides(cfg, tk, script)
    om, (hm,) = common.build_together(cfg, tk, overrides=over, batches=(2,))
    filt = assets_io.mel_filters(cfg.num_mel_bins)
    clips = [synth.synth_pcm(3), synth.synth_pcm(4)]
    hm.logmel(clips); hm.encode()
    toks = np.array([[tk.sot, tk.en, tk.transcribe] + script[:T_PREFIX - 3]] * 2, dtype=np.int32)
    hid = hm.decoder_forward(toks)
    rows = []
    for b in range(2):
        xa = om.encoder_forward(O.pcm_to_mel(clips[b], filt))
        ref_h = om.decoder_forward(toks[b], xa, True)
        ref_l = om.final_linear(ref_h)
        got_l = hm.final_linear(hid[b])
        sig = ref_l.std(1)
        rows.append(dict(clip=b, enc_err=float(np.abs(hm.encoder_output(b) - xa).max()),
                         hid_max=float(np.abs(hid[b] - ref_h).max()),
                         logit_err_over_sigma=float((np.abs(got_l - ref_l).max(1) / sig).max()),
                         same_argmax=bool((got_l.argmax(1) == ref_l.argmax(1)).all())))
    _report("distil-large-v3", dict(rows=rows))
    for r in rows:
        assert r["enc_err"] <= 4e-3, r
        assert r["hid_max"] <= DISTIL_HID_BAR, r
        assert r["logit_err_over_sigma"] <= DISTIL_LOGIT_BAR_SIGMA, r
        assert r["same_argmax"], r
    hm.close(); om.close()
