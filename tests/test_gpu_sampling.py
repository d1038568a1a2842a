"""GPU: decoding at t > 0 (src/models/whisper/model.rs:340-348) and decode_with_fallback's temperature loop
(:164-191) against the oracle, under the seeded sampling contract of include/norma_hip.h.  The reference itself draws
from an entropy-seeded StdRng, so its draws cannot be reproduced: parity here is HIP path == C restatement, and the
restatement's distribution is checked on the CPU (tests/test_sampling_cpu.py)."""
import numpy as np
import pytest

import common
from norma_amd import assets_io, config, host, synth

pytestmark = pytest.mark.gpu


def test_sampler_kernel_is_bit_exact_against_the_oracle_in_every_rule_state():
    O = common.oracle_module()
    name = "test-d128"
    cfg = common.make_config(name, encoder_layers=0, decoder_layers=0)
    tk = common.tokens_for(name)
    om = common.build_oracle(cfg, tk)
    hm = common.build_hip(cfg, tk)
    rng = np.random.default_rng(8)
    V = cfg.vocab_size
    prompt = [tk.sot, tk.en, tk.transcribe]
    states = [
        (prompt, -1),                                             # first generated token: [<|0.00|>, <|1.00|>] only
        (prompt + [tk.zero_sec + 5], tk.zero_sec + 5),            # after an opening timestamp: text (or later timestamps)
        (prompt + [tk.zero_sec + 5, 700], tk.zero_sec + 5),       # after text: timestamp mass vs best text decides
        (prompt + [tk.zero_sec + 5, 700, tk.zero_sec + 9], tk.zero_sec + 9),
        (prompt + [tk.zero_sec + 5, 700, tk.zero_sec + 9, tk.zero_sec + 9], tk.zero_sec + 9),   # two timestamps in a row
    ]
    n_checked = 0
    for peaked in (False, True):
        for toks, last_ts in states:
            logits = rng.standard_normal(V).astype(np.float32) * (6.0 if peaked else 1.0)
            if peaked:
                logits[tk.zero_sec + 20] += 9.0    # a timestamp-heavy distribution flips the text-state decision
            p = np.exp(logits - logits.max()); p = (p / p.sum()).astype(np.float32)
            # first generated token: logits.broadcast_add(first_token_supress) (model.rs:336-337), else supress_tokens
            q = p + om.mask(3) if last_ts < 0 else om.apply_rules(p, toks, last_ts)
            for t in (0.2, 0.6, 1.0):
                for seed in range(6):
                    want = O.sample_token(q, t, 1000 + seed, 7, len(toks), 3)
                    got = hm.sample_rules(p, toks, last_ts, t, 1000 + seed, clip=7, attempt=3)
                    assert got == want, (peaked, toks, t, seed)
                    assert want >= 0 and np.isfinite(q[want])
                    n_checked += 1
    assert n_checked == 180
    hm.close()


def test_sampled_decode_matches_the_oracle_token_for_token():
    """Batch of 3 (clip ids clip0 + b), random weights: the flat softmax makes the draw nearly uniform over the allowed
    tokens, so a sampled sequence is a sharp test of the whole chain (state update -> rules -> weights -> draw)."""
    O = common.oracle_module()
    name = "test-d128"
    cfg = config.preset(name); tk = common.tokens_for(name)
    om = common.build_oracle(cfg, tk, seed=3)
    hm = common.build_hip(cfg, tk, seed=3, max_batch=3)
    filt = assets_io.mel_filters(cfg.num_mel_bins)
    clips = [synth.synth_pcm(k, 200000 + 40000 * k) for k in range(3)]
    hm.logmel(clips); hm.encode()
    for t, attempt in ((0.2, 1), (1.0, 5)):
        got = hm.decode_sampled(t, seed=77, clip0=10, attempt=attempt, max_new_tokens=40)
        again = hm.decode_sampled(t, seed=77, clip0=10, attempt=attempt, max_new_tokens=40)
        other = hm.decode_sampled(t, seed=78, clip0=10, attempt=attempt, max_new_tokens=40)
        for b in range(3):
            ref = om.decode(om.encoder_forward(O.pcm_to_mel(clips[b], filt)), max_new_tokens=40, temperature=t, seed=77,
                            clip=10 + b, attempt=attempt)
            assert got[b]["tokens"] == ref["tokens"], (t, b)
            assert abs(got[b]["avg_logprob"] - ref["avg_logprob"]) < 2e-2 and ref["avg_logprob"] < -1.0
            assert got[b] == again[b] and got[b]["tokens"] != other[b]["tokens"]
    greedy = hm.decode_greedy(max_new_tokens=40)
    assert greedy[0]["tokens"] != got[0]["tokens"]
    hm.close()


def _pair(over, name="test-d128", seed=0):
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    om = common.build_oracle(cfg, tk, seed=seed, overrides=over)
    d = host.Definition(host.ModelType.TinyEn, host.SelectedDevice.Rocm(0))
    model = d.blocking_try_to_model(cfg, tk, tk.en, tk.transcribe,
                                    ((n, a.astype(np.float16)) for n, a in synth.synth_weights(cfg, seed, over)))
    return cfg, tk, om, model, assets_io.mel_filters(cfg.num_mel_bins)


def test_decode_with_fallback_walks_the_temperatures_and_drops_a_hopeless_slice():
    """A scripted but unconfident model (peak logit 9.5 over a 51 864-way softmax: p(script token) ~ 0.2): the greedy pass follows the script with
    avg_logprob < -1, so decode_with_fallback goes on to t = 0.2 .. 1.0 (model.rs:175-188); those draws are nearly uniform
    over the allowed tokens, none is acceptable, the result is None and transcribe drains the slice without emitting
    anything (:89-92, :189-190).  With the fallback off the t = 0 result is used and flagged."""
    tk = common.tokens_for("test-d128")
    cfg = config.preset("test-d128")
    script = common.transcript_script(tk, n_segments=2, words_per_segment=4)
    cfg, tk, om, model, filt = _pair(common.scripted_overrides(cfg, tk, script, peak_logit=9.5))
    pcm = synth.synth_pcm(0, 160000)
    model.set_temperature_fallback(False, 0)          # opt out of the reference's loop: the t = 0 result, flagged
    segs0 = model.transcribe(pcm, final_chunk=True)
    last0 = model.last_result()
    assert last0["needed_fallback"] and last0["avg_logprob"] < -1.0 and model.buffered_samples == 0
    om.set_sampling(False, 0)
    ref0, _, _ = om.transcribe(pcm, filt, final_chunk=True)
    assert segs0 == ref0 and len(segs0) == 2
    # the default (a fresh model): the loop is on, entropy-seeded like the reference -- whatever it draws, nothing is acceptable
    cfg2, tk2, om2, fresh, _ = _pair(common.scripted_overrides(cfg, tk, script, peak_logit=9.5))
    assert fresh.transcribe(pcm, final_chunk=True) == [] and fresh.last_result()["avg_logprob"] < -1.0
    fresh.close(); om2.close()
    model.set_temperature_fallback(True, 11)
    om.set_sampling(True, 11)
    segs1 = model.transcribe(pcm, final_chunk=True)
    ref1, buf1, info1 = om.transcribe(pcm, filt, final_chunk=True)
    assert segs1 == ref1 == [] and model.buffered_samples == len(buf1) == 0
    last = model.last_result()                       # the last attempt (t = 1.0) is what both sides report
    assert abs(last["avg_logprob"] - info1["avg_logprob"]) < 2e-2 and last["avg_logprob"] < -1.0
    model.close()


def test_fallback_is_not_entered_when_the_greedy_pass_is_acceptable():
    tk = common.tokens_for("test-d128")
    script = common.transcript_script(tk, n_segments=3, words_per_segment=5)
    cfg = config.preset("test-d128")
    cfg, tk, om, model, filt = _pair(common.scripted_overrides(cfg, tk, script))
    model.set_temperature_fallback(True, 5)
    om.set_sampling(True, 5)
    pcm = synth.synth_pcm(2)
    segs = model.transcribe(pcm, final_chunk=True)
    ref, _, info = om.transcribe(pcm, filt, final_chunk=True)
    assert segs == ref and len(segs) == 3
    last = model.last_result()
    assert not last["needed_fallback"] and last["avg_logprob"] > -1.0
    model.close()
