"""GPU: the decode pool (nh_pool_*, norma_amd/pool.py) -- sequences that join and leave a running decode.

The reference's loop ends per sequence at eot (src/models/whisper/model.rs:317) and it decodes one stream at a time
(src/lib.rs:462-464), so a pool has no counterpart there; what it must reproduce is Model::decode per sequence
(model.rs:279-389): prompt, no-speech probe and exit, rules, length cap, result.  nh_decode_greedy is checked against the
oracle elsewhere (test_gpu_parity, test_gpu_audio, test_gpu_depth); here every clip that went through the pool -- admitted
while other rows were in the middle of their transcripts, at whatever row was free -- must give, bit for bit, what the same
clip gives in a lockstep batch: the step kernels are the same, only the position is read per row."""
import numpy as np
import pytest

import common
from norma_amd import config, pool, synth

pytestmark = pytest.mark.gpu


def _hip():
    from norma_amd import hip
    return hip


def _varlen_weights(cfg, tk, eot_steps, text_steps, n_calib, max_batch, seed=31):
    """weights whose transcripts end where the clip's AUDIO says (tests/common.py:audio_overrides: at the steps in eot_steps
    the vote is between a text token and eot), calibrated on the HIP encoder's own outputs as bench.py's varlen workload does"""
    rng = np.random.default_rng(seed)
    sup = set(cfg.suppress_tokens)

    def tok():
        while True:
            t = int(rng.integers(300, 40000))
            if t not in sup:
                return t
    pairs, seq = [[tok(), tok(), 0]], [0] * text_steps
    for j, e in enumerate(eot_steps, start=1):
        pairs.append([tok(), tk.eot, j]); seq[e] = j
    spec = dict(conv_amp=10.0, pos_rms=4.0, peak_logit=14.0, gamma=0.0, segment=20, pairs=pairs, att_ref=[0.0] * cfg.d_model, seq=seq)
    over0, _ = common.audio_overrides(cfg, tk, spec)
    hm = common.build_hip(cfg, tk, overrides=over0, max_batch=max_batch)
    hm.logmel([synth.synth_pcm(k) for k in range(n_calib)]); hm.encode()
    means = [hm.encoder_output(b, S=1500).mean(0, keepdims=True) for b in range(n_calib)]
    common.audio_calibrate(cfg, means, spec, vote=1.5)
    over, _ = common.audio_overrides(cfg, tk, spec)
    lastp = f"model.decoder.layers.{cfg.decoder_layers - 1}.encoder_attn.out_proj"
    for leaf in (".weight", ".bias"):
        hm.load_tensor(lastp + leaf, over[lastp + leaf].astype(np.float16))
    return hm


def _encode_into(hp, clips):
    def encode(first, n, row0, must=True):
        hp.logmel_array_rows(np.ascontiguousarray(clips[first:first + n]), row0)
        hp.encode_rows(row0, n)
    return encode


def _same(a, b):
    return (a["tokens"] == b["tokens"] and a["avg_logprob"] == b["avg_logprob"] and a["no_speech_prob"] == b["no_speech_prob"]
            and a["no_speech_exit"] == b["no_speech_exit"])


@pytest.mark.parametrize("rows,staging,check_every", [(5, 4, 4), (3, 7, 1), (16, 8, 16)])
def test_clips_through_the_pool_give_bit_for_bit_what_they_give_in_a_lockstep_batch(rows, staging, check_every):
    hip = _hip()
    name, N = "test-d128", 24
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    hm = _varlen_weights(cfg, tk, eot_steps=[2, 5, 9, 14, 22], text_steps=40, n_calib=8, max_batch=N)
    clips = np.stack([synth.synth_pcm(k) for k in range(N)])
    hm.logmel_array(clips); hm.encode()
    want = hm.decode_greedy()
    lengths = sorted({len(r["tokens"]) for r in want})
    assert len(lengths) >= 4, lengths                        # the clips really end at different steps
    hp = hip.HipWhisper(cfg, device=0, max_batch=rows + staging, share_with=hm)
    hp.set_tokens(tk, tk.en, tk.transcribe)
    dp = pool.DecodePool(hp, rows=rows, staging=staging, check_every=check_every)
    got = dp.run(N, _encode_into(hp, clips))
    assert len(got) == N and all(_same(g, w) for g, w in zip(got, want)), [i for i, (g, w) in enumerate(zip(got, want)) if not _same(g, w)]
    assert dp.encodes == -(-N // staging)
    # fewer row-steps than lockstep batches of `rows` clips in arrival order would have run
    need = sum(len(r["tokens"]) - 3 for r in want)
    assert need <= dp.row_steps
    # a second stream through the same pool object state (pool_begin starts over), other clip order
    order = list(reversed(range(N)))
    got2 = pool.DecodePool(hp, rows=rows, staging=staging, check_every=check_every).run(N, _encode_into(hp, clips[order]))
    assert all(_same(g, want[k]) for g, k in zip(got2, order))
    hm.close(); hp.close()


@pytest.mark.parametrize("rows,batch,n_enc", [(6, 4, 2), (16, 8, 1), (5, 7, 3)])
def test_a_pool_fed_by_encoder_contexts_of_the_same_weight_set_gives_the_lockstep_results(rows, batch, n_enc):
    """nh_pool_admit_from: the decoding context never runs an encoder; encoder contexts (nh_create_shared) encode `batch` clips at a
    time on their own streams and threads, their cross K/V are moved into free rows (device-side ordering by events both ways:
    the copy waits for that encoder, the context's next encoder submission waits for the copy)."""
    hip = _hip()
    name, N = "test-d128", 31
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    hm = _varlen_weights(cfg, tk, eot_steps=[2, 5, 9, 14, 22], text_steps=40, n_calib=8, max_batch=N)
    clips = np.stack([synth.synth_pcm(k) for k in range(N)])
    hm.logmel_array(clips); hm.encode()
    want = hm.decode_greedy()
    hp = hip.HipWhisper(cfg, device=0, max_batch=rows + 1, share_with=hm)
    encs = [hip.HipWhisper(cfg, device=0, max_batch=batch, share_with=hm) for _ in range(n_enc)]
    for h in [hp] + encs:
        h.set_tokens(tk, tk.en, tk.transcribe)

    def encode(i, first, n):
        encs[i].logmel_array(np.ascontiguousarray(clips[first:first + n])); encs[i].encode()
    fp = pool.FedDecodePool(hp, encs, rows=rows, batch=batch, check_every=3)
    got = fp.run(N, encode)
    bad = [i for i, (g, w) in enumerate(zip(got, want)) if not _same(g, w)]
    assert not bad, bad
    assert fp.encodes == -(-N // batch)
    # refusals: a context with other weights, a row that is not encoded, an encoder context that runs a pool itself
    other = common.build_hip(cfg, tk, max_batch=2)
    other.logmel_array(clips[:2]); other.encode()
    hp.pool_begin(rows, 0, False)
    with pytest.raises(hip.HipError):
        hp.pool_admit_from(other, 0, 0)
    with pytest.raises(hip.HipError):
        hp.pool_admit_from(encs[0], batch + 3, 0)
    other.close(); hp.close()
    for h in encs:
        h.close()
    hm.close()


def test_pool_no_speech_exit_max_new_tokens_languages_and_refusals():
    hip = _hip()
    name = "test-d128"
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    clips = np.stack([synth.synth_pcm(k) for k in range(6)])
    # (a) model.rs:308-315: position-0 logits put their mass on the no-speech token -> bare prompt, avg_logprob 0
    over = common.scripted_overrides(cfg, tk, [tk.zero_sec, 500, tk.eot])
    emb = over["model.decoder.embed_tokens.weight"]
    pos = over["model.decoder.embed_positions.weight"].copy()
    pos[0] += np.float32(4.0) * emb[tk.no_speech]
    over["model.decoder.embed_positions.weight"] = pos.astype(np.float16).astype(np.float32)
    hm = common.build_hip(cfg, tk, overrides=over, max_batch=6)
    hm.logmel_array(clips); hm.encode()
    want = hm.decode_greedy()
    assert all(w["no_speech_exit"] and w["tokens"] == [tk.sot, tk.en, tk.transcribe] for w in want)
    hp = hip.HipWhisper(cfg, device=0, max_batch=5, share_with=hm)
    hp.set_tokens(tk, tk.en, tk.transcribe)
    got = pool.DecodePool(hp, rows=2, staging=3, check_every=2).run(6, _encode_into(hp, clips))
    assert all(_same(g, w) for g, w in zip(got, want)) and got[0]["avg_logprob"] == 0.0
    hm.close(); hp.close()
    # (b) the max_new_tokens knob and per-clip language tokens (LanguageState::Detect: the prompt's second token per clip)
    script = common.transcript_script(tk, n_segments=2, words_per_segment=4, seed=9)
    over = common.scripted_overrides(cfg, tk, script)
    hm = common.build_hip(cfg, tk, overrides=over, max_batch=6)
    langs = [tk.en, tk.en + 3, tk.en + 1, tk.en, tk.en + 7, tk.en + 2]
    hm.logmel_array(clips); hm.encode(); hm.set_languages(langs)
    want = hm.decode_greedy(max_new_tokens=6)
    assert [w["tokens"][1] for w in want] == langs and all(len(w["tokens"]) <= 3 + 7 and w["tokens"][-1] == tk.eot for w in want)
    hp = hip.HipWhisper(cfg, device=0, max_batch=6, share_with=hm)
    hp.set_tokens(tk, tk.en, tk.transcribe)
    got = pool.DecodePool(hp, rows=4, staging=2, max_new_tokens=6, check_every=3, per_clip_language=True).run(6, _encode_into(hp, clips), langs=langs)
    assert all(_same(g, w) for g, w in zip(got, want))
    # (c) refusals: busy row, unfinished row, lockstep decode while a pool runs, language on a pool begun without languages
    hp.pool_begin(4, 0, False)
    _encode_into(hp, clips)(0, 2, 4, True)
    hp.pool_admit(4, 0)
    with pytest.raises(hip.HipError):
        hp.pool_admit(5, 0)                                   # row 0 is busy
    with pytest.raises(hip.HipError):
        hp.pool_admit(3, 1)                                   # not a staging row
    with pytest.raises(hip.HipError):
        hp.pool_admit(5, 1, lang=tk.en)                       # begun without per-clip languages
    with pytest.raises(hip.HipError):
        hp.pool_collect([0])                                  # has not run yet
    with pytest.raises(hip.HipError):
        hp.decode_greedy()
    flags = hp.pool_step(0)
    assert flags.tolist() == [0, 3, 3, 3]
    while hp.pool_step(8)[0] == 0:
        pass
    r0 = hp.pool_collect([0])[0]
    hm.set_languages(None)
    hm.logmel_array(clips[:1]); hm.encode()
    assert _same(r0, hm.decode_greedy()[0])
    hp.logmel_array(clips[:2]); hp.encode()                   # a fresh batch ends the pool
    assert len(hp.decode_greedy()) == 2
    with pytest.raises(hip.HipError):
        hp.pool_step(1)
    hm.close(); hp.close()


def test_distil_large_v3_pool_of_64_rows_matches_the_lockstep_decode():
    """the headline model at the bench's pool shape (64 decode rows + 32 staging rows; the 64-row step runs the K-phased logits
    kernel and the row-split GEMVs), 96 clips whose transcripts end between 40 and 360 text steps (bench.py's varlen spec)"""
    import sys
    sys.path.insert(0, common.ROOT)
    import bench
    hip = _hip()
    name, N = "distil-large-v3", 96
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    spec_steps, text_steps = bench.VARLEN_EOT_STEPS, bench.VARLEN_TEXT_STEPS
    hm = _varlen_weights(cfg, tk, eot_steps=spec_steps, text_steps=text_steps, n_calib=16, max_batch=96, seed=77)
    clips = np.stack([synth.synth_pcm(k) for k in range(N)])
    want = []
    for g in range(0, N, 32):
        hm.logmel_array(np.ascontiguousarray(clips[g:g + 32])); hm.encode()
        want.extend(hm.decode_greedy())
    lengths = [len(r["tokens"]) - 3 for r in want]
    assert len(set(lengths)) >= 5 and max(lengths) >= 4 * min(lengths), sorted(set(lengths))
    dp = pool.DecodePool(hm, rows=64, staging=32, check_every=16)
    got = dp.run(N, _encode_into(hm, clips))
    bad = [i for i, (g, w) in enumerate(zip(got, want)) if not _same(g, w)]
    assert not bad, bad
    lock = sum(32 * max(lengths[g:g + 32]) for g in range(0, N, 32))
    assert dp.row_steps < lock, (dp.row_steps, lock)
    hm.close()
