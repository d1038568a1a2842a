"""GPU: several contexts in one process, one host thread each -- the threading model of include/norma_hip.h (the
reference's Model is Send, not Sync: src/models/mod.rs:24, src/lib.rs:377,462-464; multi-GPU = N contexts on N threads).
Two contexts on the same device decode concurrently from two threads; every result must equal the sequential run bit for
bit (per-device launcher caches, per-context graphs and streams, no shared mutable statics)."""
import threading

import numpy as np
import pytest

import common
from norma_amd import config, synth

pytestmark = pytest.mark.gpu


def test_two_contexts_two_threads_match_the_sequential_run():
    name = "test-d256-mel128"
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    script = common.transcript_script(tk, n_segments=4, words_per_segment=8, seed=31)
    over = common.scripted_overrides(cfg, tk, script)
    _, hms = common.build_together(cfg, tk, seed=2, overrides=over, batches=(3, 3), with_oracle=False)
    clips = [[synth.synth_pcm(k) for k in (0, 1, 2)], [synth.synth_pcm(k, 400000) for k in (5, 6, 7)]]

    def run(i, out, rounds):
        res = []
        for _ in range(rounds):
            hms[i].logmel(clips[i]); hms[i].encode()
            res.append((hms[i].decode_greedy(), hms[i].encoder_output(1, S=hms_S[i]).copy()))
        out[i] = res
    hms_S = [1500, 1500]
    seq = [None, None]
    for i in (0, 1):
        run(i, seq, 1)
    par = [None, None]
    ths = [threading.Thread(target=run, args=(i, par, 4)) for i in (0, 1)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in (0, 1):
        assert par[i] is not None and len(par[i]) == 4
        for res, enc in par[i]:
            assert np.array_equal(enc, seq[i][0][1])
            for a, b in zip(res, seq[i][0][0]):
                assert a["tokens"] == b["tokens"] and a["avg_logprob"] == b["avg_logprob"] and a["no_speech_prob"] == b["no_speech_prob"]
                assert a["tokens"] == [tk.sot, tk.en, tk.transcribe] + script
    # a failed nh_create on one thread must not clobber the other thread's error slot
    from norma_amd import hip
    errs = {}

    def bad(i):
        try:
            hip.HipWhisper(cfg, device=99 + i, max_batch=1)
        except hip.HipError as e:
            errs[i] = str(e)
    ths = [threading.Thread(target=bad, args=(i,)) for i in (0, 1)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert "ordinal 99" in errs[0] and "ordinal 100" in errs[1]
    for h in hms:
        h.close()


def test_contexts_that_share_one_weight_set_match_private_weights_and_outlive_their_parent():
    """nh_create_shared: three contexts over ONE copy of the weights (what bench.py keeps in flight per GPU).  Loaded once
    through the parent; the children race into their first decode from three threads (the lazy tile-major repack of the
    decoder weights happens once, under the model's lock); every result equals a context with private weights bit for bit;
    the parent is destroyed FIRST and the children go on (the weights are reference counted)."""
    from norma_amd import assets_io, hip
    name = "test-d256-mel128"
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    script = common.transcript_script(tk, n_segments=3, words_per_segment=7, seed=13)
    over = common.scripted_overrides(cfg, tk, script)
    private = common.build_hip(cfg, tk, seed=2, overrides=over, max_batch=3)
    parent = common.build_hip(cfg, tk, seed=2, overrides=over, max_batch=3)
    kids = [hip.HipWhisper(cfg, device=0, max_batch=b, share_with=parent) for b in (3, 2)]
    for k in kids:
        k.set_tokens(tk, tk.en, tk.transcribe)        # tokens are per context; the mel filters came with the weights
        assert k.L.nh_missing_tensors(k._h) == 0
    clips = [synth.synth_pcm(k) for k in (0, 1, 2)]
    private.logmel(clips); private.encode()
    want = private.decode_greedy()
    want_enc = private.encoder_output(2).copy()
    out = {}

    def run(i, h, n):
        res = []
        for _ in range(3):
            h.logmel(clips[:n]); h.encode()
            res.append((h.decode_greedy(), h.encoder_output(n - 1).copy()))
        out[i] = res
    ths = [threading.Thread(target=run, args=(i, h, n)) for i, (h, n) in enumerate(((parent, 3), (kids[0], 3), (kids[1], 2)))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i, n in ((0, 3), (1, 3), (2, 2)):
        for res, enc in out[i]:
            assert np.array_equal(enc, want_enc if n == 3 else private.encoder_output(1))
            for a, b in zip(res, want):
                assert a["tokens"] == b["tokens"] == [tk.sot, tk.en, tk.transcribe] + script
                assert a["avg_logprob"] == b["avg_logprob"] and a["no_speech_prob"] == b["no_speech_prob"]
    parent.close()                                    # the family's weights must survive their first owner
    kids[0].logmel(clips); kids[0].encode()
    again = kids[0].decode_greedy()
    assert [r["tokens"] for r in again] == [r["tokens"] for r in want]
    assert [r["avg_logprob"] for r in again] == [r["avg_logprob"] for r in want]
    # a tensor re-loaded through one member is seen by the others (and the tile-major repack is rebuilt)
    emb = over["model.decoder.embed_positions.weight"].copy()
    emb[5] += 0.25 * over["model.decoder.embed_tokens.weight"][1234]
    kids[1].load_tensor("model.decoder.embed_positions.weight", emb.astype(np.float16))
    private.load_tensor("model.decoder.embed_positions.weight", emb.astype(np.float16))
    kids[0].logmel(clips); kids[0].encode()
    private.logmel(clips); private.encode()
    a, b = kids[0].decode_greedy(), private.decode_greedy()
    assert [r["tokens"] for r in a] == [r["tokens"] for r in b] and [r["avg_logprob"] for r in a] == [r["avg_logprob"] for r in b]
    for h in kids + [private]:
        h.close()
