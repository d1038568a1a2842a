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
