"""The oracle reproduces the committed golden fixtures (tests/golden/oracle_golden.json, produced by
tests/golden/make_golden.py).  CPU only.  The fixtures are oracle outputs -- the reference has no
golden vectors for this path (parity unpinned) -- so this test guards the checker against drift."""
import json
import os

import numpy as np
import pytest

import common
from norma_amd import assets_io, config, synth, vocab
from oracle import oracle as O

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.json")))


@pytest.mark.parametrize("key", sorted(GOLDEN["mel"].keys()))
def test_mel_golden(key):
    n_mel, n = (int(x) for x in key.split("/"))
    g = GOLDEN["mel"][key]
    mel = O.pcm_to_mel(synth.synth_pcm(2, n), assets_io.mel_filters(n_mel))[:, :3000]
    assert abs(mel.astype(np.float64).sum() - g["sum"]) < 1e-3
    for c, col in g["cols"].items():
        assert np.abs(mel[:, int(c)] - np.array(col)).max() < 1e-6


def test_mask_sets_golden():
    import hashlib
    for vname, g in GOLDEN["masks"].items():
        tk = vocab.VOCABS[vname]
        sup = sorted(set(vocab.default_suppress_tokens(vname)) | {tk.no_timestamps})
        assert len(sup) == g["n_suppress"]
        assert hashlib.sha256(json.dumps(sup).encode()).hexdigest()[:16] == g["sha"]
        assert [tk.zero_sec, tk.one_sec] == g["first_allowed"] and tk.n_vocab - tk.no_timestamps - 1 == 1501


@pytest.mark.parametrize("idx", [0, 1])  # the two d=128 cases; the larger ones are covered on the GPU side
def test_model_golden(idx):
    m = GOLDEN["models"][idx]
    cfg = config.preset(m["config"])
    tk = common.tokens_for(m["config"])
    over = None
    if m["scripted"]:
        over = common.scripted_overrides(cfg, tk, common.transcript_script(tk, n_segments=5, words_per_segment=8))
    om = common.build_oracle(cfg, tk, seed=m["seed"], overrides=over)
    filt = assets_io.mel_filters(cfg.num_mel_bins)
    for k, c in m["clips"].items():
        xa = om.encoder_forward(O.pcm_to_mel(synth.synth_pcm(int(k)), filt))
        for i, row in c["enc_rows"].items():
            assert np.abs(xa[int(i)][:len(row)] - np.array(row)).max() < 1e-5
        r = om.decode(xa, use_kv_cache=True)
        assert r["tokens"] == c["tokens"]
        assert abs(r["no_speech_prob"] - c["no_speech_prob"]) < 1e-9
        if c["avg_logprob"] is not None:
            assert abs(r["avg_logprob"] - c["avg_logprob"]) < 1e-6
