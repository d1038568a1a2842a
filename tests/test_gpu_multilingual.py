"""GPU: Model::detect_language (model.rs:194-210) and the multilingual prompt/task handling
(multilingual.rs:383-398, model.rs:285-289) against the oracle, on the V2 (large-v3 style) vocabulary."""
import numpy as np
import pytest

import common
from norma_amd import assets_io, config, host, synth, vocab

pytestmark = pytest.mark.gpu

NAME = "test-d256-mel128"   # V2 vocabulary: 51866 tokens, language tokens 50259..50358


def _lang_tokens(tk):
    return [tk.en + i for i in range(99)]   # Language::iter() order == vocabulary order for the 99 reference languages


def test_detect_language_matches_oracle_and_sets_the_prompt():
    from oracle import oracle as O
    cfg = config.preset(NAME)
    tk = common.tokens_for(NAME)
    langs = _lang_tokens(tk)
    # boost a different language token at position 0 for every weight seed through the positional table
    for seed, want in ((0, tk.en + 7), (1, tk.en + 42)):
        emb = synth.synth_tensor_by_name(cfg, "model.decoder.embed_tokens.weight", seed)
        pos = synth.synth_tensor_by_name(cfg, "model.decoder.embed_positions.weight", seed).copy()
        pos[0] += np.float32(30.0) * emb[want]
        over = {"model.decoder.embed_positions.weight": pos.astype(np.float16).astype(np.float32)}
        om = common.build_oracle(cfg, tk, seed=seed, overrides=over, lang=-1)
        hm = common.build_hip(cfg, tk, seed=seed, overrides=over, max_batch=2, lang=-1)
        filt = assets_io.mel_filters(cfg.num_mel_bins)
        clips = [synth.synth_pcm(0), synth.synth_pcm(1)]
        hm.logmel(clips); hm.encode()
        got, probs = hm.detect_language(langs)
        for b in range(2):
            xa = om.encoder_forward(O.pcm_to_mel(clips[b], filt))
            ref, rp = om.detect_language(xa, langs)
            assert got[b] == ref == want
            assert np.abs(probs[b] - rp).max() <= 2e-3 * rp.max() + 1e-6
        # the detected token is the prompt's language token (model.rs:285-289): [sot, lang, task]
        res = hm.decode_greedy(max_new_tokens=4)
        om.set_language(want)
        r = om.decode(om.encoder_forward(O.pcm_to_mel(clips[0], filt)), max_new_tokens=4)
        assert res[0]["tokens"][:3] == r["tokens"][:3] == [tk.sot, want, tk.transcribe]
        hm.close(); om.close()


def test_host_layer_detects_once_per_transcription_and_clears_on_final_chunk():
    cfg = config.preset(NAME)
    tk = common.tokens_for(NAME)
    script = common.transcript_script(tk, n_segments=2, words_per_segment=3)
    over = common.scripted_overrides(cfg, tk, script)
    emb = over["model.decoder.embed_tokens.weight"]
    pos = over["model.decoder.embed_positions.weight"].copy()
    want = tk.en + 13
    # position 0 decides the language; the script starts at position 2 (prompt_len - 1) as before
    pos[0] = (np.float32(1.2 / 0.02 / max(1.0, 14.0 / (cfg.d_model * 0.02))) * emb[want]).astype(np.float32)
    over["model.decoder.embed_positions.weight"] = pos.astype(np.float16).astype(np.float32)
    d = host.Definition(host.ModelType.LargeV3, host.SelectedDevice.Rocm(0))   # multilingual::ModelType, V2 vocabulary
    model = d.blocking_try_to_model(cfg, tk, -1, tk.transcribe,
                                    ((n, a.astype(np.float16)) for n, a in synth.synth_weights(cfg, 1, over)))
    model.enable_language_detection(_lang_tokens(tk))
    assert model.language_token == -1
    segs = model.transcribe(synth.synth_pcm(0, 320000), final_chunk=False)
    assert model.language_token == want              # LanguageState::set_language_token
    model.transcribe(synth.synth_pcm(1, 160000), final_chunk=True)
    assert model.language_token == -1                # self.lang.clear() on final_chunk (model.rs:153-154)
    assert model.last_result()["n_tokens"] >= 3
    model.close()
