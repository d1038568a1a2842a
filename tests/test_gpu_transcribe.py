"""GPU: the C++ host layer's Model::transcribe (buffering, 30 s windows, segment cutting, seek by
timestamps, final_chunk) against the oracle's restatement of src/models/whisper/model.rs:55-159 on
scripted fixtures (peaked decodes, so the emitted segments are unambiguous)."""
import numpy as np
import pytest

import common
from norma_amd import assets_io, config, host, synth

pytestmark = pytest.mark.gpu


def _pair(script, name="test-d128"):
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    over = common.scripted_overrides(cfg, tk, script)
    om = common.build_oracle(cfg, tk, overrides=over)
    d = host.Definition(host.ModelType.TinyEn, host.SelectedDevice.Rocm(0))
    model = d.blocking_try_to_model(cfg, tk, tk.en, tk.transcribe,
                                    ((n, a.astype(np.float16)) for n, a in synth.synth_weights(cfg, 0, over)))
    return cfg, tk, om, model, assets_io.mel_filters(cfg.num_mel_bins)


def test_per_chunk_semantics_final_chunk_drains_and_emits_all_segments():
    tk = common.tokens_for("test-d128")
    script = common.transcript_script(tk, n_segments=4, words_per_segment=5)
    cfg, tk, om, model, filt = _pair(script)
    pcm = synth.synth_pcm(0)
    segs = model.transcribe(pcm, final_chunk=True)
    ref, buf, info = om.transcribe(pcm, filt, final_chunk=True)
    assert segs == ref and len(segs) == 4
    assert model.buffered_samples == len(buf) == 0
    last = model.last_result()
    assert abs(last["avg_logprob"] - info["avg_logprob"]) < 5e-3 and not last["needed_fallback"]
    model.close()


def test_streaming_seek_keeps_the_unfinished_segment_for_the_next_call():
    """Not final, 25 s chunk (the reference's default max_chunk_len), last segment opens at a non-zero
    timestamp: the buffer is drained up to that timestamp (320 samples per 0.02 s unit, model.rs:125-127) and
    the segment's text is withheld until more audio arrives."""
    tk = common.tokens_for("test-d128")
    script = common.transcript_script(tk, n_segments=3, words_per_segment=4)
    cfg, tk, om, model, filt = _pair(script)
    pcm = synth.synth_pcm(1, 400000)
    segs = model.transcribe(pcm, final_chunk=False)
    ref, buf, _ = om.transcribe(pcm, filt, final_chunk=False)
    assert segs == ref
    assert model.buffered_samples == len(buf) and 0 < len(buf) < 400000
    # the second call appends to the carried-over buffer (model.rs:60-64) and, being final, drains everything
    more = synth.synth_pcm(2, 200000)
    segs2 = model.transcribe(more, final_chunk=True)
    ref2, buf2, _ = om.transcribe(more, filt, final_chunk=True, buf=buf)
    assert segs2 == ref2 and model.buffered_samples == len(buf2) == 0
    model.close()


def test_long_buffer_is_cut_into_30s_windows():
    tk = common.tokens_for("test-d128")
    script = [tk.zero_sec] + [700, 800, 900] + [tk.eot]   # one segment starting at 0.00 -> the whole window is consumed
    cfg, tk, om, model, filt = _pair(script)
    pcm = np.concatenate([synth.synth_pcm(3), synth.synth_pcm(4, 240000)])   # 45 s
    segs = model.transcribe(pcm, final_chunk=True)
    ref, buf, info = om.transcribe(pcm, filt, final_chunk=True)
    assert info["n_slices"] == 2 and segs == ref == [[700, 800, 900]] * 2
    assert model.buffered_samples == 0
    model.close()


def test_no_speech_result_drains_instead_of_spinning_forever():
    """Hazard H1: the reference would loop forever here (model.rs:308-315 + :95-98 + :100); both sides drain."""
    tk = common.tokens_for("test-d128")
    cfg = config.preset("test-d128")
    over = common.scripted_overrides(cfg, tk, [tk.zero_sec, 500, tk.eot])
    pos = over["model.decoder.embed_positions.weight"].copy()
    pos[0] += np.float32(4.0) * over["model.decoder.embed_tokens.weight"][tk.no_speech]
    over["model.decoder.embed_positions.weight"] = pos.astype(np.float16).astype(np.float32)
    om = common.build_oracle(cfg, tk, overrides=over)
    d = host.Definition(host.ModelType.TinyEn, host.SelectedDevice.Rocm(0))
    model = d.blocking_try_to_model(cfg, tk, tk.en, tk.transcribe,
                                    ((n, a.astype(np.float16)) for n, a in synth.synth_weights(cfg, 0, over)))
    pcm = synth.synth_pcm(0, 320000)
    segs = model.transcribe(pcm, final_chunk=False)
    ref, buf, info = om.transcribe(pcm, assets_io.mel_filters(80), final_chunk=False)
    assert segs == ref == [] and model.buffered_samples == len(buf) == 0
    assert model.last_result()["no_speech_prob"] > 0.6
    model.close()


def _bytes_to_unicode():
    bs = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


def _write_checkpoint_dir(path, cfg, tk, weights):
    """config.json + tokenizer.json (byte-level BPE layout with Whisper's added tokens) + model.safetensors, the three
    files the reference pulls through hf-hub (monolingual.rs:323-345); contents are synthetic."""
    import json
    import os
    from safetensors.numpy import save_file
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg.to_dict(), f)
    b2u = _bytes_to_unicode()
    words = {}
    for i in range(tk.eot):                       # ordinary vocabulary: " w<i>" in GPT-2 byte-level spelling
        words[i] = "".join(b2u[b] for b in f" w{i}".encode())
    vocab = {v: k for k, v in words.items()}
    added = [dict(id=tk.eot, content="<|endoftext|>", special=True), dict(id=tk.sot, content="<|startoftranscript|>", special=True),
             dict(id=tk.en, content="<|en|>", special=True), dict(id=tk.translate, content="<|translate|>", special=True),
             dict(id=tk.transcribe, content="<|transcribe|>", special=True), dict(id=tk.no_speech, content="<|nocaptions|>", special=True),
             dict(id=tk.no_timestamps, content="<|notimestamps|>", special=True)]
    for i in range(1501):
        added.append(dict(id=tk.zero_sec + i, content=f"<|{i * 0.02:.2f}|>", special=True))
    with open(os.path.join(path, "tokenizer.json"), "w") as f:
        json.dump(dict(version="1.0", added_tokens=added, model=dict(type="BPE", vocab=vocab, merges=[])), f)
    tensors = {n: a.astype(np.float16) for n, a in weights}
    tensors["proj_out.weight"] = tensors["model.decoder.embed_tokens.weight"]        # present in HF files, not read by candle
    tensors["model.encoder.embed_positions.weight"] = np.zeros((1500, cfg.d_model), np.float16)
    save_file(tensors, os.path.join(path, "model.safetensors"))
    return words


def test_local_checkpoint_directory_loads_and_yields_text(tmp_path):
    """§8(f)-1: safetensors / config.json / tokenizer.json read by the C++ host layer; special-token ids resolved by
    name as the reference does (mod.rs:86-90), text through the byte-level BPE decoder (model.rs:147)."""
    name = "test-d128"
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    script = common.transcript_script(tk, n_segments=3, words_per_segment=4)
    over = common.scripted_overrides(cfg, tk, script)
    words = _write_checkpoint_dir(str(tmp_path), cfg, tk, synth.synth_weights(cfg, 0, over))
    d = host.Definition(host.ModelType.TinyEn, host.SelectedDevice.Rocm(0))
    model = d.blocking_try_to_model_from_dir(str(tmp_path))
    pcm = synth.synth_pcm(0)
    segs = model.transcribe(pcm, final_chunk=True)
    # same tokens as the model built by injecting the tensors directly
    ref_model = d.blocking_try_to_model(cfg, tk, tk.en, tk.transcribe,
                                        ((n, a.astype(np.float16)) for n, a in synth.synth_weights(cfg, 0, over)))
    assert segs == ref_model.transcribe(pcm, final_chunk=True) and len(segs) == 3
    expect = "".join("".join(f" w{t}" for t in s) for s in segs)
    assert model.last_text() == expect
    model.close(); ref_model.close()
    # a checkpoint whose tokenizer lacks a required token fails with the reference's TokenId error
    import json, os
    tj = json.load(open(os.path.join(tmp_path, "tokenizer.json")))
    tj["added_tokens"] = [t for t in tj["added_tokens"] if t["content"] != "<|notimestamps|>"]
    json.dump(tj, open(os.path.join(tmp_path, "tokenizer.json"), "w"))
    with pytest.raises(host.WhisperError, match="Failed to get token ID for: <|notimestamps|>".replace("|", r"\|")):
        d.blocking_try_to_model_from_dir(str(tmp_path))


def test_quantized_q8_0_gguf_checkpoint_loads_and_transcribes(tmp_path):
    """§8(f)-4, the part that needs no Rust: ModelType::QuantizedTinyEn reads config-tiny-en.json / tokenizer-tiny-en.json /
    model-tiny-en-q80.gguf (the file names of monolingual.rs / multilingual.rs:195-199), dequantises the Q8_0 matrices and runs
    the ordinary fp16 path.  Expected = the same model built by handing over the dequantised tensors directly, and the oracle
    fed with those tensors.  (candle's CPU kernels for these checkpoints also quantise activations; that is not reproduced.)"""
    import json
    import os
    import gguf_writer
    name = "test-d128"
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    script = common.transcript_script(tk, n_segments=3, words_per_segment=4)
    over = common.scripted_overrides(cfg, tk, script)
    words = _write_checkpoint_dir(str(tmp_path), cfg, tk, synth.synth_weights(cfg, 0, over))
    os.rename(tmp_path / "config.json", tmp_path / "config-tiny-en.json")
    os.rename(tmp_path / "tokenizer.json", tmp_path / "tokenizer-tiny-en.json")
    os.remove(tmp_path / "model.safetensors")
    deq = gguf_writer.write_gguf(str(tmp_path / "model-tiny-en-q80.gguf"), synth.synth_weights(cfg, 0, over))
    deq16 = {n: a.astype(np.float16) for n, a in deq.items()}          # what the loader uploads (fp16 like every weight)
    d = host.Definition(host.ModelType.QuantizedTinyEn, host.SelectedDevice.Rocm(0))
    model = d.blocking_try_to_model_from_dir(str(tmp_path))
    pcm = synth.synth_pcm(0)
    segs = model.transcribe(pcm, final_chunk=True)
    ref_model = d.blocking_try_to_model(cfg, tk, tk.en, tk.transcribe, ((n, deq16[n]) for n, _ in synth.synth_weights(cfg, 0, over)))
    assert segs == ref_model.transcribe(pcm, final_chunk=True)
    om = common.oracle_module().OracleModel(cfg, tk, tk.en, tk.transcribe)
    for n, _ in synth.synth_weights(cfg, 0, over):
        om.set_tensor(n, deq16[n].astype(np.float32))
    ref, buf, info = om.transcribe(pcm, assets_io.mel_filters(cfg.num_mel_bins), final_chunk=True)
    assert segs == ref and len(segs) == 3          # 8-bit weights still follow the scripted transcript
    assert abs(model.last_result()["avg_logprob"] - info["avg_logprob"]) < 5e-3
    assert model.last_text() == "".join("".join(f" w{t}" for t in s) for s in segs)
    model.close(); ref_model.close()
    os.remove(tmp_path / "model-tiny-en-q80.gguf")
    with pytest.raises(host.WhisperError, match="cannot open"):
        d.blocking_try_to_model_from_dir(str(tmp_path))


def test_checkpoint_directory_against_independent_loaders_with_a_real_bpe_tokenizer_and_translate(tmp_path):
    """§8(f)-1/3 with nothing checked HIP-against-HIP: the checkpoint directory holds the fixture tokenizer (byte-level BPE
    with real merges, multi-byte text, special and timestamp tokens; tests/golden/make_asset_fixtures.py) and a
    model.safetensors with F16, F32 and BF16 tensors written by the `safetensors` binding.  Expected side: the ORACLE fed
    the tensors through safetensors' own loader, special ids and text through the `tokenizers` binding (the crates the
    reference calls: monolingual.rs:237-239, mod.rs:86-90, model.rs:147).  Task = translate, language fixed to <|ja|>:
    the prompt must be [sot, <|ja|>, <|translate|>] (model.rs:285-289, multilingual.rs:383-386)."""
    import json
    import os
    import shutil
    torch = pytest.importorskip("torch")
    tokenizers = pytest.importorskip("tokenizers")
    from safetensors.torch import load_file, save_file
    from norma_amd import vocab
    gold = os.path.join(common.ROOT, "tests", "golden", "assets")
    shutil.copy(os.path.join(gold, "tokenizer.json"), tmp_path / "tokenizer.json")
    tok = tokenizers.Tokenizer.from_file(str(tmp_path / "tokenizer.json"))
    V = tok.get_vocab_size(with_added_tokens=True)
    tid = tok.token_to_id
    tk = vocab.SpecialTokens(V, tid("<|endoftext|>"), tid("<|startoftranscript|>"), tid("<|en|>"), tid("<|translate|>"),
                             tid("<|transcribe|>"), tid("<|nospeech|>"), tid("<|notimestamps|>"), tid("<|0.00|>"), 8)
    assert tid("<|1.00|>") == tk.one_sec and tk.no_timestamps + 1 == tk.zero_sec
    cfg = common.make_config("test-d128", vocab_size=V, suppress_tokens=[1, 2, 7, tid("<|startoflm|>"), tid("<|startofprev|>")])
    s1 = tok.encode("naïve café 日本語のテキスト", add_special_tokens=False).ids
    s2 = tok.encode(" emoji 🙂 Привет", add_special_tokens=False).ids
    script = [tk.zero_sec] + s1 + [tk.zero_sec + 40, tk.zero_sec + 42] + s2 + [tk.eot]   # <|t|> text <|t|> <|t|> text eot
    assert not set(script) & set(cfg.suppress_tokens)
    over = common.scripted_overrides(cfg, tk, script)
    tensors = {}
    for n, a in synth.synth_weights(cfg, 0, over):
        t = torch.from_numpy(a)
        if n.endswith("layer_norm.weight") or n.endswith("layer_norm.bias"):
            tensors[n] = t                                    # F32
        elif n.endswith(".bias"):
            tensors[n] = t.to(torch.bfloat16)                 # BF16 (widened exactly by the loader)
        else:
            tensors[n] = t.to(torch.float16)                  # F16
    tensors["proj_out.weight"] = tensors["model.decoder.embed_tokens.weight"].clone()   # in HF files, not read by candle
    save_file(tensors, str(tmp_path / "model.safetensors"))
    with open(tmp_path / "config.json", "w") as f:
        json.dump(cfg.to_dict(), f)
    # ---- expected: oracle + the reference's crates ----
    ja, task = tid("<|ja|>"), tk.translate
    om = common.oracle_module().OracleModel(cfg, tk, ja, task)
    for n, t in load_file(str(tmp_path / "model.safetensors")).items():
        if n != "proj_out.weight":
            om.set_tensor(n, t.to(torch.float32).numpy().astype(np.float16).astype(np.float32) if t.dtype == torch.float16
                          else t.to(torch.float32).numpy())
    pcm = synth.synth_pcm(4)
    filt = assets_io.mel_filters(cfg.num_mel_bins)
    ref_segs, _, info = om.transcribe(pcm, filt, final_chunk=True)
    ref_dec = om.decode(om.encoder_forward(common.oracle_module().pcm_to_mel(pcm, filt)))
    assert ref_dec["tokens"] == [tk.sot, ja, task] + script
    assert ref_segs == [s1, s2]
    want_text = "".join(tok.decode(s, skip_special_tokens=True) for s in ref_segs)
    assert want_text == "naïve café 日本語のテキスト emoji 🙂 Привет"
    # ---- the HIP host layer from the directory ----
    d = host.Definition(host.ModelType.Tiny, host.SelectedDevice.Rocm(0))
    model = d.blocking_try_to_model_from_dir(str(tmp_path), language="<|ja|>", translate=True)
    segs = model.transcribe(pcm, final_chunk=True)
    assert segs == ref_segs
    assert model.last_text() == want_text
    assert abs(model.last_result()["avg_logprob"] - info["avg_logprob"]) < 5e-3
    assert model.last_result()["n_tokens"] == len(ref_dec["tokens"])
    model.close(); om.close()
