"""CPU tests of the oracle itself (the checker must be trusted before it checks anything).

The reference holds no golden vectors for this path ("parity unpinned"), so the oracle
is pinned three ways: (1) independent restatements written here in numpy / pure Python from the
reference's source text (mel, logit rules, segment iterator), (2) a secondary cross-check of the
transformer stack against `transformers`' Whisper built from a local config with tanh-GELU, and
(3) the committed golden fixtures under tests/golden (test_golden.py).
"""
import math

import numpy as np
import pytest

import common
from norma_amd import assets_io, config, synth, vocab
from oracle import oracle as O


# ---------------------------------------------------------------------------------------------------
# mel front end: SURVEY.md 3.3[A] restated with numpy (float64 FFT), independent of the C code
# ---------------------------------------------------------------------------------------------------
def mel_numpy(pcm, filt):
    n = len(pcm)
    n_len = n // 160
    if n_len % 1500:
        n_len = (n_len // 1500 + 1) * 1500
    n_len += 1500
    x = np.zeros(n_len * 160 + 400, np.float64)
    x[:n] = pcm
    hann = 0.5 * (1 - np.cos(2 * np.pi * np.arange(400) / 400))
    idx = np.arange(n_len)[:, None] * 160 + np.arange(400)[None, :]
    fr = x[idx] * hann
    fr[idx >= n_len * 160] = 0
    sp = np.fft.fft(fr, axis=1)
    P = sp.real ** 2 + sp.imag ** 2
    P2 = P[:, :201].copy()
    P2[:, 1:200] += P[:, 399:200:-1]   # interior bins doubled, bins 0 and 200 not
    mel = np.log10(np.maximum(P2 @ filt.T.astype(np.float64), 1e-10)).T
    return np.maximum(mel, mel.max() - 8) / 4 + 1


@pytest.mark.parametrize("n_mel", [80, 128])
@pytest.mark.parametrize("n", [480000, 400000, 16000, 159])
def test_mel_matches_numpy_restatement(n_mel, n):
    filt = assets_io.mel_filters(n_mel)
    pcm = synth.synth_pcm(3, n)
    got = O.pcm_to_mel(pcm, filt)
    ref = mel_numpy(pcm, filt)
    assert got.shape == ref.shape
    # f32 recursive FFT vs f64 FFT: the reference algorithm's own rounding noise is ~2e-4 here
    assert np.abs(got - ref).max() < 1e-3


def test_mel_frame_counts():
    # probe values recorded in SURVEY.md 3.3[A]-2
    assert [O.mel_frames(n) for n in (16000, 400000, 480000, 720000)] == [3000, 4500, 4500, 6000]
    assert O.mel_frames(159) == 1500 and O.mel_frames(160) == 3000


def test_mel_filters_are_slaney_filterbank():
    # the asset equals a Slaney-normalised mel filterbank (0-8 kHz, 201 bins) to <= 1e-8
    def slaney(n_mels):
        def hz_to_mel(f):
            f = np.asarray(f, dtype=np.float64)
            mel = f / (200.0 / 3)
            log = f >= 1000.0
            return np.where(log, 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) / (np.log(6.4) / 27.0), mel)

        def mel_to_hz(m):
            m = np.asarray(m, dtype=np.float64)
            return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))
        fft = np.linspace(0, 8000, 201)
        pts = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(8000.0), n_mels + 2))
        fd = np.diff(pts)
        ramps = pts[:, None] - fft[None, :]
        w = np.maximum(0, np.minimum(-ramps[:-2] / fd[:-1, None], ramps[2:] / fd[1:, None]))
        return w * (2.0 / (pts[2:n_mels + 2] - pts[:n_mels]))[:, None]
    for n_mel in (80, 128):
        assert np.abs(assets_io.mel_filters(n_mel) - slaney(n_mel)).max() < 1e-7


# ---------------------------------------------------------------------------------------------------
# logit rules: src/models/whisper/model.rs:212-277 restated in pure Python/numpy
# ---------------------------------------------------------------------------------------------------
def rules_python(p, tokens, last_ts, tk, suppress):
    V = len(p)
    NINF = np.float32(-np.inf)
    p = p.astype(np.float32).copy()
    sup = np.zeros(V, np.float32)
    sup[list(suppress) + [tk.no_timestamps]] = NINF
    idx = np.arange(V)
    past = np.where((idx > tk.no_timestamps) & (idx <= last_ts), NINF, np.float32(0))
    non_ts = np.where(idx > tk.no_timestamps, np.float32(0), NINF)
    ts = np.where(idx > tk.no_timestamps, NINF, np.float32(0))
    p = p + sup
    l = tokens[-1]
    sl = tokens[-2] if len(tokens) >= 2 else None
    if l > tk.no_timestamps:
        if sl is not None and sl >= tk.eot:
            return p + ts
        return p + past + non_ts
    sum_ts = np.float32(p[tk.no_timestamps + 1:].sum(dtype=np.float32))
    max_text = p[:tk.no_timestamps].max()
    if sum_ts >= max_text:
        return p + past + non_ts
    return p + past


def _rule_cases(tk, V, rng):
    base = np.float32(rng.random(V) ** 8)
    base /= base.sum()
    text_peak = base.copy(); text_peak[1234] = 0.9
    cases = [
        ("ts_after_prompt", base, [tk.sot, tk.en, tk.transcribe, tk.zero_sec + 3], tk.zero_sec + 3),
        ("ts_after_text", base, [tk.sot, tk.en, tk.transcribe, tk.zero_sec, 400, tk.zero_sec + 90], tk.zero_sec + 90),
        ("ts_after_ts", base, [tk.sot, tk.en, tk.transcribe, tk.zero_sec, 400, tk.zero_sec + 90, tk.zero_sec + 95], tk.zero_sec + 95),
        ("text_ts_mass_wins", base, [tk.sot, tk.en, tk.transcribe, tk.zero_sec + 7, 400], tk.zero_sec + 7),
        ("text_text_wins", text_peak, [tk.sot, tk.en, tk.transcribe, tk.zero_sec + 7, 400], tk.zero_sec + 7),
        ("all_masked_H3", base, [tk.sot, tk.en, tk.transcribe, tk.zero_sec, 400, V - 1], V - 1),
    ]
    return cases


@pytest.mark.parametrize("vname", ["EnV1", "V2"])
def test_rules_match_python_restatement(vname):
    tk = vocab.VOCABS[vname]
    name = "test-d128" if vname == "EnV1" else "test-d256-mel128"
    cfg = common.make_config(name, encoder_layers=0, decoder_layers=0)
    om = O.OracleModel(cfg, tk, tk.en, tk.transcribe)
    rng = np.random.default_rng(7)
    for label, p, toks, last in _rule_cases(tk, cfg.vocab_size, rng):
        got = om.apply_rules(p, toks, last)
        ref = rules_python(p, toks, last, tk, cfg.suppress_tokens)
        assert np.array_equal(got, ref, equal_nan=True), label
    # the four masks (monolingual.rs:386-430)
    sup = om.mask(0)
    assert set(np.nonzero(np.isneginf(sup))[0]) == set(cfg.suppress_tokens) | {tk.no_timestamps}
    assert np.isneginf(om.mask(1)[:tk.no_timestamps + 1]).all() and (om.mask(1)[tk.no_timestamps + 1:] == 0).all()
    assert (om.mask(2)[:tk.no_timestamps + 1] == 0).all() and np.isneginf(om.mask(2)[tk.no_timestamps + 1:]).all()
    first = om.mask(3)
    assert (first[tk.zero_sec:tk.one_sec + 1] == 0).all() and np.isneginf(first[:tk.zero_sec]).all() \
        and np.isneginf(first[tk.one_sec + 1:]).all()


def test_argmax_total_cmp_semantics():
    # Iterator::max_by(total_cmp): last maximum wins, +NaN beats everything, -NaN loses to everything
    import ctypes as C
    L = O.lib()

    def am(v):
        a = np.asarray(v, dtype=np.float32)
        return L.wo_argmax_total(a.ctypes.data_as(C.POINTER(C.c_float)), len(a))
    assert am([0.1, 0.5, 0.5, 0.2]) == 2
    assert am([-np.inf] * 5) == 4
    assert am([0.3, np.nan, 0.9]) == 1
    neg_nan = np.frombuffer(np.uint32(0xFFC00000).tobytes(), dtype=np.float32)[0]
    assert am([neg_nan, -np.inf, neg_nan]) == 1
    assert am([0.0, -0.0]) == 0


# ---------------------------------------------------------------------------------------------------
# transformer stack vs transformers' Whisper (secondary cross-check, NOT the reference)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,over", [
    ("test-d128", {}),
    ("tiny.en", {}),                                                  # BASELINE.json configs 1-2 at full size (EnV1 vocabulary)
    ("large-v3", {"encoder_layers": 1, "decoder_layers": 1}),        # one layer of the d = 1280 / 20-head / 128-mel / V2 shape
])
def test_transformer_stack_matches_hf_whisper_with_tanh_gelu(name, over):
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    cfg = common.make_config(name, **over)
    tk = common.tokens_for(name)
    hc = tr.WhisperConfig(
        vocab_size=cfg.vocab_size, num_mel_bins=cfg.num_mel_bins, encoder_layers=cfg.encoder_layers,
        encoder_attention_heads=cfg.encoder_attention_heads, decoder_layers=cfg.decoder_layers,
        decoder_attention_heads=cfg.decoder_attention_heads, d_model=cfg.d_model, encoder_ffn_dim=4 * cfg.d_model,
        decoder_ffn_dim=4 * cfg.d_model, activation_function="gelu_pytorch_tanh", max_source_positions=1500,
        max_target_positions=448, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, pad_token_id=tk.eot,
        bos_token_id=tk.sot, eos_token_id=tk.eot, decoder_start_token_id=tk.sot)
    hf = tr.WhisperForConditionalGeneration(hc).eval().float()
    sd = hf.state_dict()
    om = O.OracleModel(cfg, tk, tk.en, tk.transcribe)
    for n, a in synth.synth_weights(cfg, 0):
        om.set_tensor(n, a)
        sd[n].copy_(torch.from_numpy(a))
    mel = O.pcm_to_mel(synth.synth_pcm(0), assets_io.mel_filters(cfg.num_mel_bins))
    xa = om.encoder_forward(mel)
    toks = np.array([tk.sot, tk.en, tk.transcribe, tk.zero_sec, 100, 2000, 30000, tk.zero_sec + 40], dtype=np.int32)
    logits = om.final_linear(om.decoder_forward(toks, xa, True))
    with torch.no_grad():
        feats = torch.from_numpy(mel[:, :3000]).unsqueeze(0)
        ref_enc = hf.model.encoder(feats).last_hidden_state[0].numpy()
        ref_logits = hf(input_features=feats, decoder_input_ids=torch.from_numpy(toks.astype(np.int64))[None]).logits[0].numpy()
    assert np.abs(xa - ref_enc).max() < 2e-4
    assert np.abs(logits - ref_logits).max() < 1e-4
    assert (logits.argmax(1) == ref_logits.argmax(1)).all()


# ---------------------------------------------------------------------------------------------------
# decode structure
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def small_model():
    name = "test-d128"
    cfg = config.preset(name)
    tk = common.tokens_for(name)
    om = common.build_oracle(cfg, tk)
    mel = O.pcm_to_mel(synth.synth_pcm(0), assets_io.mel_filters(cfg.num_mel_bins))
    return cfg, tk, om, om.encoder_forward(mel)


def test_kv_cache_is_bit_identical_to_reference_recompute(small_model):
    cfg, tk, om, xa = small_model
    a = om.decode(xa, use_kv_cache=False, max_new_tokens=20)
    b = om.decode(xa, use_kv_cache=True, max_new_tokens=20)
    assert a["tokens"] == b["tokens"]
    assert a["no_speech_prob"] == b["no_speech_prob"]
    assert a["avg_logprob"] == b["avg_logprob"] or (math.isnan(a["avg_logprob"]) and math.isnan(b["avg_logprob"]))


def test_decode_shape_and_reference_quirks(small_model):
    cfg, tk, om, xa = small_model
    r = om.decode(xa, use_kv_cache=True)
    t = r["tokens"]
    assert t[:3] == [tk.sot, tk.en, tk.transcribe]          # H4: .en prompts carry <|en|>
    assert tk.zero_sec <= t[3] <= tk.one_sec                  # first token forced into [0.00, 1.00]
    assert t[-1] == tk.eot and len(t) <= cfg.max_target_positions
    assert t[-2] <= tk.no_timestamps                          # timestamps before the final eot are stripped
    assert not any(x in cfg.suppress_tokens for x in t[3:-1])


def test_scripted_fixture_decodes_its_script(small_model):
    cfg, tk, _, xa = small_model
    script = common.transcript_script(tk, n_segments=3, words_per_segment=5)
    om = common.build_oracle(cfg, tk, overrides=common.scripted_overrides(cfg, tk, script))
    r = om.decode(xa, use_kv_cache=True, want_steps=True)
    assert r["tokens"] == [tk.sot, tk.en, tk.transcribe] + script
    n_new = len(script)
    st = r["steps"][:n_new]
    assert (st[:, 0] > 2.0 * np.maximum(st[:, 1], 0)).all()  # every argmax well separated


def test_inclusive_boxed_by_and_transcribe_semantics(small_model):
    cfg, tk, _, _ = small_model
    script = common.transcript_script(tk, n_segments=3, words_per_segment=4)
    om = common.build_oracle(cfg, tk, overrides=common.scripted_overrides(cfg, tk, script))
    filt = assets_io.mel_filters(cfg.num_mel_bins)
    pcm = synth.synth_pcm(0)
    # H5 per-chunk semantics: transcribe(chunk_of_480000, final_chunk=True) drains the whole slice
    segs, buf, info = om.transcribe(pcm, filt, final_chunk=True)
    assert len(buf) == 0 and info["n_slices"] == 1
    # expected segments: text between <ts> ... <ts|eot> pairs (utils.rs:29-48 + model.rs:100-147)
    exp, cur, inside = [], [], False
    for t in script:
        is_b = t > tk.no_timestamps or t == tk.eot
        if is_b and not inside:
            inside, cur = True, []
        elif is_b and inside:
            exp.append(cur); inside = False
        elif inside:
            cur.append(t)
    assert segs == exp and len(segs) == 3
    # not final, first segment does not start at 0.00 -> drain to the segment start and wait
    script2 = [tk.zero_sec + 50] + script[1:]
    om2 = common.build_oracle(cfg, tk, overrides=common.scripted_overrides(cfg, tk, script2))
    segs2, buf2, _ = om2.transcribe(pcm[:400000], filt, final_chunk=False)
    assert len(buf2) < 400000
