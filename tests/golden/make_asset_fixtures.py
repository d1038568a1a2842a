#!/usr/bin/env python3
"""Fixtures that pin the asset readers (norma_amd/csrc/norma_assets.hpp) to the crates the reference itself uses.

The reference reads its tokenizer through the `tokenizers` crate (Tokenizer::from_file, token_to_id, decode with
skip_special_tokens = true: src/models/whisper/monolingual.rs:349, mod.rs:86-90, model.rs:147) and its weights through
`safetensors` (VarBuilder::from_mmaped_safetensors, monolingual.rs:237-239).  Both crates ship Python bindings of the SAME
Rust code, importable in the build container (tokenizers 0.22, safetensors 0.8).  This script uses them to write

  assets/tokenizer.json        a byte-level BPE tokenizer with real merges (trained on the multilingual corpus below),
                               the Whisper special tokens (special = true) and timestamp tokens (special = false)
  assets/tokenizer_cases.json  id sequences with the text `Tokenizer.decode(ids, skip_special_tokens)` returns for them
                               (multi-byte UTF-8 split across tokens, sequences that end inside a code point -> U+FFFD,
                               special / timestamp tokens mixed in) and `token_to_id` answers, incl. names that do not exist
  assets/weights.safetensors   F32, F16 and BF16 tensors (2-D, 1-D, 3-D) written by safetensors.numpy / torch
  assets/weights_cases.json    per-tensor dtype, shape, sum and sum of magnitudes of the values widened to f32

Run from the repository root:  python tests/golden/make_asset_fixtures.py   (deterministic; committed output)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "assets")

CORPUS = [
    "the quick brown fox jumps over the lazy dog", "she sells sea shells by the sea shore", "hello world, hello whisper",
    "naïve café déjà vu coöperate", "Grüße aus München, schöne Straße", "¿dónde está la biblioteca? ¡aquí!",
    "日本語のテキストを書きます", "今日は天気がいいですね", "中文分词测试，你好世界", "한국어 텍스트 예시입니다",
    "Привет, мир! Как дела?", "مرحبا بالعالم", "emoji 🙂🙂 rocket 🚀 family 👨‍👩‍👧", "numbers 1234567890 and symbols #$%&*()",
] * 4

LANGS = ["en", "zh", "de", "es", "ru", "ko", "fr", "ja"]


def build_tokenizer():
    from tokenizers import AddedToken, Tokenizer, decoders, models, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=600, min_frequency=2, show_progress=False,
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), special_tokens=[])
    tok.train_from_iterator(CORPUS, trainer)
    specials = ["<|endoftext|>", "<|startoftranscript|>"] + [f"<|{l}|>" for l in LANGS] + \
               ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
    tok.add_special_tokens([AddedToken(s, special=True) for s in specials])
    # timestamps are ordinary added tokens in the published Whisper tokenizers (special = false): decode keeps them
    tok.add_tokens([AddedToken("<|%.2f|>" % (0.02 * i), special=False) for i in range(0, 101)])
    return tok


def tokenizer_fixture():
    tok = build_tokenizer()
    path = os.path.join(OUT, "tokenizer.json")
    tok.save(path, pretty=False)
    from tokenizers import Tokenizer
    tok = Tokenizer.from_file(path)   # what the reference does
    rng = np.random.default_rng(7)
    V = tok.get_vocab_size(with_added_tokens=True)
    sot, eot = tok.token_to_id("<|startoftranscript|>"), tok.token_to_id("<|endoftext|>")
    t0 = tok.token_to_id("<|0.00|>")
    cases = []

    def add(ids, note):
        ids = [int(i) for i in ids]
        cases.append({"note": note, "ids": ids, "skip_special": tok.decode(ids, skip_special_tokens=True),
                      "keep_special": tok.decode(ids, skip_special_tokens=False)})
    for text in ["hello world", "naïve café", "日本語のテキスト", "emoji 🙂 rocket 🚀", "Привет, мир!", "مرحبا", " leading space", "x"]:
        ids = tok.encode(text, add_special_tokens=False).ids
        add(ids, f"encode({text!r})")
        add([sot, tok.token_to_id("<|en|>"), tok.token_to_id("<|transcribe|>"), t0] + ids + [t0 + 50, eot], "prompt + timestamps + " + text)
        if len(ids) > 2:
            add(ids[:-1], "truncated inside the text (may end inside a UTF-8 sequence)")
            add(ids[1:], "first token dropped (may start with a continuation byte)")
            add(ids[::-1], "reversed (broken UTF-8 in the middle)")
    for _ in range(40):   # random id soup over the whole vocabulary, byte tokens included
        add(rng.integers(0, V, size=int(rng.integers(1, 24))), "random ids")
    add([], "empty")
    add([eot, sot], "special tokens only")
    add([V + 5, 3, V + 100], "ids beyond the vocabulary are ignored")
    names = ["<|startoftranscript|>", "<|endoftext|>", "<|transcribe|>", "<|translate|>", "<|nospeech|>", "<|nocaptions|>",
             "<|notimestamps|>", "<|0.00|>", "<|1.00|>", "<|2.00|>", "<|2.02|>", "<|en|>", "<|ja|>", "<|xx|>", "hello", "Ġthe", "the", "a", ""]
    lookups = {n: tok.token_to_id(n) for n in names}
    with open(os.path.join(OUT, "tokenizer_cases.json"), "w") as f:
        json.dump({"made_with": "tokenizers " + __import__("tokenizers").__version__, "vocab_size": V, "cases": cases,
                   "token_to_id": lookups}, f, ensure_ascii=True, indent=0)
    return len(cases), V


def weights_fixture():
    import torch
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(11)
    t = {
        "model.encoder.conv1.weight": torch.randn(6, 4, 3, generator=g, dtype=torch.float32),
        "model.encoder.conv1.bias": torch.randn(6, generator=g, dtype=torch.float32),
        "model.decoder.embed_tokens.weight": torch.randn(37, 8, generator=g).to(torch.float16),
        "model.decoder.layers.0.fc1.weight": torch.randn(16, 8, generator=g).to(torch.bfloat16),
        "model.decoder.layer_norm.bias": torch.randn(8, generator=g).to(torch.float16),
        "scalar_like": torch.tensor([3.5], dtype=torch.float32),
    }
    path = os.path.join(OUT, "weights.safetensors")
    save_file(t, path, metadata={"format": "pt"})
    from safetensors import safe_open
    exp = {}
    with safe_open(path, framework="pt") as f:   # read back through the crate
        for k in f.keys():
            v = f.get_tensor(k)
            w = v.to(torch.float32).double()
            exp[k] = {"dtype": {torch.float32: "F32", torch.float16: "F16", torch.bfloat16: "BF16"}[v.dtype],
                      "shape": list(v.shape), "sum": float(w.sum()), "sum_abs": float(w.abs().sum())}
    with open(os.path.join(OUT, "weights_cases.json"), "w") as f:
        json.dump({"made_with": "safetensors " + __import__("safetensors").__version__, "tensors": exp}, f, indent=0)
    return len(exp)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    n, v = tokenizer_fixture()
    print(f"tokenizer.json: vocab {v}, {n} decode cases")
    print(f"weights.safetensors: {weights_fixture()} tensors")
