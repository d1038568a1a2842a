"""Whisper `Config` (the fields candle deserialises from HF `config.json`; SURVEY.md 3.3-1,
used by the reference at `src/models/whisper/monolingual.rs:347`) and the shapes of the published
checkpoints named by `ModelType` (`monolingual.rs:32-111`, `multilingual.rs:48-106`)."""
import json
from dataclasses import dataclass, field, asdict
from typing import List

from . import vocab as _vocab


@dataclass
class Config:
    num_mel_bins: int
    max_source_positions: int
    d_model: int
    encoder_attention_heads: int
    encoder_layers: int
    vocab_size: int
    max_target_positions: int
    decoder_attention_heads: int
    decoder_layers: int
    suppress_tokens: List[int] = field(default_factory=list)

    @classmethod
    def from_json(cls, path: str) -> "Config":
        with open(path) as f:
            j = json.load(f)
        return cls(**{k: j[k] for k in cls.__dataclass_fields__ if k in j})

    def to_dict(self):
        return asdict(self)


def _cfg(n_mel, d, heads, enc, dec, vocab_version):
    tk = _vocab.VOCABS[vocab_version]
    return Config(n_mel, 1500, d, heads, enc, tk.n_vocab, 448, heads, dec,
                  _vocab.default_suppress_tokens(vocab_version))


# name -> (Config factory, vocab version).  Sizes are the HF config.json values.
PRESETS = {
    "tiny.en": (lambda: _cfg(80, 384, 6, 4, 4, "EnV1"), "EnV1"),
    "base.en": (lambda: _cfg(80, 512, 8, 6, 6, "EnV1"), "EnV1"),
    "small.en": (lambda: _cfg(80, 768, 12, 12, 12, "EnV1"), "EnV1"),
    "medium.en": (lambda: _cfg(80, 1024, 16, 24, 24, "EnV1"), "EnV1"),
    "distil-medium.en": (lambda: _cfg(80, 1024, 16, 24, 2, "V1"), "V1"),
    "distil-large-v2": (lambda: _cfg(80, 1280, 20, 32, 2, "V1"), "V1"),
    "distil-large-v3": (lambda: _cfg(128, 1280, 20, 32, 2, "V2"), "V2"),
    "tiny": (lambda: _cfg(80, 384, 6, 4, 4, "V1"), "V1"),
    "base": (lambda: _cfg(80, 512, 8, 6, 6, "V1"), "V1"),
    "small": (lambda: _cfg(80, 768, 12, 12, 12, "V1"), "V1"),
    "medium": (lambda: _cfg(80, 1024, 16, 24, 24, "V1"), "V1"),
    "large": (lambda: _cfg(80, 1280, 20, 32, 32, "V1"), "V1"),
    "large-v2": (lambda: _cfg(80, 1280, 20, 32, 32, "V1"), "V1"),
    "large-v3": (lambda: _cfg(128, 1280, 20, 32, 32, "V2"), "V2"),
    # reduced shapes for fast parity tests (head dim stays 64 like every Whisper size)
    "test-d128": (lambda: _cfg(80, 128, 2, 2, 2, "EnV1"), "EnV1"),
    "test-d256-mel128": (lambda: _cfg(128, 256, 4, 2, 2, "V2"), "V2"),
}


def preset(name: str) -> Config:
    return PRESETS[name][0]()


def preset_vocab(name: str) -> str:
    return PRESETS[name][1]
