"""Whisper vocabulary layouts: special-token ids and default suppress lists.

The reference resolves these by name through `tokenizers` (`src/models/whisper/mod.rs:86-90`,
`src/models/whisper/monolingual.rs:376-384`); no `tokenizer.json` is available offline, so the ids
below come from the three published vocab layouts (`VocabVersion`, `src/models/whisper/mod.rs:57-62`)
as tabulated in SURVEY.md 8(c).  When a real tokenizer file is supplied the host layer resolves
the names through it instead (`norma_amd.whisper.special_tokens_from_tokenizer`).
"""
from dataclasses import dataclass
from typing import List

# OpenAI `non_speech_tokens` for the English-only (EnV1) and multilingual (V1) vocabularies, the
# lists real checkpoints ship as `suppress_tokens` in config.json.
_NON_SPEECH_EN = [
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 357, 366,
    438, 532, 685, 705, 796, 930, 1058, 1220, 1267, 1279, 1303, 1343, 1377, 1391, 1635, 1782, 1875,
    2162, 2361, 2488, 3467, 4008, 4211, 4600, 4808, 5299, 5855, 6329, 7203, 9609, 9959, 10563, 10786,
    11420, 11709, 11907, 13163, 13697, 13700, 14808, 15306, 16410, 16791, 17992, 19203, 19510, 20724,
    22305, 22935, 27007, 30109, 30420, 33409, 34949, 40283, 40493, 40549, 47282, 49146, 50257, 50359,
    50360, 50361]
_NON_SPEECH_MULTI = [
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503,
    522, 542, 873, 893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268, 3536,
    3846, 3961, 4183, 4667, 6585, 6647, 7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331, 12562,
    13793, 14157, 14635, 15265, 15618, 16553, 16604, 18362, 18956, 20075, 21675, 22520, 26130, 26161,
    26435, 28279, 29464, 31650, 32302, 32470, 36865, 42863, 47425, 49870, 50254, 50258, 50360, 50361,
    50362]


@dataclass(frozen=True)
class SpecialTokens:
    """Token ids norma's `Model` carries (`src/models/whisper/model.rs:37-41`) plus the two
    timestamp ids used to build `first_token_supress` (`monolingual.rs:419-430`)."""
    n_vocab: int
    eot: int
    sot: int
    en: int            # first language token (<|en|>); language i of `Language::iter()` is en + i
    translate: int
    transcribe: int
    no_speech: int
    no_timestamps: int
    zero_sec: int      # <|0.00|>
    n_languages: int

    @property
    def one_sec(self) -> int:  # <|1.00|>
        return self.zero_sec + 50


EN_V1 = SpecialTokens(51864, 50256, 50257, 50258, 50357, 50358, 50361, 50362, 50363, 99)
V1 = SpecialTokens(51865, 50257, 50258, 50259, 50358, 50359, 50362, 50363, 50364, 99)
V2 = SpecialTokens(51866, 50257, 50258, 50259, 50359, 50360, 50363, 50364, 50365, 100)

VOCABS = {"EnV1": EN_V1, "V1": V1, "V2": V2}


def default_suppress_tokens(vocab: str) -> List[int]:
    """Fixture decision of SURVEY.md 8(d): EnV1/V1 use the published lists; V2 reuses V1's with
    every id >= 50359 incremented (V2 inserted one language token)."""
    if vocab == "EnV1":
        return list(_NON_SPEECH_EN)
    if vocab == "V1":
        return list(_NON_SPEECH_MULTI)
    if vocab == "V2":
        return [t + 1 if t >= 50359 else t for t in _NON_SPEECH_MULTI]
    raise ValueError(f"unknown vocab version {vocab!r}")
