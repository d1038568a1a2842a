#!/usr/bin/env python3
"""tests/golden/language_tokens.json: the language tokens in `Language::iter()` order, derived from the reference's
src/models/whisper/languages.rs (enum variant order, lines 7-107, joined with the `token()` match arms, lines 122+).
model.rs:204 and multilingual.rs:395-398 index the language logits in exactly this order.  Data only: 99 strings.
Run in the build container (needs /root/reference):  python tests/golden/make_language_fixture.py"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def parse(path):
    src = open(path).read()
    enum_body = re.search(r"pub enum Language \{(.*?)\n\}", src, re.S).group(1)
    variants = re.findall(r"^\s*([A-Z][A-Za-z]*),\s*$", enum_body, re.M)
    token_fn = src[src.index("pub fn token(&self)"):]
    arms = dict(re.findall(r"Language::([A-Za-z]+)\s*=>\s*\"(<\|[a-z]+\|>)\"", token_fn))
    return [arms[v] for v in variants]


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/models/whisper/languages.rs"
    toks = parse(ref)
    with open(os.path.join(HERE, "language_tokens.json"), "w") as f:
        json.dump({"source": "src/models/whisper/languages.rs (MikeIvanichev/norma @ 2024_10_08)", "tokens": toks}, f, indent=0)
    print(len(toks), "language tokens")
