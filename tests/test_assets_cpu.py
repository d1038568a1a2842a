"""CPU: the asset readers of the host layer (norma_amd/csrc/norma_assets.hpp) against fixtures written by the Python
bindings of the very crates the reference uses -- `tokenizers` (Tokenizer::from_file / token_to_id / decode with
skip_special_tokens = true: monolingual.rs:349, mod.rs:86-90, model.rs:147) and `safetensors`
(VarBuilder::from_mmaped_safetensors, monolingual.rs:237-239) -- see tests/golden/make_asset_fixtures.py; and the
language table against the reference's languages.rs (tests/golden/make_language_fixture.py).  Everything goes through the
C entry points of include/norma_host.h; no GPU is touched."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

import common
from norma_amd import hip

GOLD = os.path.join(common.ROOT, "tests", "golden")
ASSETS = os.path.join(GOLD, "assets")


@pytest.fixture(scope="module")
def L():
    lib = hip.load_library()
    lib.nm_tokenizer_open.restype = C.c_void_p
    lib.nm_tokenizer_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    lib.nm_tokenizer_free.argtypes = [C.c_void_p]
    lib.nm_tokenizer_token_to_id.argtypes = [C.c_void_p, C.c_char_p]
    lib.nm_tokenizer_decode.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.c_char_p, C.c_int]
    lib.nm_safetensors_list.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    lib.nm_language_code.restype = C.c_char_p
    lib.nm_language_code.argtypes = [C.c_int]
    return lib


def _decode(L, t, ids, skip):
    a = (C.c_uint32 * max(len(ids), 1))(*ids)
    buf = C.create_string_buffer(1 << 16)
    n = L.nm_tokenizer_decode(t, a, len(ids), int(skip), buf, len(buf))
    assert 0 <= n < len(buf)
    return buf.raw[:n].decode("utf-8")     # must be valid UTF-8 (String::from_utf8_lossy)


def test_tokenizer_json_decode_and_lookups_match_the_tokenizers_crate(L):
    with open(os.path.join(ASSETS, "tokenizer_cases.json")) as f:
        fx = json.load(f)
    err = C.create_string_buffer(512)
    t = L.nm_tokenizer_open(os.path.join(ASSETS, "tokenizer.json").encode(), err, len(err))
    assert t, err.value
    assert len(fx["cases"]) >= 60
    lossy = multibyte = 0
    for c in fx["cases"]:
        assert _decode(L, t, c["ids"], True) == c["skip_special"], c["note"]
        assert _decode(L, t, c["ids"], False) == c["keep_special"], c["note"]
        lossy += "�" in c["skip_special"]
        multibyte += any(ord(ch) > 0x7f and ch != "�" for ch in c["skip_special"])
    assert lossy >= 5 and multibyte >= 10          # the fixture does exercise broken and multi-byte sequences
    for name, want in fx["token_to_id"].items():   # None = whisper::Error::TokenId
        assert L.nm_tokenizer_token_to_id(t, name.encode()) == (-1 if want is None else want), name
    assert fx["token_to_id"]["<|nocaptions|>"] is None and fx["token_to_id"]["<|nospeech|>"] is not None
    L.nm_tokenizer_free(t)


def test_safetensors_reader_matches_the_safetensors_crate(L):
    with open(os.path.join(ASSETS, "weights_cases.json")) as f:
        fx = json.load(f)["tensors"]
    buf = C.create_string_buffer(1 << 16)
    n = L.nm_safetensors_list(os.path.join(ASSETS, "weights.safetensors").encode(), buf, len(buf))
    assert n == len(fx) == 6, buf.value
    seen = set()
    for line in buf.value.decode().strip().splitlines():
        name, dtype, shape, s, sa = line.split()
        e = fx[name]
        assert dtype == e["dtype"] and [int(v) for v in shape.split("x")] == e["shape"], name
        assert float(s) == pytest.approx(e["sum"], rel=1e-12, abs=1e-12) and float(sa) == pytest.approx(e["sum_abs"], rel=1e-12), name
        seen.add(dtype)
    assert seen == {"F32", "F16", "BF16"}


def _st_file(tmp_path, name, header: dict, data: bytes, header_len=None):
    h = json.dumps(header).encode()
    p = tmp_path / name
    p.write_bytes(struct.pack("<Q", len(h) if header_len is None else header_len) + h + data)
    return str(p).encode()


def test_malformed_checkpoints_are_rejected_not_read_past_the_mapping(L, tmp_path):
    """A checkpoint is untrusted input: lying sizes must fail with a message, never index past the file."""
    buf = C.create_string_buffer(4096)
    ok = {"w": {"dtype": "F32", "shape": [2, 3], "data_offsets": [0, 24]}}
    assert L.nm_safetensors_list(_st_file(tmp_path, "ok.st", ok, b"\0" * 24), buf, len(buf)) == 1
    cases = {
        "header length near 2^64": _st_file(tmp_path, "a.st", ok, b"\0" * 24, header_len=2 ** 64 - 4),
        "header longer than the file": _st_file(tmp_path, "b.st", ok, b"\0" * 24, header_len=10 ** 6),
        "shape says 6 elements, data range holds 0 bytes": _st_file(tmp_path, "c.st", {"w": {"dtype": "F32", "shape": [2, 3], "data_offsets": [0, 0]}}, b"\0" * 24),
        "data range past the end": _st_file(tmp_path, "d.st", {"w": {"dtype": "F16", "shape": [64], "data_offsets": [0, 128]}}, b"\0" * 24),
        "shape product overflows": _st_file(tmp_path, "e.st", {"w": {"dtype": "F32", "shape": [2 ** 40, 2 ** 40], "data_offsets": [0, 24]}}, b"\0" * 24),
        "end before begin": _st_file(tmp_path, "f.st", {"w": {"dtype": "F32", "shape": [1], "data_offsets": [8, 4]}}, b"\0" * 24),
        "truncated json": _st_file(tmp_path, "g.st", ok, b"", header_len=10),
        # a dtype the reader does not widen used to skip the size cross-check, and st_to_f32 sized a vector from
        # the declared shape before looking at the dtype (bad_alloc through an extern "C" function = abort)
        "unknown dtype, huge shape": _st_file(tmp_path, "h.st", {"w": {"dtype": "I64", "shape": [2 ** 40, 1024], "data_offsets": [0, 24]}}, b"\0" * 24),
    }
    for what, path in cases.items():
        assert L.nm_safetensors_list(path, buf, len(buf)) == -1, what
        assert buf.value, what
    # an unknown dtype with a plausible shape is listed (sums 0: never converted), not rejected and not sized from
    listed = _st_file(tmp_path, "i.st", {"w": {"dtype": "I64", "shape": [3], "data_offsets": [0, 24]}}, b"\1" * 24)
    assert L.nm_safetensors_list(listed, buf, len(buf)) == 1
    assert buf.value.decode().split()[:3] == ["w", "I64", "3"] and float(buf.value.decode().split()[3]) == 0.0
    # GGUF: element count / offset arithmetic
    import gguf_writer
    good = tmp_path / "ok.gguf"
    gguf_writer.write_gguf(str(good), [("a.weight", np.arange(64, dtype=np.float32).reshape(2, 32))])
    lib_buf = C.create_string_buffer(4096)
    L.nm_gguf_list.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    assert L.nm_gguf_list(str(good).encode(), lib_buf, len(lib_buf)) == 1
    raw = bytearray(good.read_bytes())
    bad = tmp_path / "trunc.gguf"
    bad.write_bytes(bytes(raw[:len(raw) - 40]))                      # data runs past the end
    assert L.nm_gguf_list(str(bad).encode(), lib_buf, len(lib_buf)) == -1
    off = raw.find(b"a.weight") + len(b"a.weight") + 4              # n_dims (u32) then dims[0] (u64)
    huge = bytearray(raw)
    huge[off:off + 8] = struct.pack("<Q", 2 ** 62)                   # innermost dim absurdly large
    bad2 = tmp_path / "huge.gguf"
    bad2.write_bytes(bytes(huge))
    assert L.nm_gguf_list(str(bad2).encode(), lib_buf, len(lib_buf)) == -1


def test_language_table_is_the_references_language_iter_order(L):
    with open(os.path.join(GOLD, "language_tokens.json")) as f:
        want = json.load(f)["tokens"]
    assert len(want) == 99 and want[0] == "<|en|>" and want[-1] == "<|su|>"
    ref = "/root/reference/src/models/whisper/languages.rs"
    if os.path.exists(ref):                                         # build container: re-derive from the reference file itself
        import importlib.util
        spec = importlib.util.spec_from_file_location("mlf", os.path.join(GOLD, "make_language_fixture.py"))
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        assert m.parse(ref) == want
    got = []
    for i in range(99):
        got.append("<|" + L.nm_language_code(i).decode() + "|>")
    assert got == want
    assert L.nm_language_code(99) is None and L.nm_language_code(-1) is None
