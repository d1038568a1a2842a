"""CPU-side checks of the C++ host layer that mirrors norma's plugin API (no compute calls)."""
import re
import os

import numpy as np
import pytest

import common
from norma_amd import config, hip, host


def test_host_shim_exports_every_declared_symbol():
    L = hip.load_library()
    hdr = os.path.join(os.path.dirname(hip.HEADER_PATH), "norma_host.h")
    syms = sorted(set(re.findall(r"\b(nm_[a-z_0-9]+)\s*\(", open(hdr).read())))
    assert len(syms) >= 12
    assert not [s for s in syms if not hasattr(L, s)]


def test_definition_defaults_and_clamps_match_the_reference():
    # Definition::new -> CommonModelParams::new(SAMPLE_RATE * 25, 3, 3)  (monolingual.rs:124-130, mod.rs:77-85)
    d = host.Definition(host.ModelType.DistilLargeEnV3, host.SelectedDevice.Rocm(0))
    assert d.max_chunk_len == 16000 * 25
    assert d.data_buffer_size == 3 + 2        # "since we are using Thingbuff the actual buff size would be n - 2"
    d.set_responsiveness(10_000)              # monolingual.rs:147-156
    assert d.max_chunk_len == 160_000
    for bad in (999, 30_001):
        with pytest.raises(host.WhisperError, match="respnsivness"):
            d.set_responsiveness(bad)
    d.set_data_buffer_size(0)
    assert d.data_buffer_size == 2


def test_non_rocm_devices_are_rejected_loudly():
    cfg = config.preset("test-d128")
    tk = common.tokens_for("test-d128")
    d = host.Definition(host.ModelType.TinyEn, host.SelectedDevice.Cpu())
    with pytest.raises(host.WhisperError, match="Rocm"):
        d.blocking_try_to_model(cfg, tk, tk.en, tk.transcribe, [])


def test_rocm_without_gpu_fails_with_the_backend_message():
    if hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    cfg = config.preset("test-d128")
    tk = common.tokens_for("test-d128")
    d = host.Definition(host.ModelType.TinyEn, host.SelectedDevice.Rocm(0))
    with pytest.raises(host.WhisperError, match="no HIP device"):
        d.blocking_try_to_model(cfg, tk, tk.en, tk.transcribe, [])


def test_gguf_q8_0_reader_reconstructs_the_tensors(tmp_path):
    """The quantised checkpoints of the reference (model-{tiny,tiny-en}-q80.gguf, multilingual.rs:195-199) are GGUF files
    with Q8_0 matrices; the C++ reader must find every tensor, un-reverse ggml's dimension order and dequantise."""
    import gguf_writer
    from norma_amd import host
    rng = np.random.default_rng(2)
    tensors = [("model.encoder.layers.0.fc1.weight", rng.standard_normal((96, 64)).astype(np.float32)),
               ("model.encoder.layers.0.fc1.bias", rng.standard_normal(96).astype(np.float32)),
               ("model.encoder.conv1.weight", rng.standard_normal((8, 5, 3)).astype(np.float32)),     # 3-D: stays f32
               ("model.decoder.embed_tokens.weight", rng.standard_normal((40, 32)).astype(np.float32) * 0.02),
               ("odd.weight", rng.standard_normal((4, 30)).astype(np.float32))]                       # row % 32 != 0: f32
    path = str(tmp_path / "m.gguf")
    deq = gguf_writer.write_gguf(path, tensors)
    got = host.gguf_list(path)
    assert [g[0] for g in got] == [n for n, _ in tensors]
    for (name, ty, shape, s), (_, a) in zip(got, tensors):
        assert shape == a.shape
        assert ty == (8 if a.ndim == 2 and a.shape[1] % 32 == 0 else 0)
        assert abs(s - float(deq[name].astype(np.float64).sum())) <= 1e-4 * (1.0 + abs(s))
        if ty == 8:   # Q8_0 keeps ~7 bits per value
            assert 1e-4 < np.abs(deq[name] - a).max() <= np.abs(a).max() / 127.0 * 0.51 + 1e-3
    with open(path, "r+b") as f:       # corrupt magic -> a clean error, no crash
        f.write(b"XXXX")
    with pytest.raises(host.WhisperError, match="bad magic"):
        host.gguf_list(path)
