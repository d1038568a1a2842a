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
