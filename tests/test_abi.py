"""CPU-side checks of the drop-in boundary: libnorma_hip.so loads and exports every symbol that
include/norma_hip.h declares, and the host layer fails loudly (never falls back) without a GPU.
No compute entry point is called here."""
import ctypes as C
import os

import numpy as np
import pytest

import common
from norma_amd import config, hip


def test_library_exports_every_declared_symbol():
    L = hip.load_library()
    syms = hip.declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, f"declared in norma_hip.h but not exported: {missing}"


def test_strict_memory_model_build_exists_and_exports_the_same_abi():
    """The -DNH_STRICT_MEMORY_MODEL variant of k_decode.hip (portable release / acquire spelling of the logit step's
    cross-workgroup hand-off) is compiled by every build so that it cannot rot; tests/test_gpu_kernels.py runs it."""
    assert os.path.exists(hip.STRICT_LIB_PATH), "make -C norma_amd/csrc builds libnorma_hip_strict.so"
    L = C.CDLL(hip.STRICT_LIB_PATH)
    missing = [s for s in hip.declared_symbols() if not hasattr(L, s)]
    assert not missing, missing


def test_no_oracle_or_cpu_fallback_in_product_sources():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dp, _, fns in os.walk(os.path.join(root, "norma_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if "whisper_oracle" in txt or "from oracle" in txt or "import oracle" in txt:
                    bad.append(os.path.join(dp, fn))
    assert not bad, f"product sources must not reference the oracle: {bad}"


def test_create_fails_loudly_without_a_device():
    if hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(hip.HipError) as ei:
        hip.HipWhisper(config.preset("test-d128"), device=0, max_batch=1)
    assert "no HIP device" in str(ei.value)


def test_create_rejects_unsupported_config_before_touching_the_device():
    cfg = config.preset("test-d128")
    cfg.d_model = 192  # head dim != 64
    with pytest.raises(hip.HipError):
        hip.HipWhisper(cfg, device=0, max_batch=1)
    cfg = config.preset("test-d128")
    with pytest.raises(hip.HipError):
        hip.HipWhisper(cfg, device=0, max_batch=0)


def test_struct_layouts_match_the_header():
    assert C.sizeof(hip.NhConfig) == 9 * 4
    assert C.sizeof(hip.NhTokens) == 8 * 4
    assert C.sizeof(hip.NhDecodeResult) == 24
    assert C.sizeof(hip.NhTimings) == 40


def test_integration_md_binds_every_declared_function():
    """INTEGRATION.md's `norma-hip-sys` block is the binding a maintainer would paste: it must declare every function of
    include/norma_hip.h (17 of 30 were listed, the batched entry point among the missing)."""
    import re
    with open(os.path.join(common.ROOT, "INTEGRATION.md")) as f:
        md = f.read()
    block = md[md.index('extern "C" {'):]
    block = block[:block.index("```")]
    bound = set(re.findall(r"pub fn (nh_[a-z_0-9]+)", block))
    assert bound == set(hip.declared_symbols()), (set(hip.declared_symbols()) - bound, bound - set(hip.declared_symbols()))
