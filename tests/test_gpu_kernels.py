"""GPU: kernel-level bit-exactness screens that sit below the C ABI (tools/*.hip, built by __graft_entry__.build()):
the persistent ping-pong GEMM against the independent 128 x 128 kernel, and the fused LayerNorm GEMV against the
stand-alone LayerNorm + plain GEMV.  Both pairs share their per-element arithmetic order, so any difference is a
synchronisation or indexing bug (a half-tile read before its LDS-DMA landed shows up as a differing tile)."""
import os
import subprocess

import pytest

import common

pytestmark = pytest.mark.gpu
BIN = os.path.join(common.ROOT, "tools", "bin")


def _run(name, *args):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (python -c 'import __graft_entry__ as g; g.build()')")
    p = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


def test_pingpong_gemm_is_bit_identical_to_the_128_tile_kernel_over_many_launches():
    out = _run("gemm_check", "12")
    assert "288 launches" in out and ": 0 differing dwords" in out, out[-500:]


def test_fused_layernorm_gemv_is_bit_identical_to_layernorm_then_gemv():
    out = _run("lnfuse_check")
    lines = [l for l in out.splitlines() if "outputs differ" in l]
    assert len(lines) >= 20 and all(": 0 of " in l for l in lines), out[-1500:]
