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


@pytest.mark.gpu
def test_native_sample_types_are_converted_like_dasp_sample():
    """src/dtype.rs + `Sample::to_sample::<f32>` (src/lib.rs:180,207) on the GPU: nh_logmel_samples on native samples must give
    bit for bit the log-mel of nh_logmel on the f32 samples a numpy restatement of dasp_sample's conversions produces
    (signed: s / 2^(bits-1); unsigned: (s - 2^(bits-1)) / 2^(bits-1); f64: round to nearest).  dasp_sample is not vendored
    in the reference, so the formulas are restated from its published behaviour: parity unpinned for this conversion."""
    import numpy as np
    import common
    from norma_amd import assets_io, config, synth
    cfg = config.preset("test-d128")
    tk = common.tokens_for("test-d128")
    from norma_amd import hip
    hm = hip.HipWhisper(cfg, device=0, max_batch=2)
    hm.set_mel_filters(assets_io.mel_filters(cfg.num_mel_bins))
    rng = np.random.default_rng(3)
    base = np.stack([synth.synth_pcm(0, 48000), synth.synth_pcm(1, 48000)]).astype(np.float64)
    ns = [48000, 31999]

    def check(native, as_f32):
        hm.logmel_samples(np.ascontiguousarray(native), ns)
        got = [hm.get_mel(b) for b in range(2)]
        hm.logmel_array(np.ascontiguousarray(as_f32, dtype=np.float32), ns)
        for b in range(2):
            assert np.array_equal(got[b], hm.get_mel(b)), native.dtype
    for dt, bits in ((np.int8, 8), (np.int16, 16), (np.int32, 32), (np.int64, 64)):
        half = 2.0 ** (bits - 1)
        v = np.clip(np.round(base * (half - 1)), -half, half - 1).astype(dt)
        if bits >= 32:                                  # values that do not fit an f32 mantissa: the rounding mode matters
            v = v + rng.integers(-1000, 1000, size=v.shape).astype(dt)
        check(v, v.astype(np.float32) / np.float32(half))
        udt = {8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}[bits]
        un = (v.astype(np.int64) + int(half)).astype(udt) if bits < 64 else (v.view(np.uint64) ^ np.uint64(1 << 63))
        check(un, v.astype(np.float32) / np.float32(half))     # unsigned sample == the signed one shifted by 2^(bits-1)
    f64 = base * 0.999 + 1e-9 * rng.standard_normal(base.shape)
    check(f64, f64.astype(np.float32))
    f32 = base.astype(np.float32)
    check(f32, f32)
    assert [hip.load_library().nh_sample_size(c) for c in range(11)] == [4, 8, 1, 2, 4, 8, 1, 2, 4, 8, 0]
    hm.close()
