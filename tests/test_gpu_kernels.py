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
    assert "324 launches" in out and ": 0 differing dwords" in out, out[-500:]


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


_STRICT_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import common
from norma_amd import config, hip, synth
out = {}
for name, B, script_seed in (("test-d128", 5, 3), ("tiny.en", 3, 8)):
    cfg = config.preset(name); tk = common.tokens_for(name)
    script = common.transcript_script(tk, n_segments=3, words_per_segment=6, seed=script_seed)
    for label, over in (("scripted", common.scripted_overrides(cfg, tk, script)), ("random", None)):
        hm = common.build_hip(cfg, tk, overrides=over, max_batch=B)
        hm.logmel([synth.synth_pcm(k) for k in range(B)]); hm.encode()
        res = hm.decode_greedy(0 if label == "scripted" else 60)
        out[name + "/" + label] = [[r["tokens"], r["avg_logprob"].hex(), r["no_speech_prob"].hex()] for r in res]
        hm.close()
out["lib"] = hip.LIB_PATH
print("RESULT " + json.dumps(out))
"""


def test_strict_memory_model_build_gives_the_same_tokens_and_logprobs():
    """k_decode.hip's one cross-workgroup hand-off (logit_step_kernel: 8 workgroups per sequence -> last arriver) has two
    spellings: the gfx950 ISA-level default (sc1 stores, s_waitcnt vmcnt(0), relaxed ticket) and -DNH_STRICT_MEMORY_MODEL
    (RELEASE ticket + ACQUIRE fence).  The Makefile builds both on every build; this runs the strict library
    (NORMA_HIP_LIB) in a child process and compares tokens and log-probs with the default library bit for bit."""
    import json
    import sys
    from norma_amd import hip
    if not os.path.exists(hip.STRICT_LIB_PATH):
        pytest.fail(f"{hip.STRICT_LIB_PATH} not built (make -C norma_amd/csrc)")
    outs = []
    for lib in (None, hip.STRICT_LIB_PATH):
        env = dict(os.environ)
        env.pop("NORMA_HIP_LIB", None)
        if lib:
            env["NORMA_HIP_LIB"] = lib
        p = subprocess.run([sys.executable, "-c", _STRICT_CHILD, common.ROOT], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
        outs.append(json.loads(line[len("RESULT "):]))
    default, strict = outs
    assert strict.pop("lib").endswith("libnorma_hip_strict.so") and default.pop("lib").endswith("libnorma_hip.so")
    assert default == strict
    assert any(len(r[0]) > 20 for r in default["tiny.en/scripted"])
