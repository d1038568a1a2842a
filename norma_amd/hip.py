"""ctypes binding of libnorma_hip.so (include/norma_hip.h) -- the only way Python reaches the HIP path.

There is deliberately no CPU fallback here: if the shared library is missing or no MI355X is
visible, construction fails loudly (the reference's `SelectedDevice::Cuda(n)` fails the same way
when candle cannot open the device, `src/models/mod.rs:47-55`).
"""
import ctypes as C
import os
import re
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .config import Config

_HERE = os.path.dirname(os.path.abspath(__file__))
# NORMA_HIP_LIB selects another BUILD of the same HIP library (libnorma_hip_strict.so: -DNH_STRICT_MEMORY_MODEL); there is no
# other implementation to select
LIB_PATH = os.environ.get("NORMA_HIP_LIB") or os.path.join(_HERE, "libnorma_hip.so")
STRICT_LIB_PATH = os.path.join(_HERE, "libnorma_hip_strict.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "norma_hip.h")

N_SAMPLES = 480000
N_FRAMES = 3000
NH_DTYPE_F32, NH_DTYPE_F16 = 0, 1
NH_OPT_DECODE_GRAPHS, NH_OPT_FUSE_DECODE_LAYERNORM, NH_OPT_DECODER_LAYER_LIMIT, NH_OPT_ABSORBED_XATTN = 0, 1, 2, 3
# NH_SAMPLE_* of include/norma_hip.h (the types of src/dtype.rs)
SAMPLE_DTYPES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int8): 2, np.dtype(np.int16): 3,
                 np.dtype(np.int32): 4, np.dtype(np.int64): 5, np.dtype(np.uint8): 6, np.dtype(np.uint16): 7,
                 np.dtype(np.uint32): 8, np.dtype(np.uint64): 9}


class HipError(RuntimeError):
    """A non-zero status from the C ABI (message = nh_last_error)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"norma_hip status {code}: {msg}")
        self.code = code


class NhConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_mel_bins", "max_source_positions", "d_model", "encoder_attention_heads", "encoder_layers",
        "vocab_size", "max_target_positions", "decoder_attention_heads", "decoder_layers")]


class NhTokens(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("sot", "eot", "lang", "task", "no_speech", "no_timestamps", "zero_sec", "one_sec")]


class NhDecodeResult(C.Structure):
    _fields_ = [("n_tokens", C.c_int32), ("no_speech_exit", C.c_int32), ("avg_logprob", C.c_double),
                ("no_speech_prob", C.c_double)]


class NhTimings(C.Structure):
    _fields_ = [("mel_ms", C.c_float), ("encoder_ms", C.c_float), ("cross_kv_ms", C.c_float),
                ("decode_ms", C.c_float), ("decode_steps", C.c_int32), ("gemm_ms", C.c_float),
                ("gemm_launches", C.c_int32), ("gemm_flops", C.c_double)]


def declared_symbols() -> List[str]:
    """Every function include/norma_hip.h declares (used by the CPU-side ABI test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(nh_[a-z_0-9]+)\s*\(", text)))


_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  norma_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.nh_create.argtypes = [C.c_int, C.POINTER(NhConfig), C.c_int, C.POINTER(vp)]
    L.nh_create_shared.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.nh_destroy.argtypes = [vp]
    L.nh_destroy.restype = None
    L.nh_last_error.argtypes = [vp]
    L.nh_last_error.restype = C.c_char_p
    L.nh_device_count.argtypes = []
    L.nh_load_tensor.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.c_int, vp]
    L.nh_set_mel_filters.argtypes = [vp, fp, C.c_int]
    L.nh_set_tokens.argtypes = [vp, C.POINTER(NhTokens), ip, C.c_int]
    L.nh_missing_tensors.argtypes = [vp]
    L.nh_logmel.argtypes = [vp, fp, ip, C.c_int64, C.c_int]
    L.nh_logmel_device.argtypes = [vp, vp, ip, C.c_int64, C.c_int]
    L.nh_logmel_rows.argtypes = [vp, fp, ip, C.c_int64, C.c_int, C.c_int]
    L.nh_logmel_device_rows.argtypes = [vp, vp, ip, C.c_int64, C.c_int, C.c_int]
    L.nh_encode_rows.argtypes = [vp, C.c_int, C.c_int]
    L.nh_pool_begin.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.nh_pool_admit.argtypes = [vp, C.c_int, C.c_int, C.c_int32]
    L.nh_pool_admit_from.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int32]
    L.nh_pool_step.argtypes = [vp, C.c_int, ip]
    L.nh_pool_collect.argtypes = [vp, ip, C.c_int, ip, C.POINTER(NhDecodeResult)]
    L.nh_logmel_samples.argtypes = [vp, vp, C.c_int, ip, C.c_int64, C.c_int]
    L.nh_sample_size.argtypes = [C.c_int]
    L.nh_encode.argtypes = [vp]
    L.nh_decode_greedy.argtypes = [vp, ip, C.POINTER(NhDecodeResult), C.c_int]
    L.nh_decode_sampled.argtypes = [vp, ip, C.POINTER(NhDecodeResult), C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32]
    L.nh_sample_rules.argtypes = [vp, fp, ip, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, ip]
    L.nh_transcribe_batch.argtypes = [vp, vp, ip, C.c_int64, C.c_int, ip, C.POINTER(NhDecodeResult), C.c_int]
    L.nh_detect_language.argtypes = [vp, ip, C.c_int, ip, fp]
    L.nh_set_languages.argtypes = [vp, ip]
    L.nh_reset.argtypes = [vp]
    L.nh_synchronize.argtypes = [vp]
    L.nh_get_mel.argtypes = [vp, C.c_int, fp]
    L.nh_set_mel.argtypes = [vp, fp, C.c_int]
    L.nh_encoder_output.argtypes = [vp, C.c_int, fp]
    L.nh_decoder_forward.argtypes = [vp, ip, C.c_int, fp]
    L.nh_final_linear.argtypes = [vp, fp, C.c_int, fp]
    L.nh_apply_rules.argtypes = [vp, fp, ip, C.c_int, C.c_int, fp, ip]
    L.nh_get_timings.argtypes = [vp, C.POINTER(NhTimings)]
    L.nh_set_profile_gemm.argtypes = [vp, C.c_int]
    L.nh_set_option.argtypes = [vp, C.c_int, C.c_int]
    _lib = L
    return L


def device_count() -> int:
    return int(load_library().nh_device_count())


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class HipWhisper:
    """One Whisper model resident on one MI355X (one `nh_ctx`).  Not thread-safe, like the
    reference's `Model: Send` (one transcriber thread calls it, `src/lib.rs:462-464`)."""

    def __init__(self, cfg: Config, device: int = 0, max_batch: int = 1, share_with: "Optional[HipWhisper]" = None):
        """share_with: another HipWhisper on the same device whose WEIGHTS (and mel filters) this context uses
        (nh_create_shared); tokens, streams, workspaces and K/V caches are its own."""
        self.L = load_library()
        self.cfg = cfg
        self.max_batch = max_batch
        h = C.c_void_p()
        if share_with is not None:
            rc = self.L.nh_create_shared(share_with._h, max_batch, C.byref(h))
        else:
            c = NhConfig(cfg.num_mel_bins, cfg.max_source_positions, cfg.d_model, cfg.encoder_attention_heads,
                         cfg.encoder_layers, cfg.vocab_size, cfg.max_target_positions,
                         cfg.decoder_attention_heads, cfg.decoder_layers)
            rc = self.L.nh_create(device, C.byref(c), max_batch, C.byref(h))
        if rc != 0:
            raise HipError(rc, self.L.nh_last_error(None).decode())
        self._h = h
        self.batch = 0
        self._experiment_switches()

    # -- lifetime ------------------------------------------------------------------------------
    def _experiment_switches(self):
        """NORMA_HIP_ABSORBED_XATTN=1: every context runs the NH_OPT_ABSORBED_XATTN numerics prototype (lets the whole parity suite
        be run against it: DESIGN.md 8 item 1)."""
        if os.environ.get("NORMA_HIP_ABSORBED_XATTN"):
            self.set_option(NH_OPT_ABSORBED_XATTN, int(os.environ["NORMA_HIP_ABSORBED_XATTN"]))

    def close(self):
        if getattr(self, "_h", None):
            self.L.nh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            raise HipError(rc, self.L.nh_last_error(self._h).decode())

    # -- model state ---------------------------------------------------------------------------
    def load_tensor(self, name: str, arr: np.ndarray):
        if arr.dtype == np.float16:
            dt = NH_DTYPE_F16
        else:
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            dt = NH_DTYPE_F32
        arr = np.ascontiguousarray(arr)
        shape = (C.c_int64 * arr.ndim)(*arr.shape)
        self._chk(self.L.nh_load_tensor(self._h, name.encode(), dt, shape, arr.ndim, arr.ctypes.data_as(C.c_void_p)))

    def load_weights(self, weights: Iterable[Tuple[str, np.ndarray]]):
        for name, arr in weights:
            self.load_tensor(name, arr)
        missing = self.L.nh_missing_tensors(self._h)
        if missing:
            raise HipError(3, f"{missing} tensors missing after load_weights")

    def set_mel_filters(self, filters: np.ndarray):
        f = np.ascontiguousarray(filters, dtype=np.float32)
        self._chk(self.L.nh_set_mel_filters(self._h, _fp(f), f.shape[0]))

    def set_tokens(self, tokens, lang: int, task: int, suppress: Optional[Sequence[int]] = None):
        tk = NhTokens(tokens.sot, tokens.eot, lang, task, tokens.no_speech, tokens.no_timestamps,
                      tokens.zero_sec, tokens.one_sec)
        sup = np.asarray(self.cfg.suppress_tokens if suppress is None else suppress, dtype=np.int32)
        self._chk(self.L.nh_set_tokens(self._h, C.byref(tk), _ip(sup), len(sup)))

    # -- hot path ------------------------------------------------------------------------------
    def logmel(self, clips: Sequence[np.ndarray]):
        """clips: list of f32 PCM arrays (each <= 480000 samples)."""
        B = len(clips)
        ns = np.array([len(c) for c in clips], dtype=np.int32)
        stride = int(max(ns.max(), 1))
        buf = np.zeros((B, stride), dtype=np.float32)
        for b, c in enumerate(clips):
            buf[b, :len(c)] = c
        self._chk(self.L.nh_logmel(self._h, _fp(buf), _ip(ns), stride, B))
        self.batch = B

    def logmel_array(self, pcm: np.ndarray, n_samples: Optional[Sequence[int]] = None):
        """pcm: contiguous f32 [batch][stride] in HOST memory, handed to nh_logmel as is (no staging copy here)."""
        assert pcm.dtype == np.float32 and pcm.ndim == 2 and pcm.flags.c_contiguous
        B, stride = pcm.shape
        ns = np.full(B, stride, dtype=np.int32) if n_samples is None else np.asarray(n_samples, dtype=np.int32)
        self._chk(self.L.nh_logmel(self._h, _fp(pcm), _ip(ns), stride, B))
        self.batch = B

    def logmel_samples(self, pcm: np.ndarray, n_samples: Optional[Sequence[int]] = None):
        """pcm: contiguous [batch][stride] array of a native capture type (src/dtype.rs: u8 .. f64); converted to f32 on the
        GPU with dasp_sample's formulas (nh_logmel_samples)."""
        assert pcm.ndim == 2 and pcm.flags.c_contiguous and pcm.dtype in SAMPLE_DTYPES
        B, stride = pcm.shape
        ns = np.full(B, stride, dtype=np.int32) if n_samples is None else np.asarray(n_samples, dtype=np.int32)
        self._chk(self.L.nh_logmel_samples(self._h, pcm.ctypes.data_as(C.c_void_p), SAMPLE_DTYPES[pcm.dtype], _ip(ns), stride, B))
        self.batch = B

    def logmel_device(self, pcm_dev_ptr: int, n_samples: Sequence[int], stride: int):
        """PCM already resident in HBM (device pointer, f32 [batch][stride])."""
        ns = np.asarray(n_samples, dtype=np.int32)
        self._chk(self.L.nh_logmel_device(self._h, C.c_void_p(pcm_dev_ptr), _ip(ns), stride, len(ns)))
        self.batch = len(ns)

    def logmel_device_rows(self, pcm_dev_ptr: int, n_samples: Sequence[int], stride: int, row0: int):
        """Clips for rows [row0, row0 + len(n_samples)) of the context (several encoder batches, one joint decode)."""
        ns = np.asarray(n_samples, dtype=np.int32)
        self._chk(self.L.nh_logmel_device_rows(self._h, C.c_void_p(pcm_dev_ptr), _ip(ns), stride, len(ns), row0))
        self.batch = row0 + len(ns) if row0 > 0 else len(ns)

    def logmel_array_rows(self, pcm: np.ndarray, row0: int):
        """pcm: contiguous f32 [batch][stride] in HOST memory -> rows [row0, row0 + batch)."""
        assert pcm.dtype == np.float32 and pcm.ndim == 2 and pcm.flags.c_contiguous
        B, stride = pcm.shape
        ns = np.full(B, stride, dtype=np.int32)
        self._chk(self.L.nh_logmel_rows(self._h, _fp(pcm), _ip(ns), stride, B, row0))
        self.batch = row0 + B if row0 > 0 else B

    def encode_rows(self, row0: int, batch: int):
        self._chk(self.L.nh_encode_rows(self._h, row0, batch))

    # ---- decode pool (include/norma_hip.h: nh_pool_*): rows [0, rows) decode at their own positions, the rows above stage
    # encoder output; norma_amd/pool.py holds the refill policy
    def pool_begin(self, rows: int, max_new_tokens: int = 0, per_clip_language: bool = False):
        self._chk(self.L.nh_pool_begin(self._h, rows, max_new_tokens, int(per_clip_language)))
        self.pool_rows = rows

    def pool_admit(self, src_row: int, dst_row: int, lang: int = -1):
        self._chk(self.L.nh_pool_admit(self._h, src_row, dst_row, lang))

    def pool_admit_from(self, enc: "HipWhisper", src_row: int, dst_row: int, lang: int = -1):
        """the clip encoded in row src_row of ANOTHER context of the same weight set joins this context's pool"""
        self._chk(self.L.nh_pool_admit_from(self._h, enc._h, src_row, dst_row, lang))

    def pool_step(self, n_steps: int) -> np.ndarray:
        """n_steps tokens for every busy row; returns the rows' flags (0 running, 1 finished, 2 no-speech exit, 3 empty)."""
        done = np.zeros(self.pool_rows, dtype=np.int32)
        self._chk(self.L.nh_pool_step(self._h, n_steps, _ip(done)))
        return done

    def pool_collect(self, rows: Sequence[int]) -> List[dict]:
        rows = np.asarray(rows, dtype=np.int32)
        toks = np.zeros((len(rows), self.cfg.max_target_positions), dtype=np.int32)
        res = (NhDecodeResult * len(rows))()
        self._chk(self.L.nh_pool_collect(self._h, _ip(rows), len(rows), _ip(toks), res))
        return [dict(tokens=toks[i, :res[i].n_tokens].tolist(), avg_logprob=res[i].avg_logprob,
                     no_speech_prob=res[i].no_speech_prob, no_speech_exit=bool(res[i].no_speech_exit)) for i in range(len(rows))]

    def get_mel(self, b: int, frames: int = N_FRAMES) -> np.ndarray:
        out = np.zeros((self.cfg.num_mel_bins, frames), dtype=np.float32)
        self._chk(self.L.nh_get_mel(self._h, b, _fp(out)))
        return out

    def set_mel(self, mel: np.ndarray):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        assert mel.ndim == 3 and mel.shape[1:] == (self.cfg.num_mel_bins, N_FRAMES)
        self._chk(self.L.nh_set_mel(self._h, _fp(mel), mel.shape[0]))
        self.batch = mel.shape[0]

    def encode(self):
        self._chk(self.L.nh_encode(self._h))

    def encoder_output(self, b: int, S: int = 1500) -> np.ndarray:
        out = np.zeros((S, self.cfg.d_model), dtype=np.float32)
        self._chk(self.L.nh_encoder_output(self._h, b, _fp(out)))
        return out

    def _results(self, toks: np.ndarray, res) -> List[dict]:
        out = []
        for b in range(self.batch):
            n = res[b].n_tokens
            out.append(dict(tokens=toks[b, :n].tolist(), avg_logprob=res[b].avg_logprob,
                            no_speech_prob=res[b].no_speech_prob, no_speech_exit=bool(res[b].no_speech_exit)))
        return out

    def decode_greedy(self, max_new_tokens: int = 0) -> List[dict]:
        B = self.batch
        toks = np.zeros((B, self.cfg.max_target_positions), dtype=np.int32)
        res = (NhDecodeResult * B)()
        self._chk(self.L.nh_decode_greedy(self._h, _ip(toks), res, max_new_tokens))
        return self._results(toks, res)

    def decode_sampled(self, temperature: float, seed: int, clip0: int = 0, attempt: int = 1,
                       max_new_tokens: int = 0) -> List[dict]:
        """Model::decode at t > 0 (model.rs:340-348) under the seeded sampling contract of include/norma_hip.h."""
        B = self.batch
        toks = np.zeros((B, self.cfg.max_target_positions), dtype=np.int32)
        res = (NhDecodeResult * B)()
        self._chk(self.L.nh_decode_sampled(self._h, _ip(toks), res, max_new_tokens, float(temperature), int(seed),
                                           int(clip0), int(attempt)))
        return self._results(toks, res)

    def transcribe_batch_device(self, pcm_dev_ptr: int, n_samples: Sequence[int], stride: int,
                                max_new_tokens: int = 0) -> List[dict]:
        """pcm already resident in HBM (device pointer, f32 [batch][stride])."""
        ns = np.asarray(n_samples, dtype=np.int32)
        B = len(ns)
        toks = np.zeros((B, self.cfg.max_target_positions), dtype=np.int32)
        res = (NhDecodeResult * B)()
        self._chk(self.L.nh_transcribe_batch(self._h, C.c_void_p(pcm_dev_ptr), _ip(ns), stride, B, _ip(toks), res,
                                             max_new_tokens))
        self.batch = B
        return self._results(toks, res)

    def detect_language(self, lang_tokens: Sequence[int], want_probs: bool = True):
        """Model::detect_language for every clip; the result also becomes the decode prompt's language token."""
        lt = np.ascontiguousarray(lang_tokens, dtype=np.int32)
        out = np.zeros(self.batch, dtype=np.int32)
        probs = np.zeros((self.batch, len(lt)), dtype=np.float32) if want_probs else None
        self._chk(self.L.nh_detect_language(self._h, _ip(lt), len(lt), _ip(out), _fp(probs) if want_probs else None))
        return out.tolist(), probs

    def set_languages(self, langs: Optional[Sequence[int]]):
        if langs is None:
            self._chk(self.L.nh_set_languages(self._h, None))
        else:
            a = np.ascontiguousarray(langs, dtype=np.int32)
            self._chk(self.L.nh_set_languages(self._h, _ip(a)))

    def reset(self):
        self._chk(self.L.nh_reset(self._h))

    def synchronize(self):
        self._chk(self.L.nh_synchronize(self._h))

    # -- fine-grained parity views ---------------------------------------------------------------
    def decoder_forward(self, tokens: np.ndarray) -> np.ndarray:
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        B, T = tokens.shape
        assert B == self.batch
        out = np.zeros((B, T, self.cfg.d_model), dtype=np.float32)
        self._chk(self.L.nh_decoder_forward(self._h, _ip(tokens), T, _fp(out)))
        return out

    def final_linear(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.cfg.d_model)
        out = np.zeros((x.shape[0], self.cfg.vocab_size), dtype=np.float32)
        self._chk(self.L.nh_final_linear(self._h, _fp(x), x.shape[0], _fp(out)))
        return out

    def apply_rules(self, probs: np.ndarray, tokens: Sequence[int], last_timestamp: int):
        p = np.ascontiguousarray(probs, dtype=np.float32)
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.zeros_like(p)
        am = C.c_int32(0)
        self._chk(self.L.nh_apply_rules(self._h, _fp(p), _ip(t), len(t), last_timestamp, _fp(out), C.byref(am)))
        return out, int(am.value)

    def sample_rules(self, probs: np.ndarray, tokens: Sequence[int], last_timestamp: int, temperature: float, seed: int,
                     clip: int = 0, attempt: int = 1) -> int:
        """The sampler alone: rules + one seeded draw on a soft-maxed probability vector (-1: everything masked)."""
        p = np.ascontiguousarray(probs, dtype=np.float32)
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        tok = C.c_int32(0)
        self._chk(self.L.nh_sample_rules(self._h, _fp(p), _ip(t), len(t), last_timestamp, float(temperature), int(seed),
                                         int(clip), int(attempt), C.byref(tok)))
        return int(tok.value)

    # -- instrumentation ---------------------------------------------------------------------------
    def set_profile_gemm(self, enable: bool):
        self._chk(self.L.nh_set_profile_gemm(self._h, int(enable)))

    def set_option(self, option: int, value: int):
        """A/B switches of include/norma_hip.h (NH_OPT_*): how the decode step is launched, never what it computes."""
        self._chk(self.L.nh_set_option(self._h, int(option), int(value)))

    def timings(self) -> dict:
        t = NhTimings()
        self._chk(self.L.nh_get_timings(self._h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in NhTimings._fields_}
