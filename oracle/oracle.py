"""ctypes wrapper around oracle/libwhisper_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing
under norma_amd/ does.  See whisper_oracle.c for what the oracle restates and why parity is
"unpinned" (no reference-side golden vectors exist for this path).
"""
import ctypes as C
import os
import subprocess

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # parked, not spinning, between parallel regions

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwhisper_oracle.so")

USE_KV_CACHE = 1


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("n_mel", "n_audio_ctx", "d", "n_head", "n_enc", "n_vocab", "n_text_ctx", "n_dec")]


class _Tok(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("sot", "eot", "lang", "task", "no_speech", "no_timestamps", "zero_sec", "one_sec")]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "whisper_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def default_threads() -> int:
    env = os.environ.get("NORMA_ORACLE_THREADS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def set_num_threads(n: int):
    lib().wo_set_num_threads(int(n))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        dp = C.POINTER(C.c_double)
        L.wo_create.restype = C.c_void_p
        L.wo_create.argtypes = [C.POINTER(_Cfg)]
        L.wo_free.argtypes = [C.c_void_p]
        L.wo_set_tensor.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_long]
        L.wo_set_tokens.argtypes = [C.c_void_p, C.POINTER(_Tok), ip, C.c_int]
        L.wo_get_mask.argtypes = [C.c_void_p, C.c_int, fp]
        L.wo_mel_frames.restype = C.c_long
        L.wo_mel_frames.argtypes = [C.c_long]
        L.wo_pcm_to_mel.argtypes = [C.c_int, fp, C.c_long, fp, fp]
        L.wo_encoder_forward.argtypes = [C.c_void_p, fp, C.c_int, C.c_long, fp]
        L.wo_reset_kv_cache.argtypes = [C.c_void_p]
        L.wo_decoder_forward.argtypes = [C.c_void_p, ip, C.c_int, fp, C.c_int, C.c_int, fp]
        L.wo_final_linear.argtypes = [C.c_void_p, fp, C.c_int, fp]
        L.wo_apply_rules.argtypes = [C.c_void_p, fp, ip, C.c_int, C.c_int]
        L.wo_argmax_total.argtypes = [fp, C.c_int]
        L.wo_decode.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, C.c_int, ip, dp, dp, fp]
        L.wo_transcribe.argtypes = [C.c_void_p, fp, fp, C.POINTER(C.c_long), C.c_int, C.c_int, C.c_int,
                                    ip, C.c_int, ip, dp, dp]
        L.wo_detect_language.argtypes = [C.c_void_p, fp, C.c_int, ip, C.c_int, fp]
        L.wo_set_language.argtypes = [C.c_void_p, C.c_int]
        L.wo_num_threads.restype = C.c_int
        L.wo_set_num_threads.argtypes = [C.c_int]
        up = C.POINTER(C.c_uint32)
        L.wo_philox.argtypes = [up, up, up]
        L.wo_sexp.argtypes = [C.c_float]; L.wo_sexp.restype = C.c_float
        L.wo_sample_token.argtypes = [fp, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.wo_sample_token.restype = C.c_int
        L.wo_decode_t.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, C.c_int, ip, dp, dp, fp, C.c_double, C.c_uint64, C.c_int, C.c_int]
        L.wo_decode_t.restype = C.c_int
        L.wo_set_sampling.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        # never oversubscribe: GPU boxes expose every host core but grant a 16-core share
        L.wo_set_num_threads(default_threads())
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def mel_frames(n_samples: int) -> int:
    return int(lib().wo_mel_frames(n_samples))


def pcm_to_mel(pcm: np.ndarray, filters: np.ndarray) -> np.ndarray:
    """candle `audio::pcm_to_mel` (called at src/models/whisper/model.rs:74): -> [n_mel][n_len]."""
    pcm = np.ascontiguousarray(pcm, dtype=np.float32)
    filters = np.ascontiguousarray(filters, dtype=np.float32)
    n_mel = filters.shape[0]
    n_len = mel_frames(len(pcm))
    out = np.zeros((n_mel, n_len), dtype=np.float32)
    lib().wo_pcm_to_mel(n_mel, _f(pcm), len(pcm), _f(filters), _f(out))
    return out


def num_threads() -> int:
    return int(lib().wo_num_threads())


class OracleModel:
    """Batch-1 f32 Whisper with norma's decode policy.  `cfg` is norma_amd.config.Config,
    `tokens` a norma_amd.vocab.SpecialTokens, `lang`/`task` token ids (lang < 0: none)."""

    def __init__(self, cfg, tokens, lang: int, task: int, weights=None):
        self.cfg = cfg
        c = _Cfg(cfg.num_mel_bins, cfg.max_source_positions, cfg.d_model, cfg.encoder_attention_heads,
                 cfg.encoder_layers, cfg.vocab_size, cfg.max_target_positions, cfg.decoder_layers)
        self._h = lib().wo_create(C.byref(c))
        self.tok = _Tok(tokens.sot, tokens.eot, lang, task, tokens.no_speech, tokens.no_timestamps,
                        tokens.zero_sec, tokens.one_sec)
        sup = np.asarray(cfg.suppress_tokens, dtype=np.int32)
        lib().wo_set_tokens(self._h, C.byref(self.tok), _i(sup), len(sup))
        if weights is not None:
            for name, arr in weights:
                self.set_tensor(name, arr)

    def close(self):
        if self._h:
            lib().wo_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tensor(self, name: str, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        rc = lib().wo_set_tensor(self._h, name.encode(), _f(arr), arr.size)
        if rc < 0:
            raise ValueError(f"oracle: set_tensor({name}) failed rc={rc}")

    def mask(self, which: int) -> np.ndarray:
        out = np.zeros(self.cfg.vocab_size, dtype=np.float32)
        lib().wo_get_mask(self._h, which, _f(out))
        return out

    def encoder_forward(self, mel: np.ndarray) -> np.ndarray:
        """mel [n_mel][>=frames] (only the first min(3000, n) frames are used, model.rs:88)."""
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        frames = min(3000, mel.shape[1])
        S = (frames + 2 - 3) // 2 + 1
        out = np.zeros((S, self.cfg.d_model), dtype=np.float32)
        lib().wo_encoder_forward(self._h, _f(mel), frames, mel.shape[1], _f(out))
        return out

    def reset_kv_cache(self):
        lib().wo_reset_kv_cache(self._h)

    def decoder_forward(self, tokens, xa: np.ndarray, flush: bool) -> np.ndarray:
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        xa = np.ascontiguousarray(xa, dtype=np.float32)
        out = np.zeros((len(tokens), self.cfg.d_model), dtype=np.float32)
        lib().wo_decoder_forward(self._h, _i(tokens), len(tokens), _f(xa), xa.shape[0], int(flush), _f(out))
        return out

    def final_linear(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.cfg.d_model)
        out = np.zeros((x.shape[0], self.cfg.vocab_size), dtype=np.float32)
        lib().wo_final_linear(self._h, _f(x), x.shape[0], _f(out))
        return out

    def apply_rules(self, probs: np.ndarray, tokens, last_timestamp: int) -> np.ndarray:
        p = np.ascontiguousarray(probs, dtype=np.float32).copy()
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        lib().wo_apply_rules(self._h, _f(p), _i(t), len(t), last_timestamp)
        return p

    def detect_language(self, xa: np.ndarray, lang_tokens):
        """Model::detect_language (model.rs:194-210) -> (token id, probabilities over lang_tokens)."""
        xa = np.ascontiguousarray(xa, dtype=np.float32)
        lt = np.ascontiguousarray(lang_tokens, dtype=np.int32)
        probs = np.zeros(len(lt), dtype=np.float32)
        tok = lib().wo_detect_language(self._h, _f(xa), xa.shape[0], _i(lt), len(lt), _f(probs))
        return int(tok), probs

    def set_language(self, lang_token: int):
        lib().wo_set_language(self._h, int(lang_token))

    def set_sampling(self, enable_fallback: bool, seed: int = 0):
        """decode_with_fallback's sampled attempts (model.rs:175-188) on/off inside transcribe, and their seed."""
        lib().wo_set_sampling(self._h, int(enable_fallback), int(seed))

    def decode(self, xa: np.ndarray, use_kv_cache: bool = True, max_new_tokens: int = 0, want_steps=False,
               temperature: float = 0.0, seed: int = 0, clip: int = 0, attempt: int = 0):
        """Model::decode (model.rs:279-389); temperature > 0 samples under the seeded contract (whisper_oracle.c)."""
        xa = np.ascontiguousarray(xa, dtype=np.float32)
        toks = np.zeros(self.cfg.max_target_positions + 2, dtype=np.int32)
        alp = C.c_double(0)
        nsp = C.c_double(0)
        steps = np.zeros((self.cfg.max_target_positions, 4), dtype=np.float32) if want_steps else None
        n = lib().wo_decode_t(self._h, _f(xa), xa.shape[0], USE_KV_CACHE if use_kv_cache else 0,
                              max_new_tokens, _i(toks), C.byref(alp), C.byref(nsp),
                              _f(steps) if want_steps else None, float(temperature), int(seed), int(clip), int(attempt))
        res = dict(tokens=toks[:n].tolist(), avg_logprob=alp.value, no_speech_prob=nsp.value)
        if want_steps:
            res["steps"] = steps
        return res

    def transcribe(self, pcm: np.ndarray, filters: np.ndarray, final_chunk: bool, buf=None,
                   use_kv_cache: bool = True, max_new_tokens: int = 0):
        """Model::transcribe (model.rs:55-159) on `buf + pcm`; returns (segments, new_buf, info)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if buf is not None and len(buf):
            pcm = np.concatenate([np.ascontiguousarray(buf, dtype=np.float32), pcm])
        work = pcm.copy()
        blen = C.c_long(len(work))
        filters = np.ascontiguousarray(filters, dtype=np.float32)
        cap = 64 * (self.cfg.max_target_positions + 2)
        out = np.zeros(cap, dtype=np.int32)
        ns = C.c_int(0)
        alp = C.c_double(0)
        nsp = C.c_double(0)
        n = lib().wo_transcribe(self._h, _f(filters), _f(work), C.byref(blen), int(final_chunk),
                                USE_KV_CACHE if use_kv_cache else 0, max_new_tokens, _i(out), cap,
                                C.byref(ns), C.byref(alp), C.byref(nsp))
        if n < 0:
            raise RuntimeError("oracle: transcribe output overflow")
        segs, cur = [], []
        for t in out[:n].tolist():
            if t == -1:
                segs.append(cur)
                cur = []
            else:
                cur.append(t)
        return segs, work[:blen.value].copy(), dict(n_slices=ns.value, avg_logprob=alp.value,
                                                    no_speech_prob=nsp.value)


def philox(ctr, key):
    """philox4x32-10 (the sampling contract's generator): 4 + 2 uint32 in, 4 uint32 out."""
    c = (C.c_uint32 * 4)(*[int(v) & 0xFFFFFFFF for v in ctr])
    k = (C.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in key])
    o = (C.c_uint32 * 4)()
    lib().wo_philox(c, k, o)
    return [int(v) for v in o]


def sexp(y: float) -> float:
    return float(lib().wo_sexp(float(y)))


def sample_token(q: np.ndarray, temperature: float, seed: int, clip: int, step: int, attempt: int) -> int:
    """Draw from softmax(q / t) of the rule-masked probabilities q under the seeded contract; -1: everything masked."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    inv_t = np.float32(1.0) / np.float32(temperature)
    return int(lib().wo_sample_token(_f(q), len(q), float(inv_t), int(seed), int(clip), int(step), int(attempt)))
