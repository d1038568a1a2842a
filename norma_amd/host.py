"""Python face of the C++ host layer (norma_amd/csrc/norma_host.hpp through include/norma_host.h).

Names and behaviour follow the reference's plugin API so that tests read like the reference's own
usage (README.md:16-52, src/models/mod.rs:13-56, src/models/whisper/monolingual.rs:113-174):

    definition = monolingual.Definition(ModelType.DistilLargeEnV3, SelectedDevice.Rocm(0))
    model = definition.blocking_try_to_model(...)      # ModelDefinition::blocking_try_to_model
    segments = model.transcribe(pcm, final_chunk=True)  # Model::transcribe

The transcribe policy (buffering, 30 s windows, seek by timestamps, final_chunk) runs in C++ on top of
the C ABI; nothing here computes anything.
"""
import ctypes as C
import enum
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import assets_io, hip
from .config import Config


@dataclass(frozen=True)
class SelectedDevice:
    """`enum SelectedDevice { Cpu, Cuda(usize), Metal }` (src/models/mod.rs:36-43) + `Rocm(usize)`."""
    kind: int = 0
    ordinal: int = 0

    @staticmethod
    def Cpu():
        return SelectedDevice(0, 0)

    @staticmethod
    def Cuda(n: int):
        return SelectedDevice(1, n)

    @staticmethod
    def Metal():
        return SelectedDevice(2, 0)

    @staticmethod
    def Rocm(n: int):
        return SelectedDevice(3, n)


class ModelType(enum.IntEnum):
    """monolingual::ModelType (monolingual.rs:32-46) and multilingual::ModelType (multilingual.rs:47-57)."""
    TinyEn = 0
    BaseEn = 1
    SmallEn = 2
    MediumEn = 3
    DistilMediumEn = 4
    DistilLargeEnV2 = 5
    DistilLargeEnV3 = 6
    QuantizedTinyEn = 7     # q8_0 GGUF checkpoint (monolingual.rs), files *-tiny-en*
    QuantizedTiny = 8       # q8_0 GGUF checkpoint (multilingual.rs:49), files *-tiny*
    # multilingual::ModelType (multilingual.rs:47-57): language=None detects the language, a fixed "<|xx|>" is MultiAsMono
    Tiny = 9
    Base = 10
    Small = 11
    Medium = 12
    Large = 13
    LargeV2 = 14
    LargeV3 = 15


QUANTIZED_EXT = {ModelType.QuantizedTinyEn: "tiny-en", ModelType.QuantizedTiny: "tiny"}


class WhisperError(RuntimeError):
    """whisper::Error / TranscriberError (src/models/whisper/mod.rs:64-84, model.rs:44-46)."""


_bound = False


def _lib():
    global _bound
    L = hip.load_library()
    if not _bound:
        vp = C.c_void_p
        L.nm_definition_new.restype = vp
        L.nm_definition_new.argtypes = [C.c_int, C.c_int, C.c_size_t]
        L.nm_definition_free.argtypes = [vp]
        L.nm_definition_set_responsiveness.argtypes = [vp, C.c_uint64]
        L.nm_definition_max_chunk_len.restype = C.c_size_t
        L.nm_definition_max_chunk_len.argtypes = [vp]
        L.nm_definition_data_buffer_size.restype = C.c_size_t
        L.nm_definition_data_buffer_size.argtypes = [vp]
        L.nm_definition_set_data_buffer_size.argtypes = [vp, C.c_size_t]
        L.nm_tensors_new.restype = vp
        L.nm_tensors_add.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.c_int, vp]
        L.nm_tensors_free.argtypes = [vp]
        L.nm_definition_blocking_try_to_model.restype = vp
        L.nm_definition_blocking_try_to_model.argtypes = [vp, C.POINTER(hip.NhConfig), C.POINTER(hip.NhTokens),
                                                          C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_float), C.c_int,
                                                          vp, C.c_char_p, C.c_int]
        L.nm_definition_blocking_try_to_model_from_dir.restype = vp
        L.nm_definition_blocking_try_to_model_from_dir.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_char_p,
                                                                   C.c_int, C.c_char_p, C.c_int]
        L.nm_model_last_text.argtypes = [vp, C.c_char_p, C.c_int]
        L.nm_model_enable_language_detection.argtypes = [vp, C.POINTER(C.c_int32), C.c_int]
        L.nm_model_language_token.argtypes = [vp]
        L.nm_gguf_list.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.nm_model_set_temperature_fallback.argtypes = [vp, C.c_int, C.c_uint64]
        L.nm_model_free.argtypes = [vp]
        L.nm_model_transcribe.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                          C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.c_char_p, C.c_int]
        L.nm_model_last_result.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                           C.POINTER(C.c_int)]
        _bound = True
    return L


class Model:
    """whisper::Model: `Data = f32`, `SAMPLE_RATE = 16000` (model.rs:48-52)."""
    SAMPLE_RATE = 16000

    def __init__(self, handle, ctx_len: int):
        self._h = handle
        self._ctx_len = ctx_len
        self.buffered_samples = 0

    def close(self):
        if self._h:
            _lib().nm_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transcribe(self, data: np.ndarray, final_chunk: bool) -> List[List[int]]:
        """Model::transcribe (model.rs:55-159): returns the token ids of each emitted segment."""
        data = np.ascontiguousarray(data, dtype=np.float32)
        cap = 64 * (self._ctx_len + 2)
        out = np.zeros(cap, dtype=np.int32)
        n_out, buffered = C.c_int(0), C.c_size_t(0)
        err = C.create_string_buffer(512)
        rc = _lib().nm_model_transcribe(self._h, data.ctypes.data_as(C.POINTER(C.c_float)), len(data), int(final_chunk),
                                        out.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(n_out), C.byref(buffered),
                                        err, 512)
        self.buffered_samples = int(buffered.value)
        if rc:
            raise WhisperError(err.value.decode())
        segs, cur = [], []
        for t in out[:n_out.value].tolist():
            if t == -1:
                segs.append(cur)
                cur = []
            else:
                cur.append(t)
        return segs

    def enable_language_detection(self, lang_tokens: Sequence[int]):
        """multilingual LanguageState::Detect: tokens of `Language::iter()` (languages.rs:7-107) in order."""
        a = np.ascontiguousarray(lang_tokens, dtype=np.int32)
        _lib().nm_model_enable_language_detection(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)), len(a))

    def set_temperature_fallback(self, enable: bool, seed: int = 0):
        """decode_with_fallback's sampled attempts (model.rs:175-188) on/off; draws follow the seeded sampling contract."""
        _lib().nm_model_set_temperature_fallback(self._h, int(enable), int(seed))

    @property
    def language_token(self) -> int:
        return int(_lib().nm_model_language_token(self._h))

    def last_text(self) -> str:
        """Concatenated text of the last transcribe call (needs a model loaded with a tokenizer)."""
        n = _lib().nm_model_last_text(self._h, None, 0)
        buf = C.create_string_buffer(n + 1)
        _lib().nm_model_last_text(self._h, buf, n + 1)
        return buf.value.decode("utf-8", errors="replace")

    def last_result(self) -> dict:
        a, n, f, k = C.c_double(0), C.c_double(0), C.c_int(0), C.c_int(0)
        _lib().nm_model_last_result(self._h, C.byref(a), C.byref(n), C.byref(f), C.byref(k))
        return dict(avg_logprob=a.value, no_speech_prob=n.value, needed_fallback=bool(f.value), n_tokens=k.value)


def gguf_list(path: str):
    """Tensors of a GGUF file as the C++ reader sees them: [(name, ggml_type, shape, sum of dequantised values)]."""
    _lib()
    buf = C.create_string_buffer(1 << 20)
    n = _lib().nm_gguf_list(path.encode(), buf, len(buf))
    if n < 0:
        raise WhisperError(buf.value.decode())
    out = []
    for line in buf.value.decode().splitlines():
        name, ty, shape, s = line.split(" ")
        out.append((name, int(ty), tuple(int(x) for x in shape.split("x")), float(s)))
    return out


class Definition:
    """whisper::monolingual::Definition (monolingual.rs:113-174)."""

    def __init__(self, model: ModelType, device: SelectedDevice):
        self.model, self.device = model, device
        self._h = _lib().nm_definition_new(int(model), device.kind, device.ordinal)

    def __del__(self):
        try:
            if self._h:
                _lib().nm_definition_free(self._h)
                self._h = None
        except Exception:
            pass

    def set_responsiveness(self, period_ms: int):
        if _lib().nm_definition_set_responsiveness(self._h, period_ms):
            raise WhisperError("The respnsivness must be over 1 second and under 30")

    def set_data_buffer_size(self, n: int):
        _lib().nm_definition_set_data_buffer_size(self._h, n)

    @property
    def max_chunk_len(self) -> int:
        return int(_lib().nm_definition_max_chunk_len(self._h))

    @property
    def data_buffer_size(self) -> int:
        return int(_lib().nm_definition_data_buffer_size(self._h))

    def blocking_try_to_model_from_dir(self, path: str, language: Optional[str] = "<|en|>", translate: bool = False,
                                       ctx_len: int = 448) -> Model:
        """Load config.json / tokenizer.json / model.safetensors from a local directory (the files the reference
        fetches through hf-hub, monolingual.rs:323-345).  language=None: multilingual Definition, the language is
        detected (multilingual.rs:463-466)."""
        import json
        import os
        ext = QUANTIZED_EXT.get(self.model)   # quantised checkpoints: config-{ext}.json, tokenizer-{ext}.json, model-{ext}-q80.gguf
        with open(os.path.join(path, f"config-{ext}.json" if ext else "config.json")) as f:
            n_mel = json.load(f)["num_mel_bins"]
        filt = np.ascontiguousarray(assets_io.mel_filters(n_mel), dtype=np.float32)
        err = C.create_string_buffer(512)
        h = _lib().nm_definition_blocking_try_to_model_from_dir(self._h, path.encode(), filt.ctypes.data_as(C.POINTER(C.c_float)),
                                                                filt.shape[0], (language or "").encode(), int(translate), err, 512)
        if not h:
            raise WhisperError(err.value.decode())
        return Model(C.c_void_p(h), ctx_len)

    def blocking_try_to_model(self, cfg: Config, tokens, lang: int, task: int,
                              weights: Iterable[Tuple[str, np.ndarray]], mel_filters: Optional[np.ndarray] = None) -> Model:
        """ModelDefinition::blocking_try_to_model (monolingual.rs:320-451) with local pieces instead of hf-hub."""
        L = _lib()
        c = hip.NhConfig(cfg.num_mel_bins, cfg.max_source_positions, cfg.d_model, cfg.encoder_attention_heads,
                         cfg.encoder_layers, cfg.vocab_size, cfg.max_target_positions, cfg.decoder_attention_heads,
                         cfg.decoder_layers)
        tk = hip.NhTokens(tokens.sot, tokens.eot, lang, task, tokens.no_speech, tokens.no_timestamps, tokens.zero_sec,
                          tokens.one_sec)
        sup = np.asarray(cfg.suppress_tokens, dtype=np.int32)
        filt = np.ascontiguousarray(assets_io.mel_filters(cfg.num_mel_bins) if mel_filters is None else mel_filters,
                                    dtype=np.float32)
        tl = L.nm_tensors_new()
        keep = []
        try:
            for name, arr in weights:
                arr = np.ascontiguousarray(arr)
                dt = hip.NH_DTYPE_F16 if arr.dtype == np.float16 else hip.NH_DTYPE_F32
                if dt == hip.NH_DTYPE_F32:
                    arr = np.ascontiguousarray(arr, dtype=np.float32)
                keep.append(arr)
                shape = (C.c_int64 * arr.ndim)(*arr.shape)
                L.nm_tensors_add(tl, name.encode(), dt, shape, arr.ndim, arr.ctypes.data_as(C.c_void_p))
            err = C.create_string_buffer(512)
            h = L.nm_definition_blocking_try_to_model(self._h, C.byref(c), C.byref(tk),
                                                      sup.ctypes.data_as(C.POINTER(C.c_int32)), len(sup),
                                                      filt.ctypes.data_as(C.POINTER(C.c_float)), filt.shape[0], tl, err, 512)
            if not h:
                raise WhisperError(err.value.decode())
            return Model(C.c_void_p(h), cfg.max_target_positions)
        finally:
            L.nm_tensors_free(tl)
