"""Loader for the mel filterbank tables (the reference's `include_bytes!` at
`src/models/whisper/monolingual.rs:217-228`)."""
import os

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


class MelBinsError(ValueError):
    """Mirrors `whisper::Error::MelBins` (`src/models/whisper/mod.rs:79-80`)."""


def mel_filters(n_mel: int) -> np.ndarray:
    if n_mel not in (80, 128):
        raise MelBinsError(f"Unexpected number of mel bins (num_mel_bins), got: {n_mel}")
    raw = np.fromfile(os.path.join(_HERE, f"mel_filters_{n_mel}.f32le"), dtype="<f4")
    return np.ascontiguousarray(raw.reshape(n_mel, 201).astype(np.float32))
