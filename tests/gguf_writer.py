"""Test helper: write a GGUF v3 file the way candle's `tensor-tools quantize --quantization q8_0` lays out the
lmz/candle-whisper checkpoints: 2-D `*.weight` matrices whose row length is a multiple of 32 as Q8_0, everything else F32.
Returns the dequantised tensors (what a loader must reconstruct)."""
import struct

import numpy as np

GGML_F32, GGML_F16, GGML_Q8_0 = 0, 1, 8


def q8_0_quantize(a: np.ndarray):
    """ggml quantize_row_q8_0_reference: per block of 32, d = amax / 127 (stored as f16), q = round(x / d)."""
    x = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 32)
    amax = np.abs(x).max(axis=1)
    d = (amax / np.float32(127.0)).astype(np.float32)
    inv = np.where(d > 0, np.float32(1.0) / np.where(d > 0, d, 1), 0).astype(np.float32)
    q = np.rint(x * inv[:, None]).astype(np.int8)
    d16 = d.astype(np.float16)
    return d16, q


def q8_0_dequantize(d16: np.ndarray, q: np.ndarray, shape):
    return (d16.astype(np.float32)[:, None] * q.astype(np.float32)).reshape(shape)


def _s(b: str) -> bytes:
    e = b.encode()
    return struct.pack("<Q", len(e)) + e


def write_gguf(path, tensors, alignment=32, version=3):
    """tensors: iterable of (name, float array).  Returns {name: dequantised float32 array}."""
    infos, blobs, out = [], [], {}
    off = 0
    for name, a in tensors:
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim == 2 and name.endswith(".weight") and a.shape[1] % 32 == 0:
            d16, q = q8_0_quantize(a)
            blob = b"".join(d16[i].tobytes() + q[i].tobytes() for i in range(len(d16)))
            ty = GGML_Q8_0
            out[name] = q8_0_dequantize(d16, q, a.shape)
        else:
            blob, ty = a.tobytes(), GGML_F32
            out[name] = a
        infos.append((name, a.shape, ty, off))
        pad = (-len(blob)) % alignment
        blobs.append(blob + b"\0" * pad)
        off += len(blob) + pad
    kv = [("general.architecture", 8, _s("whisper")), ("general.alignment", 4, struct.pack("<I", alignment)),
          ("general.quantization_version", 4, struct.pack("<I", 2)),
          ("norma.test.array", 9, struct.pack("<IQ", 5, 3) + struct.pack("<3i", 1, 2, 3))]   # an array value the reader must skip
    hdr = struct.pack("<IIQQ", 0x46554747, version, len(infos), len(kv))
    for k, t, v in kv:
        hdr += _s(k) + struct.pack("<I", t) + v
    for name, shape, ty, o in infos:
        hdr += _s(name) + struct.pack("<I", len(shape))
        for dim in reversed(shape):              # ggml order: contiguous dimension first
            hdr += struct.pack("<Q", dim)
        hdr += struct.pack("<IQ", ty, o)
    hdr += b"\0" * ((-len(hdr)) % alignment)
    with open(path, "wb") as f:
        f.write(hdr)
        for b in blobs:
            f.write(b)
    return out
