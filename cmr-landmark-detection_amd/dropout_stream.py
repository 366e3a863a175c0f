"""Host mirror of the counter-based dropout stream in csrc/rvip_common.h (hash32 / dropout_key /
dropout_keep).  The device regenerates the keep-mask of a Dropout layer in forward and backward from
(seed, optimizer step, dropout layer id, element index); this NumPy twin lets a caller reproduce the exact
mask (parity tests feed it to the CPU oracle).  Bit-for-bit identical by construction: integer arithmetic."""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _hash32(x):
    x = np.asarray(x, np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & _M32
    x ^= x >> np.uint64(16)
    return x


def dropout_key(seed, step, layer_id):
    inner = (np.uint64(step) * np.uint64(0x9E3779B9) + np.uint64(layer_id) * np.uint64(0x85EBCA6B) + np.uint64(0x27d4eb2f)) & _M32
    return _hash32(np.uint64(seed) ^ _hash32(inner))


def dropout_thr(rate):
    keep = np.float32(1.0) - np.float32(rate)
    t = int(np.float32(keep * np.float32(65536.0) + np.float32(0.5)))
    return min(t, 65536)


def keep_mask(shape, rate, seed, step, layer_id):
    """uint8 keep-mask (1 = keep) of a tensor of ``shape`` (flattened row-major = NHWC element index)."""
    n = int(np.prod(shape))
    key = dropout_key(seed, step, layer_id)
    pairs = np.arange((n + 1) // 2, dtype=np.uint64)
    h = _hash32((pairs & _M32) ^ key)
    bits = np.empty(2 * pairs.size, np.uint64)
    bits[0::2] = h & np.uint64(0xFFFF)
    bits[1::2] = h >> np.uint64(16)
    return (bits[:n] < np.uint64(dropout_thr(rate))).astype(np.uint8).reshape(shape)
