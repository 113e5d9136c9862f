"""Helpers shared by the parity tests: call the C ABI on host buffers."""
import ctypes

import numpy as np

from pysubstringsearch_amd import _ffi


def sa_gpu(data, device: int = 0) -> np.ndarray:
    """pss_sa_build (host pointers in, host pointers out) -> int32 array."""
    t = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else \
        np.ascontiguousarray(data, dtype=np.uint8)
    sa = np.full(t.size, -1, dtype=np.int32)
    buf = t if t.size else np.zeros(1, np.uint8)
    out = sa if sa.size else np.zeros(1, np.int32)
    rc = _ffi.lib.pss_sa_build(buf.ctypes.data, out.ctypes.data, t.size, device)
    _ffi.check(rc)
    return sa


def gen_corpus(kind: int, n: int, chunk_index: int = 0) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    _ffi.check(_ffi.lib.pss_gen_corpus(kind, out.ctypes.data, n, chunk_index))
    return out
