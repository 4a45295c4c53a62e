"""MWHC perfect-hash builder (product side; native C++ in csrc/aix_builder.hip).

Replaces the reference's `compute_mphf_seq` binary: same input (a list of keys), bit-identical `.pf`.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from ._lib import check, lib, vp


def _take(p: vp, n: C.c_uint64) -> bytes:
    try:
        return C.string_at(p, n.value)
    finally:
        lib().aix_free(p)


def build_pf_fixed(keys, key_len: int) -> bytes:
    """keys: bytes or uint8 array of n*key_len ASCII bytes (e.g. 23-mers)."""
    a = np.frombuffer(keys, dtype=np.uint8) if isinstance(keys, (bytes, bytearray)) else np.ascontiguousarray(keys, dtype=np.uint8).reshape(-1)
    assert a.shape[0] % key_len == 0
    p, n = vp(), C.c_uint64()
    check(lib().aix_pf_build(a.ctypes.data_as(vp), a.shape[0] // key_len, key_len, C.byref(p), C.byref(n)), "aix_pf_build")
    return _take(p, n)


def build_pf(keys: Sequence) -> bytes:
    """keys: sequence of str/bytes of any lengths (one per line of the reference's keys file)."""
    bs = [k.encode() if isinstance(k, str) else bytes(k) for k in keys]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    data = np.frombuffer(b"".join(bs), dtype=np.uint8)
    p, n = vp(), C.c_uint64()
    check(lib().aix_pf_build_ragged(data.ctypes.data_as(vp), offs.ctypes.data_as(vp), len(bs), C.byref(p), C.byref(n)),
          "aix_pf_build_ragged")
    return _take(p, n)


def build_pf_codes(codes: np.ndarray, k: int) -> bytes:
    """keys given as 2-bit codes (first base most significant); hashed as their ASCII strings."""
    c = np.ascontiguousarray(codes, dtype=np.uint64)
    p, n = vp(), C.c_uint64()
    check(lib().aix_pf_build_codes(c.ctypes.data_as(vp), c.shape[0], k, C.byref(p), C.byref(n)), "aix_pf_build_codes")
    return _take(p, n)


def build_pf_codes_t(keys_t, k: int) -> bytes:
    """GPU construction (parallel peeling) for keys already in HBM (int64/uint64 tensor of 2-bit codes). The result is a
    valid emphf `.pf` for the reference, not byte-identical to `build_pf_codes` (see aix_builder_gpu.hip)."""
    import torch
    p, n = vp(), C.c_uint64()
    with torch.cuda.device(keys_t.device):
        check(lib().aix_pf_build_codes_dev(vp(keys_t.data_ptr()), keys_t.numel(), k, keys_t.device.index,
                                           vp(torch.cuda.current_stream().cuda_stream), C.byref(p), C.byref(n)), "aix_pf_build_codes_dev")
    return _take(p, n)


def build_all_13mers_pf(path: str | None = None) -> bytes:
    """The MPHF over all 4^13 13-mers in 2-bit order (generate_all_13mers + compute_mphf_seq)."""
    p, n = vp(), C.c_uint64()
    check(lib().aix_pf_build_all_13mers(C.byref(p), C.byref(n)), "aix_pf_build_all_13mers")
    img = _take(p, n)
    if path:
        with open(path, "wb") as f:
            f.write(img)
    return img


def all_13mers_pf_path(path: str | None = None) -> str:
    """Path of the all-13-mers `.pf` (22 MB, the 13-mer mode's fixed perfect hash; sha256 pinned in tests/golden/pf13.json),
    built with build_all_13mers_pf on first use. Default location: <repo>/data/all_13mers.pf or $AIX_PF13."""
    import os
    if path is None:
        path = os.environ.get("AIX_PF13") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "all_13mers.pf")
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + f".tmp{os.getpid()}"
        build_all_13mers_pf(tmp)
        os.replace(tmp, path)                       # several ranks may race here: each writes its own file, the rename is atomic
    return path
