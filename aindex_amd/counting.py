"""Distinct k-mer counting on the GPU — the K1 row (replaces the reference's `kmer_counter`).

window codes (hand-written HIP kernel, csrc/aix_kernels.hip:k_window_codes) -> rocPRIM radix sort + run-length inside
the library (aix_count_distinct / aix_count_distinct_dev; buffers beyond 2^31 windows piece by piece). Output = the (k-mer, count) set the reference writes to
./output.txt (count_kmers.cpp:362-382), as arrays sorted by key; the reference's own order is
"count descending, ties unspecified", so parity is on the set.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, vp

INVALID = 0xFFFFFFFFFFFFFFFF


def normalize(buf: bytes, fmt: int = _lib.FMT_AUTO, fasta_mode: int = 1) -> bytes:
    """Host-side record normalisation to PLAIN form (see aix_normalize_reads)."""
    a = np.frombuffer(buf, dtype=np.uint8)
    out = np.empty(a.shape[0] + 2, dtype=np.uint8)
    n = C.c_uint64()
    check(lib().aix_normalize_reads(a.ctypes.data_as(vp), a.shape[0], fmt, fasta_mode, out.ctypes.data_as(vp), C.byref(n)),
          "aix_normalize_reads")
    return out[: n.value].tobytes()


def normalize_t(raw_t, fmt: int, fasta_mode: int = 1):
    """Device-side record normalisation: uint8 tensor of raw FASTA/FASTQ bytes in HBM -> PLAIN uint8 tensor."""
    import torch
    n = raw_t.numel()
    out = torch.empty(n + 16, dtype=torch.uint8, device=raw_t.device)
    m = C.c_uint64()
    with torch.cuda.device(raw_t.device):
        check(lib().aix_normalize_reads_dev(vp(raw_t.data_ptr()), n, fmt, fasta_mode, vp(out.data_ptr()), C.byref(m),
                                            vp(torch.cuda.current_stream().cuda_stream)), "aix_normalize_reads_dev")
    return out[: m.value]


def window_codes_t(plain_t, k: int, canon_mode: int):
    """int64 tensor (u64 bit patterns) of len-k+1 canonical window codes; -1 (= ~0) marks invalid windows."""
    import torch
    n = plain_t.numel()
    out = torch.empty(max(n - k + 1, 0), dtype=torch.int64, device=plain_t.device)
    if out.numel():
        with torch.cuda.device(plain_t.device):
            check(lib().aix_window_codes_dev(vp(plain_t.data_ptr()), n, k, canon_mode, vp(out.data_ptr()),
                                             vp(torch.cuda.current_stream().cuda_stream)), "aix_window_codes_dev")
    return out


def count_distinct_t(plain_t, k: int, canon_mode: int = _lib.CANON_TRUE_RC, min_count: int = 1):
    """(keys int64 tensor sorted ascending, counts int64 tensor) for a PLAIN buffer already in HBM — aix_count_distinct_dev:
    window-code kernel + rocPRIM radix sort / run-length inside the library; torch only owns the result tensors."""
    import torch
    dev = plain_t.device
    res, n = vp(), C.c_uint64()
    with torch.cuda.device(dev):
        stream = vp(torch.cuda.current_stream().cuda_stream)
        check(lib().aix_count_distinct_dev(vp(plain_t.data_ptr()), plain_t.numel(), k, canon_mode, min_count, dev.index, stream, C.byref(res)),
              "aix_count_distinct_dev")
        try:
            check(lib().aix_distinct_size(res, C.byref(n)), "aix_distinct_size")
            keys = torch.empty(n.value, dtype=torch.int64, device=dev)
            counts = torch.empty(n.value, dtype=torch.int64, device=dev)
            check(lib().aix_distinct_copy_dev(res, vp(keys.data_ptr()) if n.value else None, vp(counts.data_ptr()) if n.value else None, stream),
                  "aix_distinct_copy_dev")
        finally:
            lib().aix_distinct_free(res)
    return keys, counts


def merge_counts_t(keys_t, counts_t, min_count: int = 1):
    """Device tensors of (key, count) pairs with repeated keys -> (keys ascending, summed counts >= min_count), all int64
    (u64 bit patterns); aix_merge_counts_dev: rocPRIM sort + reduce-by-key inside the library."""
    import torch
    dev = keys_t.device
    keys_t, counts_t = keys_t.contiguous(), counts_t.contiguous()
    res, n = vp(), C.c_uint64()
    with torch.cuda.device(dev):
        stream = vp(torch.cuda.current_stream().cuda_stream)
        check(lib().aix_merge_counts_dev(vp(keys_t.data_ptr()) if keys_t.numel() else None, vp(counts_t.data_ptr()) if keys_t.numel() else None,
                                         keys_t.numel(), min_count, dev.index, stream, C.byref(res)), "aix_merge_counts_dev")
        try:
            check(lib().aix_distinct_size(res, C.byref(n)), "aix_distinct_size")
            keys = torch.empty(n.value, dtype=torch.int64, device=dev)
            counts = torch.empty(n.value, dtype=torch.int64, device=dev)
            check(lib().aix_distinct_copy_dev(res, vp(keys.data_ptr()) if n.value else None, vp(counts.data_ptr()) if n.value else None, stream),
                  "aix_distinct_copy_dev")
        finally:
            lib().aix_distinct_free(res)
    return keys, counts


def _take_distinct(kp, cp, n):
    try:
        m = n.value
        if m == 0 or not kp.value or not cp.value:
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64)
        keys = np.ctypeslib.as_array(C.cast(kp, C.POINTER(C.c_uint64)), shape=(m,)).copy()
        counts = np.ctypeslib.as_array(C.cast(cp, C.POINTER(C.c_uint64)), shape=(m,)).copy()
    finally:
        lib().aix_free(kp)
        lib().aix_free(cp)
    return keys, counts


def count_distinct_file(path: str, k: int, canon_mode: int = _lib.CANON_TRUE_RC, min_count: int = 1, fmt: int = _lib.FMT_FASTA, device: int = 0):
    """kmer_counter on a FILE, streamed (aix_count_distinct_file): parts of the file cross the link while earlier ones are normalised and
    appended to the current piece; pieces are counted as they fill and their sorted sets merged. Returns (keys, counts, stats dict)."""
    import os
    kp, cp, n = vp(), vp(), C.c_uint64()
    st = _lib.IngestStats()
    check(lib().aix_count_distinct_file(os.fsencode(path), fmt, k, canon_mode, min_count, device, C.byref(kp), C.byref(cp), C.byref(n), C.byref(st)),
          f"aix_count_distinct_file({path})")
    keys, counts = _take_distinct(kp, cp, n)
    return keys, counts, st.as_dict()


def merge_runs_t(keys_t, counts_t, run_offsets, min_count: int = 1):
    """Device tensors holding sorted, repeat-free runs back to back (run r = [run_offsets[r], run_offsets[r + 1])) -> (keys ascending,
    summed counts >= min_count), int64 tensors (u64 bit patterns). aix_merge_runs_dev: a tree of two-way merges, no sort."""
    import torch
    dev = keys_t.device
    keys_t, counts_t = keys_t.contiguous(), counts_t.contiguous()
    offs = np.ascontiguousarray(run_offsets, dtype=np.uint64)
    res, n = vp(), C.c_uint64()
    with torch.cuda.device(dev):
        stream = vp(torch.cuda.current_stream().cuda_stream)
        check(lib().aix_merge_runs_dev(vp(keys_t.data_ptr()) if keys_t.numel() else None, vp(counts_t.data_ptr()) if keys_t.numel() else None,
                                       offs.ctypes.data_as(vp), offs.shape[0] - 1, min_count, dev.index, stream, C.byref(res)), "aix_merge_runs_dev")
        try:
            check(lib().aix_distinct_size(res, C.byref(n)), "aix_distinct_size")
            keys = torch.empty(n.value, dtype=torch.int64, device=dev)
            counts = torch.empty(n.value, dtype=torch.int64, device=dev)
            check(lib().aix_distinct_copy_dev(res, vp(keys.data_ptr()) if n.value else None, vp(counts.data_ptr()) if n.value else None, stream),
                  "aix_distinct_copy_dev")
        finally:
            lib().aix_distinct_free(res)
    return keys, counts


def count_distinct(buf: bytes, k: int, canon_mode: int = _lib.CANON_TRUE_RC, min_count: int = 1, fmt: int = _lib.FMT_FASTA,
                   device: int = 0):
    """kmer_counter replacement for a host buffer, entirely behind the C ABI (aix_count_distinct: HIP window kernel +
    rocPRIM sort / run-length). Returns (keys uint64, counts uint64) sorted by key."""
    a = np.frombuffer(buf, dtype=np.uint8)
    kp, cp, n = vp(), vp(), C.c_uint64()
    check(lib().aix_count_distinct(a.ctypes.data_as(vp), a.shape[0], fmt, k, canon_mode, min_count, device, C.byref(kp), C.byref(cp), C.byref(n)),
          "aix_count_distinct")
    return _take_distinct(kp, cp, n)
