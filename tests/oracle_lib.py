"""ctypes binding of oracle/libaix_oracle.so — TEST INFRASTRUCTURE ONLY (the checker).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class Mphf(C.Structure):
    _fields_ = [("n", C.c_uint64), ("D", C.c_uint64), ("seed", C.c_uint64), ("B", C.c_uint64),
                ("W", C.c_uint64), ("R", C.c_uint64), ("words", u64p), ("ranks", u64p)]


class Index23(C.Structure):
    _fields_ = [("f", Mphf), ("n", C.c_uint64), ("checker", u64p), ("tf", u32p)]


class Index13(C.Structure):
    _fields_ = [("f", Mphf), ("tf", u64p)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "libaix_oracle.so")
        src = os.path.join(ORACLE_DIR, "aix_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libaix_oracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.aixo_mphf_lookup.restype = C.c_uint64
        L.aixo_mphf_lookup.argtypes = [C.POINTER(Mphf), C.c_char_p, C.c_uint64]
        L.aixo_encode23.restype = C.c_uint64
        L.aixo_encode13.restype = C.c_uint32
        L.aixo_revdna23.restype = C.c_uint64
        L.aixo_revdna23.argtypes = [C.c_uint64]
        L.aixo_revdna13.restype = C.c_uint32
        L.aixo_revdna13.argtypes = [C.c_uint32]
        for name in ("aixo_tf23", "aixo_tf13"):
            getattr(L, name).restype = C.c_uint32
        for name in ("aixo_kid23", "aixo_strand23", "aixo_total23", "aixo_hash23", "aixo_total13",
                     "aixo_rc_refx86", "aixo_rc_true"):
            getattr(L, name).restype = C.c_uint64
        L.aixo_rc_refx86.argtypes = [C.c_uint64, C.c_int]
        L.aixo_rc_true.argtypes = [C.c_uint64, C.c_int]
        L.aixo_count_distinct.restype = C.c_int64
        L.aixo_count_distinct.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_uint64,
                                          C.POINTER(u64p), C.POINTER(u64p)]
        L.aixo_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


class OracleMphf:
    def __init__(self, path):
        self.s = Mphf()
        rc = lib().aixo_mphf_load(path.encode(), C.byref(self.s))
        if rc:
            raise OSError(f"aixo_mphf_load({path}) = {rc}")

    def lookup(self, s: bytes) -> int:
        return lib().aixo_mphf_lookup(C.byref(self.s), s, len(s))

    def __del__(self):
        try:
            lib().aixo_mphf_free(C.byref(self.s))
        except Exception:
            pass


class OracleIndex23:
    def __init__(self, pf, tf_bin, kmers_bin):
        self.s = Index23()
        rc = lib().aixo_index23_load(pf.encode(), tf_bin.encode(), kmers_bin.encode(), C.byref(self.s))
        if rc:
            raise OSError(f"aixo_index23_load = {rc}")
        self.n = self.s.n

    @classmethod
    def from_prefix(cls, prefix):
        return cls(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin")

    def checker(self):
        return np.ctypeslib.as_array(self.s.checker, shape=(self.n,)).copy()

    def tf_array(self):
        return np.ctypeslib.as_array(self.s.tf, shape=(self.n,)).copy()

    def tf(self, s: bytes):
        return lib().aixo_tf23(C.byref(self.s), s, C.c_uint64(len(s)))

    def kid(self, s):
        return lib().aixo_kid23(C.byref(self.s), s, C.c_uint64(len(s)))

    def strand(self, s):
        return lib().aixo_strand23(C.byref(self.s), s, C.c_uint64(len(s)))

    def total(self, s):
        return lib().aixo_total23(C.byref(self.s), s, C.c_uint64(len(s)))

    def both(self, s):
        a, b = C.c_uint32(), C.c_uint32()
        lib().aixo_both23(C.byref(self.s), s, C.c_uint64(len(s)), C.byref(a), C.byref(b))
        return a.value, b.value

    def hash(self, s):
        return lib().aixo_hash23(C.byref(self.s), s, C.c_uint64(len(s)))

    def tf_batch(self, kmers: np.ndarray, threads: int = 1) -> np.ndarray:
        """kmers: contiguous uint8 array of N*23 bytes."""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8).reshape(-1)
        n = kmers.shape[0] // 23
        out = np.empty(n, dtype=np.uint32)
        if threads > 1:
            lib().aixo_tf23_batch_mt(C.byref(self.s), _p(kmers, C.c_char_p), C.c_uint64(n), _p(out, u32p), threads)
        else:
            lib().aixo_tf23_batch(C.byref(self.s), _p(kmers, C.c_char_p), C.c_uint64(n), _p(out, u32p))
        return out

    def hash_batch(self, kmers: np.ndarray) -> np.ndarray:
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8).reshape(-1)
        n = kmers.shape[0] // 23
        out = np.empty(n, dtype=np.uint64)
        lib().aixo_hash23_batch(C.byref(self.s), _p(kmers, C.c_char_p), C.c_uint64(n), _p(out, u64p))
        return out

    def info_batch(self, kmers: np.ndarray):
        """(kid u64, strand u8, total u64, fwd u32, rc u32) of every 23-mer of a contiguous N*23 byte array."""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8).reshape(-1)
        n = kmers.shape[0] // 23
        kid, strand, total = np.empty(n, np.uint64), np.empty(n, np.uint8), np.empty(n, np.uint64)
        fwd, rc = np.empty(n, np.uint32), np.empty(n, np.uint32)
        lib().aixo_info23_batch(C.byref(self.s), _p(kmers, C.c_char_p), C.c_uint64(n), _p(kid, u64p), _p(strand, C.POINTER(C.c_uint8)), _p(total, u64p),
                                _p(fwd, u32p), _p(rc, u32p))
        return kid, strand, total, fwd, rc

    def coverage(self, seq: bytes, cutoff: int = 0) -> np.ndarray:
        out = np.zeros(max(0, len(seq) - 22), dtype=np.uint32)
        lib().aixo_coverage23(C.byref(self.s), seq, C.c_uint64(len(seq)), C.c_uint32(cutoff), _p(out, u32p))
        return out

    def count23_fixed(self, buf: bytes, is_fasta: bool, canon_mode: int) -> np.ndarray:
        out = np.zeros(self.n, dtype=np.uint32)
        lib().aixo_count23_fixed(C.byref(self.s), buf, C.c_uint64(len(buf)), int(is_fasta), canon_mode, _p(out, u32p))
        return out

    def positions(self, reads: bytes):
        tf = self.tf_array()
        indices = np.zeros(self.n + 1, dtype=np.uint64)
        lib().aixo_indices_prefix(_p(tf, u32p), C.c_uint64(self.n), _p(indices, u64p))
        pos = np.zeros(int(indices[-1]), dtype=np.uint64)
        lib().aixo_positions_fill(C.byref(self.s), reads, C.c_uint64(len(reads)), _p(indices, u64p), _p(pos, u64p))
        return indices, pos

    def __del__(self):
        try:
            lib().aixo_index23_free(C.byref(self.s))
        except Exception:
            pass


class OracleIndex13:
    def __init__(self, pf, tf: np.ndarray):
        """tf: uint64[4^13] in mphf order (kept alive here)."""
        self.m = OracleMphf(pf)
        self.tf_arr = np.ascontiguousarray(tf, dtype=np.uint64)
        assert self.tf_arr.shape[0] == 4 ** 13
        self.s = Index13()
        self.s.f = self.m.s
        self.s.tf = _p(self.tf_arr, u64p)

    def tf(self, s: bytes):
        return lib().aixo_tf13(C.byref(self.s), s, C.c_uint64(len(s)))

    def total(self, s):
        return lib().aixo_total13(C.byref(self.s), s, C.c_uint64(len(s)))

    def both(self, s):
        a, b = C.c_uint64(), C.c_uint64()
        lib().aixo_both13(C.byref(self.s), s, C.c_uint64(len(s)), C.byref(a), C.byref(b))
        return a.value, b.value

    def tf_batch(self, kmers: np.ndarray, threads: int = 1) -> np.ndarray:
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8).reshape(-1)
        n = kmers.shape[0] // 13
        out = np.empty(n, dtype=np.uint32)
        if threads > 1:
            lib().aixo_tf13_batch_mt(C.byref(self.s), _p(kmers, C.c_char_p), C.c_uint64(n), _p(out, u32p), threads)
        else:
            lib().aixo_tf13_batch(C.byref(self.s), _p(kmers, C.c_char_p), C.c_uint64(n), _p(out, u32p))
        return out

    def coverage(self, seq: bytes, cutoff: int = 0) -> np.ndarray:
        out = np.zeros(max(0, len(seq) - 12), dtype=np.uint32)
        lib().aixo_coverage13(C.byref(self.s), seq, C.c_uint64(len(seq)), C.c_uint32(cutoff), _p(out, u32p))
        return out


def jenkins(s: bytes, seed: int):
    out = (C.c_uint64 * 3)()
    lib().aixo_jenkins64(s, C.c_uint64(len(s)), C.c_uint64(seed), out)
    return [out[0], out[1], out[2]]


def count13(mphf: OracleMphf, buf: bytes, fmt: int = -1, threads: int = 1) -> np.ndarray:
    counts = np.zeros(4 ** 13, dtype=np.uint64)
    if threads > 1:
        lib().aixo_count13_mt(C.byref(mphf.s), buf, C.c_uint64(len(buf)), fmt, _p(counts, u64p), threads)
    else:
        lib().aixo_count13(C.byref(mphf.s), buf, C.c_uint64(len(buf)), fmt, _p(counts, u64p))
    return counts


def positions13(mphf: OracleMphf, tf64: np.ndarray, reads: bytes):
    """N3 restatement (compute_aindex13.cpp, one thread) with tf as u64[4^13]: (indices u64[4^13 + 1], positions u64[sum tf])."""
    tf64 = np.ascontiguousarray(tf64, dtype=np.uint64)
    assert tf64.shape[0] == 4 ** 13
    indices = np.zeros(4 ** 13 + 1, dtype=np.uint64)
    lib().aixo_indices_prefix64(_p(tf64, u64p), C.c_uint64(4 ** 13), _p(indices, u64p))
    pos = np.zeros(int(indices[-1]), dtype=np.uint64)
    lib().aixo_positions_fill13(C.byref(mphf.s), reads, C.c_uint64(len(reads)), _p(indices, u64p), _p(pos, u64p))
    return indices, pos


def count_distinct(fasta: bytes, k: int, canon_mode: int, min_count: int = 1):
    kp, cp = u64p(), u64p()
    m = lib().aixo_count_distinct(fasta, len(fasta), k, canon_mode, min_count, C.byref(kp), C.byref(cp))
    keys = np.ctypeslib.as_array(kp, shape=(max(m, 1),))[:m].copy()
    cnts = np.ctypeslib.as_array(cp, shape=(max(m, 1),))[:m].copy()
    lib().aixo_free(kp)
    lib().aixo_free(cp)
    return keys, cnts


def index_scatter(mphf: OracleMphf, keys: np.ndarray, tfs: np.ndarray):
    keys = np.ascontiguousarray(keys, dtype=np.uint8).reshape(-1)
    n = keys.shape[0] // 23
    tfs = np.ascontiguousarray(tfs, dtype=np.uint32)
    checker = np.zeros(n, dtype=np.uint64)
    tf = np.zeros(n, dtype=np.uint32)
    rc = lib().aixo_index_scatter(C.byref(mphf.s), _p(keys, C.c_char_p), _p(tfs, u32p), C.c_uint64(n),
                                  _p(checker, u64p), _p(tf, u32p))
    return rc, checker, tf
