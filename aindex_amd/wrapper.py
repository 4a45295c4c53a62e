"""AindexWrapper — drop-in counterpart of the reference's pybind11 class `aindex_cpp.AindexWrapper`
(src/python_wrapper.cpp:130-1316, bindings :1320-2135) for the tf / hash / batch-query path.

Same method names, argument meaning and return types; the work runs on the MI355X through
libaindex_hip.so. Differences are limited to failure behaviour: where the reference calls
std::terminate()/exit() (missing files: python_wrapper.cpp:265,413,1118) this class raises
FileNotFoundError / AixError instead of killing the interpreter.

Mode semantics preserved: get_tf_value(s) dispatch on `is_13mer_mode` — set by any
13-mer load and never cleared — not on the k-mer length.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .engine import Index

TOTAL_13MERS = 4 ** 13
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _enc(s) -> bytes:
    return s.encode("latin-1") if isinstance(s, str) else bytes(s)


_PYFAST = False


_PACK_THREADS = max(1, min(32, (os.cpu_count() or 2) // 2))      # worker threads of the list[str] / list[int] conversions


def _pyfast():
    """The optional CPython helper built next to the library (csrc/aix_pyfast.c); None when it is not there."""
    global _PYFAST
    if _PYFAST is False:
        _PYFAST = None
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_aix_pyfast.so")
        if os.path.exists(path):
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("_aix_pyfast", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _PYFAST = mod
            except Exception:                                  # noqa: BLE001 — host glue only, the Python path is equivalent
                _PYFAST = None
    return _PYFAST


class _Stage:
    """Pinned staging of the list surface, kept between calls: the packed query bytes on the way in, the uint32 answers on the way out
    (aix_host_alloc). A fresh 460 MB bytes object per call is page-faulted by the packing threads — that cost more than the packing and the
    lookup together — and pageable buffers cross the link at 2/3 of the pinned rate. Blocks grow geometrically; above AIX_STAGE_MAX_MB
    (default 4096) per direction, or while another thread is inside, the caller takes the unstaged path. Answers never depend on it."""

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self._blk = {}                                     # name -> (pointer, capacity in bytes, uint8 array over it)
        self._max = int(os.environ.get("AIX_STAGE_MAX_MB", "4096")) << 20

    def view(self, name: str, nbytes: int):
        """uint8 array of at least nbytes pinned bytes (the caller holds self.lock), or None."""
        import ctypes as C
        blk = self._blk.get(name)
        if blk is not None and blk[1] >= nbytes:
            return blk[2]
        if nbytes > self._max:
            return None
        cap = min(self._max, max(nbytes + nbytes // 2, 1 << 24))
        lib = _lib.lib()
        if blk is not None:
            del self._blk[name]
            lib.aix_host_free(blk[0])
        p = C.c_void_p()
        if lib.aix_host_alloc(cap, C.byref(p)) != 0 or not p.value:
            return None
        arr = np.ctypeslib.as_array((C.c_uint8 * cap).from_address(p.value))
        self._blk[name] = (p, cap, arr)
        return arr

    def close(self):
        with self.lock:
            blks, self._blk = self._blk, {}
        for p, _, _ in blks.values():
            _lib.lib().aix_host_free(p)

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001 — interpreter shutdown
            pass


class AindexWrapper:
    def __init__(self, device: int = 0):
        self._device = device
        self._stage = _Stage()
        self._ix23: Optional[Index] = None
        self._ix13: Optional[Index] = None
        self._is_13mer_mode = False
        self._tf13_host: Optional[np.ndarray] = None     # mmap of the 13-mer tf file (u64, mphf order)
        self._checker_host: Optional[np.ndarray] = None
        # read/write attributes of the reference class (python_wrapper.cpp:1820-1838)
        self.aindex_loaded = False
        self.n_reads = 0
        self.n_kmers = 0
        self.reads_size = 0
        self.max_tf = 0
        self._reads = None
        self._positions = None
        self._indices = None
        self._ridx_start = None

    # ---- loaders -----------------------------------------------------------------------------
    @staticmethod
    def _need(*paths):
        for p in paths:
            if not os.path.exists(p):
                raise FileNotFoundError(f"Required file not found: {p}")

    def load(self, hash_filename: str, tf_file: str, kmers_bin_filename: str, kmers_text_filename: str = ""):
        """python_wrapper.cpp:228-245 (kmers_text_filename is unused for k = 23, hash.cpp:388-423)."""
        self._need(hash_filename, tf_file, kmers_bin_filename)
        if self._ix23 is not None:
            self._ix23.close()
        self._ix23 = Index.open_23(hash_filename, tf_file, kmers_bin_filename, self._device)
        self._checker_host = None
        self.n_kmers = self._ix23.n

    def load_hash_file(self, hash_filename: str, tf_file: str, kmers_bin_filename: str, kmers_text_filename: str = ""):
        self.load(hash_filename, tf_file, kmers_bin_filename, kmers_text_filename)      # :247-259

    def load_from_prefix_23mer(self, prefix: str, reads_file: str = ""):
        """:1103-1132 — files prefix.pf / .tf.bin / .kmers.bin."""
        self.load_hash_file(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin", prefix + ".txt")
        if reads_file:
            self.load_reads(reads_file)

    def load_13mer_index(self, hash_file: str, tf_file: str):
        """:404-437 — tf file = u64[4^13] in mphf order (count_kmers13 output)."""
        self._need(hash_file, tf_file)
        if self._ix13 is not None:
            self._ix13.close()
        self._ix13 = Index.open_13(hash_file, tf_file, self._device)
        self._tf13_host = np.memmap(tf_file, dtype=np.uint64, mode="r", shape=(TOTAL_13MERS,))
        self._is_13mer_mode = True
        self.n_kmers = TOTAL_13MERS

    def load_from_prefix_13mer(self, prefix: str, reads_file: str = ""):
        self.load_13mer_index(prefix + ".pf", prefix + ".tf.bin")                       # :1162-1188
        if reads_file:
            self.load_reads(reads_file)

    # ---- reads + positions index (N2/N4 tier: host-side views over the reference's files) ---------
    def load_reads_index(self, index_file: str):
        """:261-279 — `.ridx` lines "rid\tstart\tend"; intervals are (start, end + 1)."""
        self._need(index_file)
        import ctypes as C
        from ._lib import lib, vp, check
        n, p = C.c_uint64(), vp()
        check(lib().aix_ridx_load(index_file.encode(), C.byref(n), C.byref(p)), "aix_ridx_load")      # parsed by the library (10^8 lines are no numpy.loadtxt job)
        try:
            a = np.frombuffer(C.string_at(p, 24 * n.value), dtype=np.uint64).reshape(-1, 3) if n.value else np.zeros((0, 3), np.uint64)
        finally:
            lib().aix_free(p)
        self._ridx_rid, self._ridx_start, self._ridx_end = a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy()
        self.n_reads = int(a.shape[0])
        # the reference scans its interval list linearly for every query (python_wrapper.cpp:66-74). When the intervals are
        # sorted and disjoint — what compute_reads writes — the first match of that scan is found by bisection instead.
        st, en = self._ridx_start, self._ridx_end
        self._ridx_sorted = bool(a.shape[0] == 0 or (np.all(en >= st) and np.all(st[1:] > en[:-1])))

    def load_reads(self, reads_file: str):
        """:281-322 — mmap the reads file and read the sibling `.ridx`."""
        self._need(reads_file)
        self._reads = np.memmap(reads_file, dtype=np.uint8, mode="r")
        self.reads_size = int(self._reads.shape[0])
        self.load_reads_index(reads_file[: reads_file.rfind(".")] + ".ridx")

    def load_reads_in_memory(self, reads_file: str):
        """:324-359 — the same as load_reads, but the bytes are read into a private buffer (`new char[length]` there)
        instead of being mapped: later changes to the file do not show through. An unreadable file leaves the wrapper
        without reads, as the reference's early `return` does (:337-340)."""
        if not os.path.isfile(reads_file):
            self._reads, self.reads_size = None, 0
            return
        self._reads = np.fromfile(reads_file, dtype=np.uint8)
        self.reads_size = int(self._reads.shape[0])
        self.load_reads_index(reads_file[: reads_file.rfind(".")] + ".ridx")

    def load_aindex(self, index_file: str, indices_file: str, max_tf: int = 0):
        """:361-402 — .index.bin (positions, 1-based, 0 = empty) and .indices.bin (n+1 offsets)."""
        self._need(index_file, indices_file)
        self._need23()
        self._positions = np.memmap(index_file, dtype=np.uint64, mode="r") if os.path.getsize(index_file) else np.zeros(0, np.uint64)
        self._indices = np.memmap(indices_file, dtype=np.uint64, mode="r")
        self.max_tf = max_tf
        self.aindex_loaded = True

    def load_aindex_from_prefix_23mer(self, prefix: str, max_tf: int = 0, reads_file: str = ""):
        self.load_aindex(prefix + ".index.bin", prefix + ".indices.bin", max_tf)       # :1134-1160
        if reads_file and getattr(self, "_reads", None) is None:
            self.load_reads(reads_file)

    def load_13mer_aindex(self, index_file: str, indices_file: str, ref_compat: bool = False):
        """:439-471. The reference maps only `.indices.bin` here and never sets `positions_13mer`, so its 13-mer position
        queries always answer []; the mirror maps BOTH files (the `.index.bin` our compute_aindex13 writes, N3) so that
        get_positions_13mer (:1070-1100) can answer what it is written to answer. A missing index file keeps the reference's
        behaviour (indices only, queries return []), and so does ref_compat=True (or AIX_REF_COMPAT=1): the positions file is
        not mapped and every get_positions_13mer answers [] as the reference's does."""
        self._need(indices_file)
        self._indices13 = np.memmap(indices_file, dtype=np.uint64, mode="r")
        self._positions13 = None
        if ref_compat or os.environ.get("AIX_REF_COMPAT") == "1":
            self.aindex_loaded = True
            return
        if index_file and os.path.isfile(index_file) and os.path.getsize(index_file):
            self._positions13 = np.memmap(index_file, dtype=np.uint64, mode="r")
        self.aindex_loaded = True

    def load_aindex_from_prefix_13mer(self, prefix: str, reads_file: str = "", ref_compat: bool = False):
        self.load_13mer_aindex(prefix + ".index.bin", prefix + ".indices.bin", ref_compat)

    def build_aindex(self, reads_file: str, prefix: Optional[str] = None):
        """compute_aindex replacement (A1/A2 on the GPU): returns (indices, positions) and, with a prefix,
        writes `<prefix>.indices.bin` / `<prefix>.index.bin` exactly like the reference (hash.hpp:470-486)."""
        reads = open(reads_file, "rb").read()
        if self._is_13mer_mode and self._ix13 is not None:              # N3: compute_aindex13 (forward strand, strict ACGT, u64 tf)
            indices, pos = self._ix13.positions_fill(reads)
        else:
            indices, pos = self._need23().positions_fill(reads)
        if prefix:
            pos.tofile(prefix + ".index.bin")
            indices.tofile(prefix + ".indices.bin")
        return indices, pos

    def get_read_by_rid(self, rid: int) -> str:
        if getattr(self, "_reads", None) is None or rid >= self.n_reads or rid < 0:     # :666-675
            return ""
        return bytes(self._reads[int(self._ridx_start[rid]):int(self._ridx_end[rid])]).decode("latin-1")

    def get_read(self, start: int, end: int, revcomp: bool = False) -> str:
        if getattr(self, "_reads", None) is None or start >= self.reads_size or end >= self.reads_size or start > end:
            return ""                                                                    # :677-698
        b = bytes(self._reads[start:end])
        if revcomp:
            b = b[::-1].translate(_COMP)
        return b.decode("latin-1")

    def _interval(self, pos: int):
        if not self.aindex_loaded or getattr(self, "_ridx_start", None) is None or self.n_reads == 0:
            return None
        # IntervalTree::query(pos, pos + 1) of python_wrapper.cpp:66-74: first interval in file order with
        # start <= pos + 1 and (end + 1) >= pos
        if getattr(self, "_ridx_sorted", False):
            i0 = int(np.searchsorted(self._ridx_end, np.uint64(max(pos - 1, 0)), side="left"))   # first interval with end + 1 >= pos
            return i0 if i0 < self.n_reads and int(self._ridx_start[i0]) <= pos + 1 else None
        hit = np.nonzero((self._ridx_start <= np.uint64(pos + 1)) & (self._ridx_end + np.uint64(1) >= np.uint64(pos)))[0]
        return int(hit[0]) if hit.shape[0] else None

    def get_rid(self, pos: int) -> int:
        i = self._interval(pos)                                                          # :757-772
        return int(self._ridx_rid[i]) if i is not None else 0

    def get_start(self, pos: int) -> int:
        i = self._interval(pos)                                                          # :774-789
        return int(self._ridx_start[i]) if i is not None else 0

    # ---- helpers -----------------------------------------------------------------------------
    def _need23(self) -> Index:
        if self._ix23 is None:
            raise RuntimeError("23-mer index not loaded (the reference dereferences a null hash_map here)")
        return self._ix23

    # ---- tf queries --------------------------------------------------------------------------
    def get_tf_values(self, kmers: List[str]) -> List[int]:
        """:653-664 — dispatch on is_13mer_mode."""
        if self._is_13mer_mode:
            return self.get_tf_values_13mer(kmers)
        return self.get_tf_values_23mer(kmers)

    def get_tf_value(self, kmer: str) -> int:
        return self.get_tf_values([kmer])[0]                                            # :644-651

    @staticmethod
    def _packed(kmers, k: int):
        """A batch that is ALREADY one buffer of N*k ASCII bytes — bytes / bytearray / memoryview, a numpy 'S<k>' or (N, k) uint8 array,
        or ONE str of N*k characters — as that buffer (no per-item Python work), else None. (A str of exactly k characters is one query,
        a longer one whose length is a multiple of k is the joined batch.)"""
        if isinstance(kmers, (bytes, bytearray, memoryview)):
            return kmers if len(kmers) % k == 0 else None
        if isinstance(kmers, str):
            if len(kmers) % k:
                return None
            fast = _pyfast()
            if fast is not None and hasattr(fast, "ascii_view"):
                return fast.ascii_view(kmers)                    # the str's own character buffer (None unless it is ASCII); the caller's str outlives the call
            return kmers.encode("ascii") if kmers.isascii() else None
        if isinstance(kmers, np.ndarray):
            if kmers.dtype.kind == "S" and kmers.dtype.itemsize == k:
                return np.ascontiguousarray(kmers).view(np.uint8).reshape(-1)
            if kmers.dtype == np.uint8 and (kmers.ndim == 1 or (kmers.ndim == 2 and kmers.shape[1] == k)) and kmers.size % k == 0:
                return np.ascontiguousarray(kmers).reshape(-1)
        return None

    @staticmethod
    def _to_list(a: np.ndarray) -> List[int]:
        """uint32 answers as the list[int] the reference returns; values up to 256 (nearly every term frequency) are filled in by worker
        threads from CPython's cached small ints (csrc/aix_pyfast.c: u32_list), ndarray.tolist() otherwise."""
        fast = _pyfast()
        if fast is not None and hasattr(fast, "u32_list") and a.dtype == np.uint32 and a.flags.c_contiguous:
            return fast.u32_list(a, a.shape[0], _PACK_THREADS)
        return a.tolist()

    @staticmethod
    def _join_fixed(kmers, k: int):
        """One bytes object of len(kmers)*k bytes when every item is a k-character str/bytes, else None."""
        fast = _pyfast()
        if fast is not None:
            flat = fast.join_fixed(kmers, k, _PACK_THREADS)      # C loop over the list (worker threads read the items' buffers): no intermediate str / bytes objects
            if flat is not None or not kmers:
                return flat
            # not all-ASCII k-character items: let the exact Python rules below decide (latin-1 characters are still one byte)
        try:
            if isinstance(kmers[0], str):
                flat = "".join(kmers).encode("latin-1")
            else:
                flat = b"".join(kmers)
        except (TypeError, UnicodeEncodeError):
            return None
        if len(flat) != k * len(kmers) or not set(map(len, kmers)) <= {k}:      # set(map(len, .)) runs in C: 2.4x the generator form
            return None
        return flat

    def _tf_list_staged(self, ix: Index, kmers, k: int):
        """list of k-character items -> list[int] through the pinned staging (packing threads write the query bytes straight into it, the
        answers come back into it); None when this batch has to take the general path (helper missing, an odd item, staging busy or too small)."""
        fast = _pyfast()
        if fast is None or not hasattr(fast, "join_fixed_into") or not isinstance(kmers, (list, tuple)) or len(kmers) < 4096:
            return None
        st = self._stage
        if not st.lock.acquire(blocking=False):
            return None
        try:
            n = len(kmers)
            qin, aout = st.view("in", n * k), st.view("out", 4 * n)
            if qin is None or aout is None:
                return None
            if fast.join_fixed_into(kmers, k, qin, _PACK_THREADS) is None:
                return None
            out = aout[: 4 * n].view(np.uint32)
            ix.tf_ascii_into(qin[: n * k], out)
            return fast.u32_list(out, n, _PACK_THREADS)
        finally:
            st.lock.release()

    def _tf_list_flat(self, ix: Index, flat, k: int) -> List[int]:
        """One packed buffer of N*k query bytes -> list[int]; the answers come back into the pinned staging when it is free."""
        from .engine import _as_u8
        a = _as_u8(flat, k)
        n = a.shape[0] // k
        st, fast = self._stage, _pyfast()
        if n >= 4096 and fast is not None and hasattr(fast, "u32_list") and st.lock.acquire(blocking=False):
            try:
                aout = st.view("out", 4 * n)
                if aout is not None:
                    out = aout[: 4 * n].view(np.uint32)
                    ix.tf_ascii_into(a, out)
                    return fast.u32_list(out, n, _PACK_THREADS)
            finally:
                st.lock.release()
        return self._to_list(ix.tf_ascii(a))

    def get_tf_values_23mer(self, kmers: List[str]) -> List[int]:
        if len(kmers) == 0:
            return []
        staged = self._tf_list_staged(self._need23(), kmers, 23)
        if staged is not None:
            return staged
        flat = self._packed(kmers, 23)                                                  # one buffer of N*23 bytes: nothing to do per item
        if flat is None and not isinstance(kmers, (str, bytes, bytearray, memoryview, np.ndarray)):
            flat = self._join_fixed(kmers, 23)                                          # common case: a list of 23-mers
        if flat is not None:
            return self._tf_list_flat(self._need23(), flat, 23)
        if isinstance(kmers, (str, bytes, bytearray, memoryview)):
            kmers = [kmers]                                                             # one query of another length
        return self._to_list(self._need23().tf_ragged(kmers))                           # :1219-1228, any lengths

    def get_tf_value_23mer(self, kmer: str) -> int:
        return self.get_tf_values_23mer([kmer])[0]

    def get_tf_values_13mer(self, kmers: List[str]) -> List[int]:
        """:938-980 (defined but not bound in the reference, although aindex.py:148 calls it)."""
        if not self._is_13mer_mode:
            return [0] * len(kmers)
        if len(kmers) == 0:
            return []
        staged = self._tf_list_staged(self._ix13, kmers, 13)
        if staged is not None:
            return staged
        flat = self._packed(kmers, 13)
        if flat is None and not isinstance(kmers, (str, bytes, bytearray, memoryview, np.ndarray)):
            flat = self._join_fixed(kmers, 13)
        if flat is not None:
            return self._tf_list_flat(self._ix13, flat, 13)
        if isinstance(kmers, (str, bytes, bytearray, memoryview)):
            kmers = [kmers]
        return self._to_list(self._ix13.tf_ragged(kmers))

    def get_tf_value_13mer(self, kmer: str) -> int:
        return self.get_tf_values_13mer([kmer])[0] if self._is_13mer_mode and self._ix13 else 0

    # numpy fast paths (zero Python-object overhead): (N, k) uint8 ASCII or flat bytes
    def get_tf_values_array(self, kmers_u8) -> np.ndarray:
        ix = self._ix13 if self._is_13mer_mode else self._need23()
        return ix.tf_ascii(kmers_u8)

    def _fixed(self, kmers: List[str], k: int):
        """Split into the indices of length-k strings and their joined bytes (others answer 0)."""
        bs = [_enc(s) for s in kmers]
        idx = [i for i, b in enumerate(bs) if len(b) == k]
        return idx, b"".join(bs[i] for i in idx)

    def get_total_tf_values_23mer(self, kmers: List[str]) -> List[int]:
        out = [0] * len(kmers)                                                          # :1230-1257; len != 23 -> 0
        idx, data = self._fixed(kmers, 23)
        if idx:
            for i, v in zip(idx, self._need23().total_ascii(data).tolist()):
                out[i] = v
        return out

    def get_total_tf_value_23mer(self, kmer: str) -> int:
        return self.get_total_tf_values_23mer([kmer])[0]

    def get_tf_both_directions_23mer_batch(self, kmers: List[str]) -> List[Tuple[int, int]]:
        out = [(0, 0)] * len(kmers)                                                     # :1259-1286
        idx, data = self._fixed(kmers, 23)
        if idx:
            f, r = self._need23().both_ascii(data)
            for i, a, b in zip(idx, f.tolist(), r.tolist()):
                out[i] = (a, b)
        return out

    def get_tf_both_directions_23mer(self, kmer: str) -> Tuple[int, int]:
        return self.get_tf_both_directions_23mer_batch([kmer])[0]

    def get_total_tf_values_13mer(self, kmers: List[str]) -> List[int]:
        if not self._is_13mer_mode:                                                     # :548-562
            return [0] * len(kmers)
        out = [0] * len(kmers)
        idx, data = self._fixed(kmers, 13)
        if idx:
            for i, v in zip(idx, self._ix13.total_ascii(data).tolist()):
                out[i] = v
        return out

    def get_total_tf_value_13mer(self, kmer: str) -> int:
        return self.get_total_tf_values_13mer([kmer])[0]

    def get_tf_both_directions_13mer_batch(self, kmers: List[str]) -> List[Tuple[int, int]]:
        if not self._is_13mer_mode:                                                     # :594-608
            return [(0, 0)] * len(kmers)
        out = [(0, 0)] * len(kmers)
        idx, data = self._fixed(kmers, 13)
        if idx:
            f, r = self._ix13.both_ascii(data)
            for i, a, b in zip(idx, f.tolist(), r.tolist()):
                out[i] = (a, b)
        return out

    def get_tf_both_directions_13mer(self, kmer: str) -> Tuple[int, int]:
        return self.get_tf_both_directions_13mer_batch([kmer])[0]

    def get_13mer_tf_array(self) -> List[int]:
        if not self._is_13mer_mode:                                                     # :983-991, u32-truncated copy
            return []
        return (np.asarray(self._tf13_host) & np.uint64(0xFFFFFFFF)).astype(np.uint32).tolist()

    def get_13mer_tf_array_numpy(self) -> np.ndarray:
        """Same data as get_13mer_tf_array without materialising 67 M Python ints."""
        return (np.asarray(self._tf13_host) & np.uint64(0xFFFFFFFF)).astype(np.uint32) if self._is_13mer_mode else np.zeros(0, np.uint32)

    def get_tf_by_index_13mer(self, index: int) -> int:
        if not self._is_13mer_mode or index >= TOTAL_13MERS or index < 0:               # :993-998
            return 0
        return int(self._tf13_host[index]) & 0xFFFFFFFF

    # ---- hash / ids --------------------------------------------------------------------------
    def get_hash_values(self, kmers: List[str]) -> List[int]:
        """:629-636 — raw mphf::lookup; only fixed 23-byte strings are supported here."""
        idx, data = self._fixed(kmers, 23)
        if len(idx) != len(kmers):
            raise ValueError("get_hash_values: every k-mer must have length 23")
        return self._need23().hash_ascii(data).tolist() if kmers else []

    def get_hash_value(self, kmer: str) -> int:
        return self.get_hash_values([kmer])[0]

    def _kid_strand(self, kmer: str):
        b = _enc(kmer)
        if len(b) != 23:
            return 0, 0
        kid, strand = self._need23().kid_strand_ascii(b)
        return int(kid[0]), int(strand[0])

    def get_kid_by_kmer(self, kmer: str) -> int:
        return self._kid_strand(kmer)[0]                                                # :700-716 (0 when absent)

    def get_strand(self, kmer: str) -> int:
        return self._kid_strand(kmer)[1]                                                # :726-742

    def _checker(self) -> np.ndarray:
        if self._checker_host is None:
            self._checker_host = self._need23().checker_array()
        return self._checker_host

    @staticmethod
    def _decode23(code: int) -> str:
        return "".join("ACGT"[(code >> (2 * (22 - i))) & 3] for i in range(23))

    def get_kmer_by_kid(self, kid: int) -> str:
        ix = self._need23()                                                             # :718-724
        if kid >= ix.n or kid < 0:
            return ""
        return self._decode23(int(self._checker()[kid]))

    def get_kmer_info(self, kid: int):
        ix = self._need23()                                                             # :744-755 -> (tf, kmer, rc)
        if kid >= ix.n or kid < 0:
            return (0, "", "")
        kmer = self._decode23(int(self._checker()[kid]))
        return (int(ix.tf_array()[kid]), kmer, self.get_reverse_complement_23mer(kmer))

    def get_reverse_complement_13mer(self, kmer: str) -> str:
        return _enc(kmer)[::-1].translate(_COMP).decode("latin-1")                      # :505-517

    def get_reverse_complement_23mer(self, kmer: str) -> str:
        b = _enc(kmer)                                                                  # :1288-1299: via the 2-bit code
        if len(b) != 23:
            return ""
        code = 0
        for c in b:
            code = (code << 2) | {65: 0, 67: 1, 71: 2, 84: 3}.get(c, 0)
        rc = _lib.lib().aix_selftest_revcomp(code, 23)
        return self._decode23(rc)

    # ---- metadata ----------------------------------------------------------------------------
    def get_hash_size(self) -> int:
        if self._is_13mer_mode:                                                         # :846-851
            return TOTAL_13MERS
        return self._ix23.n if self._ix23 is not None else 0

    def get_reads_size(self) -> int:
        return self.reads_size

    def get_index_info(self) -> str:
        lines = [f"Mode: {'13-mer' if self._is_13mer_mode else '23-mer'}", f"Total k-mers: {self.n_kmers}"]
        ix = self._ix13 if self._is_13mer_mode else self._ix23
        if ix is not None:
            i = ix.info
            lines.append(f"HBM resident bytes: {i['device_bytes']} on device {i['device']}")
        return "\n".join(lines)

    def get_13mer_statistics(self) -> dict:
        if not self._is_13mer_mode:                                                     # :1038-1068
            return {}
        tf = np.asarray(self._tf13_host)
        nz = tf[tf != 0]
        return {"total_kmers": TOTAL_13MERS, "non_zero_kmers": int(nz.shape[0]), "max_frequency": int(tf.max()),
                "total_count": int(tf.sum(dtype=np.uint64))}

    def get_23mer_statistics(self) -> str:
        if self._is_13mer_mode:                                                         # :1301-1315
            return "Not in 23-mer mode"
        return (f"23-mer Index Statistics:\nTotal k-mers: {self.n_kmers}\nTotal reads: {self.n_reads}\n"
                f"AIndex loaded: {'Yes' if self.aindex_loaded else 'No'}\nReads loaded: No\nHash map size: {self.get_hash_size()}\n")

    # ---- positions (get_positions dispatches on length, :826-831) ---------------------------------
    def get_positions(self, kmer: str) -> List[int]:
        b = _enc(kmer)
        if len(b) == 13:
            return self.get_positions_13mer(kmer)
        if len(b) == 23:
            return self.get_positions_23mer(kmer)
        return []

    def get_positions_13mer(self, kmer: str) -> List[int]:
        """:1070-1100 — 13-mer mode, exactly 13 upper-case A/C/G/T, bucket = mphf(kmer), the non-zero entries of
        positions[indices[h] : indices[h + 1]] minus one (0-based). [] when no positions file is mapped (the reference's
        permanent state, see load_13mer_aindex)."""
        if not self._is_13mer_mode or self._ix13 is None or getattr(self, "_positions13", None) is None:
            return []
        b = _enc(kmer)
        if len(b) != 13 or any(c not in b"ACGT" for c in b):
            return []
        h = self._mphf13_slot(b)
        if h >= TOTAL_13MERS:
            return []
        lo, hi = int(self._indices13[h]), min(int(self._indices13[h + 1]), int(self._positions13.shape[0]))
        seg = np.asarray(self._positions13[lo:hi])
        return (seg[seg != 0] - np.uint64(1)).tolist()

    def _mphf13_slot(self, kmer13: bytes) -> int:
        """hasher_13mer.lookup(kmer) (:1087) through the library (aix_hash_batch_ascii on the 13-mer handle)."""
        return int(self._ix13.hash_ascii(kmer13)[0])

    def get_positions_23mer(self, kmer: str) -> List[int]:
        """:800-822 via PHASH_MAP::get_pfid (hash.hpp:150-170): the lexicographically smaller of the k-mer and the
        decode of its reverse complement is looked up; an absent k-mer gives [] (the reference reads
        indices[n+1] out of bounds there)."""
        if self._ix23 is None or getattr(self, "_positions", None) is None:
            return []
        b = _enc(kmer)
        code = 0
        for c in b:
            code = (code << 2) | {65: 0, 67: 1, 71: 2, 84: 3}.get(c, 0)
        rc = _lib.lib().aix_selftest_revcomp(code, 23)
        rev = self._decode23(rc).encode()
        s, want = (b, code) if b <= rev else (rev, rc)
        h = int(self._ix23.hash_ascii(s)[0])
        if h >= self._ix23.n or int(self._checker()[h]) != want:
            return []
        seg = np.asarray(self._positions[int(self._indices[h]):int(self._indices[h + 1])])
        return (seg[seg != 0] - np.uint64(1)).tolist()

    def get_reads_se_by_kmer(self, kmer: str, max_reads: int = 100) -> List[str]:
        """Reads that hold an indexed occurrence of the k-mer, each read once, at most max_reads. The reference's version
        (:857-911) walks `positions[kmer_id] .. positions[kmer_id + 1]` over `indices[]` — its two arrays crossed — with the
        unverified mphf value of the k-mer: undefined behaviour, so there is no parity target; this is what it intends."""
        if not self.aindex_loaded:
            return []
        out, seen = [], set()
        for p in self.get_positions(kmer):
            i = self._interval(p)
            if i is None or i in seen:
                continue
            seen.add(i)
            read = self.get_read_by_rid(int(self._ridx_rid[i]))
            if read:
                out.append(read)
            if len(out) >= max_reads:
                break
        return out

    def debug_kmer_tf_values(self):
        """:913-936 — for the stored k-mers at slots 1, 10, 100, ...: one line per read that holds an indexed occurrence."""
        if self._ix23 is None:
            return
        tf = self._ix23.tf_array()
        for h in (1, 10, 100, 1000, 10000, 100000):
            if h >= self.n_kmers:
                continue
            kmer = self.get_kmer_by_kid(h)
            for _ in self.get_reads_se_by_kmer(kmer, 100):
                print(kmer, kmer, h, int(tf[h]))

    def close(self):
        for ix in (self._ix23, self._ix13):
            if ix is not None:
                ix.close()
        self._ix23 = self._ix13 = None
        self._stage.close()
