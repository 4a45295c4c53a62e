"""AIndex — counterpart of the reference's pure-Python `aindex.core.aindex.AIndex`
(aindex/core/aindex.py:48-793) for the tf / coverage path, backed by the MI355X engine.

Same public names and behaviour for every member of the reference class: load_hash / load_13mer_index /
load_from_prefix* / load_aindex / load_reads / load_reads_index, get_tf_value(s), get_hash_value(s), get_kid_by_kmer,
get_kmer_by_kid, get_strand, get_kmer_info, __getitem__/__contains__/get/__len__, iter_sequence_kmers,
get_sequence_coverage (one kernel launch instead of a Python loop), 13-mer array access, positions and reads access
(get_positions / pos / get_rid / get_start / get_read* / get_rid2poses / iter_reads*, pinned against the reference's
compiled module in tests/golden/small23/access.json), get_header, k-mers by frequency. Documented deviations:
load_from_prefix auto-detect tests `.kmers.bin` first (the reference's order always selects 13-mer mode);
kmer_type="auto" follows the loaded mode (the reference's probe always lands on "13mer"); get_reads_by_kmer returns the
reads holding an indexed occurrence (the reference's implementation reads its two arrays crossed: undefined behaviour).
"""
from __future__ import annotations

import logging
import os
from enum import IntEnum
from typing import List, Optional, Tuple

import numpy as np

from .wrapper import AindexWrapper

logger = logging.getLogger(__name__)


class Strand(IntEnum):            # aindex.py:29-32
    NOT_FOUND = 0
    FORWARD = 1
    REVERSE = 2


def hamming_distance(s1: str, s2: str) -> int:
    """aindex.py:44-46 — mismatches over the common prefix, positions holding an N on either side ignored."""
    return sum(1 for a, b in zip(s1, s2) if a != b and a != "N" and b != "N")


def get_revcomp(sequence: str) -> str:
    c = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "a": "t", "c": "g", "g": "c", "t": "a", "n": "n", "~": "~", "[": "]", "]": "["}
    return "".join(c.get(x, x) for x in reversed(sequence))


class AIndex:
    def __init__(self, device: int = 0):
        self._wrapper = AindexWrapper(device)
        self._loaded = False
        self.reads_size = 0
        self.max_tf = 0
        self.loaded_header = False                                # aindex.py:55-61
        self.loaded_intervals = False
        self.loaded_reads = False
        self.rid2start = {}
        self.chrm2start = {}
        self.headers = {}
        self._hdr_start = np.zeros(0, dtype=np.int64)             # header intervals [start, start + length), file order
        self._hdr_end = np.zeros(0, dtype=np.int64)

    # ---- loading -----------------------------------------------------------------------------
    def load_hash(self, hash_file: str, tf_file: str, kmers_bin_file: str, kmers_text_file: str = ""):
        for f in (hash_file, tf_file, kmers_bin_file):           # aindex.py:63-79
            if not os.path.exists(f):
                raise FileNotFoundError(f"File not found: {f}")
        self._wrapper.load(hash_file, tf_file, kmers_bin_file, kmers_text_file)
        self._loaded = True

    load_hash_file = load_hash

    def load_13mer_index(self, hash_file: str, tf_file: str):
        if not os.path.exists(hash_file):
            raise FileNotFoundError(f"13-mer hash file not found: {hash_file}")
        if not os.path.exists(tf_file):
            raise FileNotFoundError(f"13-mer tf file not found: {tf_file}")
        self._wrapper.load_13mer_index(hash_file, tf_file)
        self._loaded = True

    @staticmethod
    def load_13mer_index_static(hash_file: str, tf_file: str, device: int = 0) -> "AIndex":
        ix = AIndex(device)
        ix.load_13mer_index(hash_file, tf_file)
        return ix

    @staticmethod
    def load_23mer_index(hash_file: str, tf_file: str, kmers_bin_file: str, kmers_text_file: str = "", device: int = 0) -> "AIndex":
        ix = AIndex(device)
        ix.load_hash(hash_file, tf_file, kmers_bin_file, kmers_text_file)
        return ix

    @staticmethod
    def load_from_prefix(prefix: str, kmer_size: Optional[int] = None, max_tf: int = 100000, load_aindex: bool = True,
                         load_reads: bool = False, device: int = 0) -> "AIndex":
        ix = AIndex(device)
        if kmer_size is None:
            pf, tf, kb = f"{prefix}.pf", f"{prefix}.tf.bin", f"{prefix}.kmers.bin"
            if os.path.exists(pf) and os.path.exists(tf) and os.path.exists(kb):
                kmer_size = 23                                   # deviation: .kmers.bin is tested first
            elif os.path.exists(pf) and os.path.exists(tf):
                kmer_size = 13
            else:
                raise FileNotFoundError(f"Could not auto-detect k-mer size for prefix '{prefix}'")
        reads_file = ""
        if load_reads:                                           # aindex.py:478-487: <prefix>.reads, else without the .23. / .13. infix
            reads_file = f"{prefix}.reads"
            if not os.path.exists(reads_file):
                reads_file = reads_file.replace(".23.", ".").replace(".13.", ".")
                if not os.path.exists(reads_file):
                    logger.warning(f"Reads file not found: {reads_file}")
                    reads_file = ""
        if kmer_size == 13:
            ix.load_from_prefix_13mer(prefix, load_aindex=load_aindex, reads_file=reads_file)
        elif kmer_size == 23:
            ix.load_from_prefix_23mer(prefix, max_tf=100000 if max_tf is None else max_tf, load_aindex=load_aindex, reads_file=reads_file)
        else:
            raise ValueError(f"Unsupported kmer size: {kmer_size}. Only 13 and 23 are supported.")
        return ix

    def load_from_prefix_23mer(self, prefix: str, max_tf: int = 100, load_aindex: bool = True, reads_file: str = ""):
        """aindex.py:501-521. A missing positions index is a warning here (the reference's C++ loader calls std::terminate())."""
        self._wrapper.load_from_prefix_23mer(prefix, reads_file)
        self._loaded = True
        self._after_reads()
        if load_aindex:
            try:
                self._wrapper.load_aindex_from_prefix_23mer(prefix, max_tf, reads_file)
                self.max_tf = max_tf
            except Exception as e:
                logger.warning(f"Could not load 23-mer AIndex from prefix {prefix}: {e}")

    def load_from_prefix_13mer(self, prefix: str, load_aindex: bool = True, reads_file: str = ""):
        self._wrapper.load_from_prefix_13mer(prefix, reads_file)                       # aindex.py:523-543
        self._loaded = True
        self._after_reads()
        if load_aindex:
            try:
                self._wrapper.load_aindex_from_prefix_13mer(prefix, reads_file)
            except Exception as e:
                logger.warning(f"Could not load 13-mer AIndex from prefix {prefix}: {e}")

    def _after_reads(self):
        self.reads_size = self._wrapper.reads_size
        self.loaded_reads = self.reads_size > 0

    # ---- queries -----------------------------------------------------------------------------
    def get_tf_value(self, kmer: str) -> int:
        return self._wrapper.get_tf_value(kmer) if self._loaded else 0

    def get_tf_values(self, kmers: List[str]) -> List[int]:
        return self._wrapper.get_tf_values(kmers) if self._loaded else [0] * len(kmers)

    def get_tf_values_13mer(self, kmers: List[str]) -> List[int]:
        return self._wrapper.get_tf_values_13mer(kmers) if self._loaded else [0] * len(kmers)

    def get_tf_values_array(self, kmers_u8) -> np.ndarray:
        """(N, k) uint8 ASCII -> uint32 tf, no Python objects on the way."""
        return self._wrapper.get_tf_values_array(kmers_u8)

    def _req(self):
        if not self._loaded:
            raise RuntimeError("Index not loaded")

    def get_hash_value(self, kmer: str) -> int:
        self._req()
        return self._wrapper.get_hash_value(kmer)

    def get_hash_values(self, kmers: List[str]) -> List[int]:
        self._req()
        return self._wrapper.get_hash_values(kmers)

    def get_kid_by_kmer(self, kmer: str) -> int:
        self._req()
        return self._wrapper.get_kid_by_kmer(kmer)

    def get_kmer_by_kid(self, kid: int) -> str:
        self._req()
        return self._wrapper.get_kmer_by_kid(kid)

    def get_strand(self, kmer: str) -> Strand:
        self._req()
        return Strand(self._wrapper.get_strand(kmer))

    def get_kmer_info(self, kid: int) -> Tuple[str, str, int]:
        self._req()                                              # aindex.py:195-207 -> (kmer, rkmer, tf)
        kmer = self.get_kmer_by_kid(kid)
        return kmer, get_revcomp(kmer), self.get_tf_value(kmer)

    def get_kmer_info_by_kid(self, kid: int, k: int = 23):
        return self.get_kmer_info(kid)

    def get_hash_size(self) -> int:
        self._req()
        return self._wrapper.get_hash_size()

    def get_reads_size(self) -> int:
        return self._wrapper.get_reads_size()

    def __len__(self) -> int:
        return self.get_hash_size()

    def __getitem__(self, kmer: str) -> int:
        return self.get_tf_value(kmer)

    def __contains__(self, kmer: str) -> bool:
        return self[kmer] > 0

    def get(self, kmer: str, default: int = 0) -> int:
        tf = self[kmer]
        return tf if tf > 0 else default

    def iter_sequence_kmers(self, sequence: str, k: int = 23):
        kmers = [sequence[i:i + k] for i in range(len(sequence) - k + 1)]     # aindex.py:306-312, one batch call
        kmers = [s for s in kmers if "\n" not in s and "~" not in s]
        for s, tf in zip(kmers, self.get_tf_values(kmers)):
            yield s, tf

    def get_sequence_coverage(self, seq: str, cutoff: int = 0, k: int = 23) -> list:
        """aindex.py:314-322. Every window goes through get_tf_value in the reference, i.e. the mode's
        own k decides what can match; windows are k bytes long."""
        return self.get_sequences_coverage([seq], cutoff, k)[0].tolist()

    def get_sequences_coverage(self, seqs: List[str], cutoff: int = 0, k: int = 23) -> List[np.ndarray]:
        if not self._loaded:
            return [np.zeros(max(0, len(s) - k + 1), dtype=np.uint32) for s in seqs]
        w = self._wrapper
        ix = w._ix13 if w._is_13mer_mode else w._ix23
        if ix is not None and ix.k == k:
            return ix.coverage(seqs, cutoff)
        # window length differs from the index's k: the reference's get_tf_value then sees strings of
        # the wrong length — go through the exact ragged path
        out = []
        for s in seqs:
            wins = [s[i:i + k] for i in range(len(s) - k + 1)]
            tf = np.array(self.get_tf_values(wins), dtype=np.uint32) if wins else np.zeros(0, dtype=np.uint32)
            tf[tf < cutoff] = 0
            out.append(tf)
        return out

    def print_sequence_coverage(self, seq: str, cutoff: int = 0):
        for i, tf in enumerate(self.get_sequence_coverage(seq, cutoff)):
            print(i, seq[i:i + 23], tf)

    # ---- reads + positions index (N2 tier: thin calls into the wrapper, aindex.py:88-133,162-181,271-343) ----------
    def load_reads(self, reads_file: str):
        if not os.path.exists(reads_file):
            raise FileNotFoundError(f"Reads file not found: {reads_file}")
        self._wrapper.load_reads(reads_file)
        self.reads_size = self._wrapper.reads_size
        self.loaded_reads = True

    def load_aindex(self, index_file: str, indices_file: str, max_tf: int):
        for name, path in (("index", index_file), ("indices", indices_file)):
            if not os.path.exists(path):
                raise FileNotFoundError(f"{name} file not found: {path}")
        self._wrapper.load_aindex(index_file, indices_file, max_tf)
        self.max_tf = max_tf

    def load_13mer_aindex(self, index_file: str, indices_file: str):
        for name, path in (("index", index_file), ("indices", indices_file)):
            if not os.path.exists(path):
                raise FileNotFoundError(f"{name} file not found: {path}")
        self._wrapper.load_13mer_aindex(index_file, indices_file)

    def load_reads_index(self, index_file: str, header_file: Optional[str] = None):
        """aindex.py:101-130 — `.ridx` lines "rid<TAB>start<TAB>end" and, optionally, `.header` lines
        "head<TAB>start<TAB>length". The reference keeps both in one IntervalTree; here they are two sorted arrays."""
        self.rid2start, self.chrm2start, self.headers = {}, {}, {}
        with open(index_file) as fh:
            for line in fh:
                rid, start, end = line.rstrip("\n").split("\t")
                self.rid2start[int(rid)] = (int(start), int(end))
        self._wrapper.load_reads_index(index_file)
        self.loaded_intervals = True
        if header_file:
            starts, ends = [], []
            with open(header_file) as fh:
                for rid, line in enumerate(fh):
                    head, start, length = line.rstrip("\n").split("\t")
                    self.headers[rid] = head
                    self.chrm2start[head.split()[0].split(".")[0]] = int(start)
                    starts.append(int(start))
                    ends.append(int(start) + int(length))
            self._hdr_start = np.array(starts, dtype=np.int64)
            self._hdr_end = np.array(ends, dtype=np.int64)
            self.loaded_header = True

    def get_positions(self, kmer: str) -> List[int]:
        return self._wrapper.get_positions(kmer)

    def get_positions_13mer(self, kmer: str) -> List[int]:
        return self._wrapper.get_positions_13mer(kmer)

    def pos(self, kmer: str) -> List[int]:
        return self.get_positions(kmer)

    def get_read_by_rid(self, rid: int) -> str:
        return self._wrapper.get_read_by_rid(rid)

    def get_read(self, start: int, end: int, revcomp: bool = False) -> str:
        return self._wrapper.get_read(start, end, revcomp)

    def get_rid(self, pos: int) -> int:
        return self._wrapper.get_rid(pos)

    def get_start(self, pos: int) -> int:
        return self._wrapper.get_start(pos)

    def get_rid2poses(self, kmer: str) -> dict:
        """aindex.py:333-341 — read id -> offsets of the k-mer inside that read."""
        hits = {}
        for p in self.pos(kmer):
            hits.setdefault(self.get_rid(p), []).append(p - self.get_start(p))
        return hits

    def get_reads_by_kmer(self, kmer: str, max_reads: int = 100) -> List[str]:
        """aindex.py:162-166 over get_reads_se_by_kmer (see the wrapper for the one documented deviation)."""
        if not self._wrapper.aindex_loaded:
            raise RuntimeError("Aindex not loaded")
        return self._wrapper.get_reads_se_by_kmer(kmer, max_reads)

    def iter_reads(self):
        if self.reads_size == 0:                                  # aindex.py:271-278
            raise RuntimeError("Reads were not loaded.")
        for rid in range(self.n_reads):
            yield rid, self.get_read_by_rid(rid)

    def iter_reads_se(self):
        if self.reads_size == 0:                                  # aindex.py:280-290
            raise RuntimeError("Reads were not loaded.")
        for rid in range(self.n_reads):
            for idx, sub in enumerate(self.get_read_by_rid(rid).split("~")):
                yield rid, idx, sub

    def get_header(self, pos: int) -> Optional[str]:
        """aindex.py:296-304 — header of the record whose interval [start, start + length) holds pos; None before any
        header file was loaded, '' when no interval holds pos."""
        if not self.loaded_header:
            return None
        hit = np.nonzero((self._hdr_start <= pos) & (pos < self._hdr_end))[0]
        return self.headers.get(int(hit[0]), "") if hit.shape[0] else ""

    # ---- k-mers by frequency (aindex.py:574-793) ---------------------------------------------------------------
    def _index_to_13mer(self, index: int) -> str:
        return "".join("ACGT"[(index >> (2 * (12 - i))) & 3] for i in range(13))

    def _kmer_type(self, kmer_type: str) -> str:
        if kmer_type == "auto":                                   # the reference probes get_13mer_tf_array(), which
            return "13mer" if self._wrapper._is_13mer_mode else "23mer"   # answers [] without raising in 23-mer mode: it always
        if kmer_type not in ("13mer", "23mer"):                   # lands on "13mer"; we pick by the loaded mode (deviation)
            raise ValueError(f"Unsupported kmer_type: {kmer_type}. Use '13mer', '23mer', or 'auto'")
        return kmer_type

    def _frequencies(self, kmer_type: str):
        """(labels as a callable index -> str, tf array) exactly as the reference enumerates them: 13-mer mode walks the
        tf array in FILE order (mphf order) and labels entry i with the base-4 spelling of i (aindex.py:633-649 — the label
        is the 2-bit decoding of the mphf index, not the k-mer counted there; kept for drop-in parity); 23-mer mode walks
        kid = 0..n-1 and asks get_tf_value(get_kmer_by_kid(kid)) — one batch lookup here."""
        if kmer_type == "13mer":
            tf = np.asarray(self._wrapper.get_13mer_tf_array_numpy()).astype(np.uint64) & np.uint64(0xFFFFFFFF)   # u32 view of the API
            return self._index_to_13mer, tf
        if self.n_kmers == 0:
            raise RuntimeError("23-mer index not properly loaded")
        from . import synth
        codes = self._wrapper._checker() & np.uint64((1 << 46) - 1)                    # get_kmer_by_kid(kid) = decode of checker[kid]
        kmers = synth.decode_kmers(codes, 23)                                          # (n, 23) uint8: no Python strings until one is yielded
        tf = self.get_tf_values_array(kmers).astype(np.uint64)                         # get_tf_value(kmer) for every kid: ONE batch lookup
        return (lambda i: bytes(kmers[i]).decode()), tf

    def iter_kmers_by_frequency(self, min_tf: int = 1, max_kmers: Optional[int] = None, kmer_type: str = "auto"):
        if not self._loaded:
            raise RuntimeError("Index not loaded")
        label, tf = self._frequencies(self._kmer_type(kmer_type))
        keep = np.nonzero(tf >= np.uint64(max(min_tf, 0)))[0]
        order = keep[np.argsort(-tf[keep].astype(np.int64), kind="stable")]       # descending tf, ties in enumeration order
        if max_kmers is not None:
            order = order[:max_kmers]
        for i in order:
            yield label(int(i)), int(tf[i])

    def get_top_kmers(self, n: int = 100, min_tf: int = 1, kmer_type: str = "auto") -> List[Tuple[str, int]]:
        return list(self.iter_kmers_by_frequency(min_tf=min_tf, max_kmers=n, kmer_type=kmer_type))

    def get_kmer_frequency_stats(self, kmer_type: str = "auto") -> dict:
        if not self._loaded:
            raise RuntimeError("Index not loaded")
        kt = self._kmer_type(kmer_type)
        _, tf = self._frequencies(kt)
        nz = tf[tf > 0]
        total = int(tf.shape[0])
        return {"kmer_type": kt, "total_kmers": total, "non_zero_kmers": int(nz.shape[0]), "zero_kmers": total - int(nz.shape[0]),
                "max_tf": int(nz.max()) if nz.shape[0] else 0, "min_tf": int(nz.min()) if nz.shape[0] else 0,
                "avg_tf": (int(nz.sum(dtype=np.uint64)) / int(nz.shape[0])) if nz.shape[0] else 0,
                "total_tf": int(tf.sum(dtype=np.uint64)) if nz.shape[0] else 0,
                "coverage": (int(nz.shape[0]) / total) if total else 0}

    # ---- 13-mer array access -----------------------------------------------------------------
    def get_13mer_tf_array(self) -> List[int]:
        return self._wrapper.get_13mer_tf_array()

    def get_tf_by_index_13mer(self, index: int) -> int:
        return self._wrapper.get_tf_by_index_13mer(index)

    def get_total_tf_value_13mer(self, kmer: str) -> int:
        return self._wrapper.get_total_tf_value_13mer(kmer)

    def get_total_tf_values_13mer(self, kmers: List[str]) -> List[int]:
        return self._wrapper.get_total_tf_values_13mer(kmers)

    def get_index_info(self) -> str:
        return self._wrapper.get_index_info()

    @property
    def n_kmers(self) -> int:
        return self._wrapper.n_kmers

    @property
    def n_reads(self) -> int:
        return self._wrapper.n_reads

    @property
    def aindex_loaded(self) -> bool:
        return self._wrapper.aindex_loaded
