"""AIndex — counterpart of the reference's pure-Python `aindex.core.aindex.AIndex`
(aindex/core/aindex.py:48-793) for the tf / coverage path, backed by the MI355X engine.

Same public names and behaviour for: load_hash / load_13mer_index / load_from_prefix* ,
get_tf_value(s), get_hash_value(s), get_kid_by_kmer, get_kmer_by_kid, get_strand, get_kmer_info,
__getitem__/__contains__/get/__len__, iter_sequence_kmers, get_sequence_coverage (one kernel launch
instead of a Python loop), 13-mer array access. Documented deviation: load_from_prefix
auto-detect tests `.kmers.bin` first — the reference's order always selects 13-mer mode.
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


def get_revcomp(sequence: str) -> str:
    c = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "a": "t", "c": "g", "g": "c", "t": "a", "n": "n", "~": "~", "[": "]", "]": "["}
    return "".join(c.get(x, x) for x in reversed(sequence))


class AIndex:
    def __init__(self, device: int = 0):
        self._wrapper = AindexWrapper(device)
        self._loaded = False
        self.reads_size = 0
        self.max_tf = 0

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
    def load_from_prefix(prefix: str, kmer_size: Optional[int] = None, max_tf: int = 100000, load_aindex: bool = False,
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
        if kmer_size == 13:
            ix.load_from_prefix_13mer(prefix)
        elif kmer_size == 23:
            ix.load_from_prefix_23mer(prefix, max_tf=max_tf)
        else:
            raise ValueError(f"Unsupported kmer size: {kmer_size}. Only 13 and 23 are supported.")
        return ix

    def load_from_prefix_23mer(self, prefix: str, max_tf: int = 100, load_aindex: bool = False, reads_file: str = ""):
        self._wrapper.load_from_prefix_23mer(prefix, "")
        self._loaded = True

    def load_from_prefix_13mer(self, prefix: str, load_aindex: bool = False, reads_file: str = ""):
        self._wrapper.load_from_prefix_13mer(prefix, "")
        self._loaded = True

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
