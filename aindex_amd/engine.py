"""Index — a k-mer index resident in MI355X HBM, driven through the C ABI (include/aindex_hip.h).

Two families of methods:
  * numpy / bytes in host memory  -> staged through HBM by the library (`aix_*`)
  * torch tensors already in HBM  -> zero-copy, asynchronous on torch's current stream (`aix_*_dev`)
All results are bit-exact with the reference's CPU implementation (see tests/).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import check, lib, vp


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(vp)


def _as_u8(kmers, k: int) -> np.ndarray:
    """Accept bytes / (N,k) uint8 / flat uint8 and return a flat contiguous uint8 array of N*k bytes."""
    if isinstance(kmers, (bytes, bytearray, memoryview)):
        a = np.frombuffer(kmers, dtype=np.uint8)
    else:
        a = np.ascontiguousarray(kmers, dtype=np.uint8).reshape(-1)
    if a.shape[0] % k:
        raise ValueError(f"k-mer buffer length {a.shape[0]} is not a multiple of k={k}")
    return a


def _ragged(items: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in items]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return np.frombuffer(b"".join(bs), dtype=np.uint8), offs


def _stream_ptr(device=None):
    """torch's current stream ON `device` (the index's device): the library switches to the handle's device for the call,
    so a stream of whichever device happens to be current would be a stream of the wrong device."""
    import torch
    return vp(torch.cuda.current_stream(device).cuda_stream)


class Index:
    def __init__(self, handle: vp):
        self._h = handle
        self._info = _lib.Info()
        check(lib().aix_index_info(self._h, C.byref(self._info)), "aix_index_info")

    # ---- lifecycle ---------------------------------------------------------------------------
    @classmethod
    def open_23(cls, pf: str, tf_bin: str, kmers_bin: str, device: int = 0) -> "Index":
        h = vp()
        check(lib().aix_index_open_23(pf.encode(), tf_bin.encode(), kmers_bin.encode(), device, C.byref(h)),
              f"aix_index_open_23({pf})")
        return cls(h)

    @classmethod
    def open_13(cls, pf: str, tf_bin: Optional[str], device: int = 0) -> "Index":
        h = vp()
        check(lib().aix_index_open_13(pf.encode(), tf_bin.encode() if tf_bin else None, device, C.byref(h)),
              f"aix_index_open_13({pf})")
        return cls(h)

    @classmethod
    def create_23(cls, pf_bytes: bytes, checker: np.ndarray, tf: np.ndarray, device: int = 0) -> "Index":
        checker = np.ascontiguousarray(checker, dtype=np.uint64)
        tf = np.ascontiguousarray(tf, dtype=np.uint32)
        assert checker.shape == tf.shape
        h = vp()
        buf = np.frombuffer(pf_bytes, dtype=np.uint8)
        check(lib().aix_index_create_23(_np_ptr(buf), buf.shape[0], _np_ptr(checker), _np_ptr(tf), checker.shape[0],
                                        device, C.byref(h)), "aix_index_create_23")
        return cls(h)

    @classmethod
    def create_13(cls, pf_bytes: bytes, tf: Optional[np.ndarray] = None, device: int = 0) -> "Index":
        if tf is not None:
            tf = np.ascontiguousarray(tf, dtype=np.uint64)
            assert tf.shape[0] == _lib.TOTAL_13MERS
        h = vp()
        buf = np.frombuffer(pf_bytes, dtype=np.uint8)
        check(lib().aix_index_create_13(_np_ptr(buf), buf.shape[0], _np_ptr(tf), device, C.byref(h)), "aix_index_create_13")
        return cls(h)

    @classmethod
    def build_23_codes_t(cls, pf_bytes: bytes, keys_t, counts_t=None, device: Optional[int] = None) -> "Index":
        """I1 on the device: scatter (2-bit code, count) pairs that already live in HBM through the MPHF and
        keep the result resident. keys_t: int64/uint64 tensor; counts_t: int32 tensor or None (tf = 0)."""
        import torch
        dev = keys_t.device.index if device is None else device
        h = vp()
        buf = np.frombuffer(pf_bytes, dtype=np.uint8)
        with torch.cuda.device(dev):
            check(lib().aix_index_build_23_codes_dev(_np_ptr(buf), buf.shape[0], vp(keys_t.data_ptr()),
                                                     vp(counts_t.data_ptr()) if counts_t is not None else None,
                                                     keys_t.numel(), dev, _stream_ptr(dev), C.byref(h)), "aix_index_build_23_codes_dev")
        return cls(h)

    def close(self):
        if self._h is not None and self._h.value:
            lib().aix_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- metadata ----------------------------------------------------------------------------
    @property
    def k(self) -> int:
        return self._info.k

    @property
    def n(self) -> int:
        return self._info.n

    @property
    def device(self) -> int:
        return self._info.device

    @property
    def canonical_only(self) -> bool:
        return bool(self._info.canonical_only)

    @property
    def info(self) -> dict:
        i = self._info
        check(lib().aix_index_info(self._h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in i._fields_ if not f.startswith("reserved")}

    def set_canonical_fastpath(self, enabled: bool):
        check(lib().aix_index_set_canonical_fastpath(self._h, int(enabled)))

    def set_fingerprint_filter(self, enabled: bool):
        check(lib().aix_index_set_fingerprint_filter(self._h, int(enabled)))

    def set_early_exit(self, enabled: bool):
        check(lib().aix_index_set_early_exit(self._h, int(enabled)))

    def set_bucket_table(self, enabled: bool, lanes: int = 0):
        """Verification table on / off (answers are identical); lanes = lanes sharing one bucket read (8, 4, 2, 1; 0 = keep)."""
        check(lib().aix_index_set_bucket_table(self._h, int(enabled), lanes))

    def set_absence_filter(self, enabled: bool):
        """Blocked Bloom filter in front of the verification table on / off (answers are identical)."""
        check(lib().aix_index_set_absence_filter(self._h, int(enabled)))

    def set_minimizer_table(self, enabled: bool):
        """Minimizer-keyed copy of the verification table (streaming consumers) on / off (answers are identical)."""
        check(lib().aix_index_set_minimizer_table(self._h, int(enabled)))

    def probe_profile(self) -> dict:
        """What one probe that FINDS its key reads under the current settings (bench.py's roofline accounting)."""
        i = self.info
        if i["bucket_table"] and i["minimizer_lines"]:
            return {"name": "minimizer-keyed verification table: one 128-byte line per super-k-mer (about 7 consecutive windows), re-used from registers / L1",
                    "bytes_per_hit_probe": 128.0, "lines_per_hit_probe": 1.0, "hbm_lines_per_hit_probe": 0.2}
        if i["bucket_table"]:
            return {"name": f"verification table: one 128-byte bucket line per probe ({i['bucket_lanes']} lanes per line)",
                    "bytes_per_hit_probe": 128.0, "lines_per_hit_probe": 1.0}
        return {"name": "three 16-byte MPHF records + one 16-byte key record", "bytes_per_hit_probe": 64.0, "lines_per_hit_probe": 4.0}

    def set_tf_13(self, tf: np.ndarray):
        tf = np.ascontiguousarray(tf, dtype=np.uint64)
        assert tf.shape[0] == _lib.TOTAL_13MERS
        check(lib().aix_index_set_tf_13(self._h, _np_ptr(tf)), "aix_index_set_tf_13")

    def tf_array(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.uint64 if self.k == 13 else np.uint32)
        check(lib().aix_index_get_tf(self._h, _np_ptr(out), out.nbytes), "aix_index_get_tf")
        return out

    def checker_array(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.uint64)
        check(lib().aix_index_get_checker(self._h, _np_ptr(out), out.shape[0]), "aix_index_get_checker")
        return out

    # ---- host-memory queries -----------------------------------------------------------------
    def tf_ascii(self, kmers) -> np.ndarray:
        a = _as_u8(kmers, self.k)
        n = a.shape[0] // self.k
        out = np.empty(n, dtype=np.uint32)
        check(lib().aix_tf_batch_ascii(self._h, _np_ptr(a), n, _np_ptr(out)), "aix_tf_batch_ascii")
        return out

    def tf_ascii_into(self, a: np.ndarray, out: np.ndarray) -> np.ndarray:
        """tf_ascii on caller-owned buffers (contiguous uint8[n*k] in, uint32[n] out) — e.g. pinned staging kept between calls."""
        assert a.dtype == np.uint8 and out.dtype == np.uint32 and a.flags.c_contiguous and out.flags.c_contiguous and a.shape[0] == out.shape[0] * self.k
        check(lib().aix_tf_batch_ascii(self._h, _np_ptr(a), out.shape[0], _np_ptr(out)), "aix_tf_batch_ascii")
        return out

    def tf_codes(self, codes: np.ndarray) -> np.ndarray:
        c = np.ascontiguousarray(codes, dtype=np.uint64)
        out = np.empty(c.shape[0], dtype=np.uint32)
        check(lib().aix_tf_batch_codes(self._h, _np_ptr(c), c.shape[0], _np_ptr(out)), "aix_tf_batch_codes")
        return out

    def tf_ragged(self, items: Sequence) -> np.ndarray:
        data, offs = _ragged(items)
        out = np.empty(len(items), dtype=np.uint32)
        check(lib().aix_tf_batch_ragged(self._h, _np_ptr(data), _np_ptr(offs), len(items), _np_ptr(out)), "aix_tf_batch_ragged")
        return out

    def hash_ascii(self, kmers) -> np.ndarray:
        a = _as_u8(kmers, self.k)
        n = a.shape[0] // self.k
        out = np.empty(n, dtype=np.uint64)
        check(lib().aix_hash_batch_ascii(self._h, _np_ptr(a), n, _np_ptr(out)), "aix_hash_batch_ascii")
        return out

    def kid_strand_ascii(self, kmers) -> Tuple[np.ndarray, np.ndarray]:
        a = _as_u8(kmers, self.k)
        n = a.shape[0] // self.k
        kid = np.empty(n, dtype=np.uint64)
        strand = np.empty(n, dtype=np.uint8)
        check(lib().aix_kid_strand_batch_ascii(self._h, _np_ptr(a), n, _np_ptr(kid), _np_ptr(strand)), "aix_kid_strand_batch_ascii")
        return kid, strand

    def both_ascii(self, kmers) -> Tuple[np.ndarray, np.ndarray]:
        a = _as_u8(kmers, self.k)
        n = a.shape[0] // self.k
        f = np.empty(n, dtype=np.uint64)
        r = np.empty(n, dtype=np.uint64)
        check(lib().aix_tf_both_batch_ascii(self._h, _np_ptr(a), n, _np_ptr(f), _np_ptr(r)), "aix_tf_both_batch_ascii")
        return f, r

    def total_ascii(self, kmers) -> np.ndarray:
        a = _as_u8(kmers, self.k)
        n = a.shape[0] // self.k
        out = np.empty(n, dtype=np.uint64)
        check(lib().aix_tf_total_batch_ascii(self._h, _np_ptr(a), n, _np_ptr(out)), "aix_tf_total_batch_ascii")
        return out

    def coverage(self, seqs: Sequence, cutoff: int = 0):
        """Per-position tf profile of each sequence (list of np.uint32 arrays, len - k + 1 each)."""
        data, offs = _ragged(seqs)
        lens = np.diff(offs).astype(np.int64)
        outl = np.maximum(lens - self.k + 1, 0).astype(np.uint64)
        ooffs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        ooffs[1:] = np.cumsum(outl, dtype=np.uint64)
        out = np.zeros(int(ooffs[-1]), dtype=np.uint32)
        if len(seqs):
            check(lib().aix_coverage_batch(self._h, _np_ptr(data), _np_ptr(offs), len(seqs), cutoff, _np_ptr(out), _np_ptr(ooffs)),
                  "aix_coverage_batch")
        return [out[int(ooffs[i]):int(ooffs[i + 1])] for i in range(len(seqs))]

    # ---- counting ----------------------------------------------------------------------------
    def count13(self, buf: bytes, fmt: int = _lib.FMT_AUTO) -> np.ndarray:
        a = np.frombuffer(buf, dtype=np.uint8)
        out = np.empty(_lib.TOTAL_13MERS, dtype=np.uint64)
        check(lib().aix_count13(self._h, _np_ptr(a), a.shape[0], fmt, _np_ptr(out)), "aix_count13")
        return out

    def count23_fixed(self, buf: bytes, fmt: int = _lib.FMT_AUTO, canon_mode: int = _lib.CANON_TRUE_RC) -> np.ndarray:
        a = np.frombuffer(buf, dtype=np.uint8)
        out = np.zeros(self.n, dtype=np.uint32)
        check(lib().aix_count23_fixed(self._h, _np_ptr(a), a.shape[0], fmt, canon_mode, _np_ptr(out)), "aix_count23_fixed")
        return out

    def count13_file(self, path: str, out_path: Optional[str] = None, fmt: int = _lib.FMT_AUTO, want_array: bool = True):
        """count_kmers13 on a FILE, streamed (aix_count13_file: the file is read part by part into pinned staging while the previous part is
        counted; the host never holds it). Returns (u64[4^13] or None, stats dict); out_path receives the reference's output file."""
        out = np.empty(_lib.TOTAL_13MERS, dtype=np.uint64) if want_array else None
        st = _lib.IngestStats()
        check(lib().aix_count13_file(self._h, os.fsencode(path), fmt, os.fsencode(out_path) if out_path else None, _np_ptr(out), C.byref(st)),
              f"aix_count13_file({path})")
        return out, st.as_dict()

    def count23_fixed_file(self, path: str, fmt: int = _lib.FMT_AUTO, canon_mode: int = _lib.CANON_TRUE_RC):
        """aix_count23_fixed on a FILE, streamed. Returns (u32[n], stats dict)."""
        out = np.zeros(self.n, dtype=np.uint32)
        st = _lib.IngestStats()
        check(lib().aix_count23_fixed_file(self._h, os.fsencode(path), fmt, canon_mode, _np_ptr(out), C.byref(st)), f"aix_count23_fixed_file({path})")
        return out, st.as_dict()

    def positions_fill(self, reads):
        """A1 + A2: (indices uint64[n+1], positions uint64[sum tf]) = the .indices.bin / .index.bin images.
        `reads`: bytes or any buffer (e.g. a numpy memmap of the reads file: nothing is copied on the host)."""
        a = np.frombuffer(reads, dtype=np.uint8)
        indices = np.empty(self.n + 1, dtype=np.uint64)
        total = C.c_uint64()
        check(lib().aix_positions_fill(self._h, _np_ptr(a), a.shape[0], _np_ptr(indices), None, 0, C.byref(total)), "aix_positions_fill")
        pos = np.zeros(total.value, dtype=np.uint64)
        check(lib().aix_positions_fill(self._h, _np_ptr(a), a.shape[0], _np_ptr(indices), _np_ptr(pos), pos.shape[0], C.byref(total)),
              "aix_positions_fill")
        return indices, pos

    def positions_indices(self) -> np.ndarray:
        """A1 alone: uint64[n+1] exclusive prefix sum of tf (the .indices.bin image)."""
        indices = np.empty(self.n + 1, dtype=np.uint64)
        total = C.c_uint64()
        check(lib().aix_positions_fill(self._h, None, 0, _np_ptr(indices), None, 0, C.byref(total)), "aix_positions_fill")
        return indices

    def positions_bucket_counts(self, reads: bytes, first_shard: bool = True) -> np.ndarray:
        """uint64[n]: windows of `reads` per bucket under A2's rules (the shard-local ppositions counters)."""
        a = np.frombuffer(reads, dtype=np.uint8)
        out = np.zeros(self.n, dtype=np.uint64)
        check(lib().aix_positions_bucket_counts(self._h, _np_ptr(a), a.shape[0], int(first_shard), _np_ptr(out)), "aix_positions_bucket_counts")
        return out

    def positions_fill_shard(self, reads: bytes, total: int, first_shard: bool, base_offset: int, filled_init: Optional[np.ndarray]) -> np.ndarray:
        """uint64[total]: this shard's entries of the positions array (zero elsewhere); see aix_positions_fill_shard."""
        a = np.frombuffer(reads, dtype=np.uint8)
        pos = np.zeros(total, dtype=np.uint64)
        f = None if filled_init is None else np.ascontiguousarray(filled_init, dtype=np.uint32)
        check(lib().aix_positions_fill_shard(self._h, _np_ptr(a), a.shape[0], int(first_shard), base_offset, _np_ptr(f), _np_ptr(pos), pos.shape[0]),
              "aix_positions_fill_shard")
        return pos

    # ---- HBM-resident (torch) entry points: asynchronous on torch's current stream ------------
    def _chk_dev(self, t):
        if not t.is_cuda or t.device.index != self.device:
            raise ValueError(f"tensor must live on cuda:{self.device}")
        if not t.is_contiguous():
            raise ValueError("tensor must be contiguous")

    def tf_ascii_t(self, kmers_t, out_t=None):
        import torch
        self._chk_dev(kmers_t)
        n = kmers_t.numel() // self.k
        if out_t is None:
            out_t = torch.empty(n, dtype=torch.int32, device=kmers_t.device)   # u32 bit patterns
        check(lib().aix_tf_batch_ascii_dev(self._h, vp(kmers_t.data_ptr()), n, vp(out_t.data_ptr()), _stream_ptr(self.device)),
              "aix_tf_batch_ascii_dev")
        return out_t

    def lines_ascii_t(self, kmers_t):
        """Instrumentation: records (128-byte lines) each tf query reads under the current settings."""
        import torch
        self._chk_dev(kmers_t)
        n = kmers_t.numel() // self.k
        out_t = torch.empty(n, dtype=torch.int32, device=kmers_t.device)
        check(lib().aix_lines_batch_ascii_dev(self._h, vp(kmers_t.data_ptr()), n, vp(out_t.data_ptr()), _stream_ptr(self.device)), "aix_lines_batch_ascii_dev")
        return out_t

    def tf_codes_t(self, codes_t, out_t=None):
        import torch
        self._chk_dev(codes_t)
        n = codes_t.numel()
        if out_t is None:
            out_t = torch.empty(n, dtype=torch.int32, device=codes_t.device)
        check(lib().aix_tf_batch_codes_dev(self._h, vp(codes_t.data_ptr()), n, vp(out_t.data_ptr()), _stream_ptr(self.device)),
              "aix_tf_batch_codes_dev")
        return out_t

    def total_ascii_t(self, kmers_t, out_t=None):
        import torch
        self._chk_dev(kmers_t)
        n = kmers_t.numel() // self.k
        if out_t is None:
            out_t = torch.empty(n, dtype=torch.int64, device=kmers_t.device)
        check(lib().aix_tf_total_batch_ascii_dev(self._h, vp(kmers_t.data_ptr()), n, vp(out_t.data_ptr()), _stream_ptr(self.device)),
              "aix_tf_total_batch_ascii_dev")
        return out_t

    def coverage_t(self, seqs_t, offs_t, out_offs_t, total_out: int, cutoff: int = 0, out_t=None):
        import torch
        self._chk_dev(seqs_t)
        m = offs_t.numel() - 1
        if out_t is None:
            out_t = torch.zeros(total_out, dtype=torch.int32, device=seqs_t.device)
        check(lib().aix_coverage_batch_dev(self._h, vp(seqs_t.data_ptr()), vp(offs_t.data_ptr()), m, seqs_t.numel(), cutoff,
                                           vp(out_t.data_ptr()), vp(out_offs_t.data_ptr()), _stream_ptr(self.device)), "aix_coverage_batch_dev")
        return out_t

    def positions_fill_t(self, reads_t):
        """A1 + A2 with the reads buffer already in HBM: (indices int64[n+1], positions int64[sum tf]) device tensors
        holding the u64 bit patterns of the .indices.bin / .index.bin images."""
        import torch
        self._chk_dev(reads_t)
        total = C.c_uint64()
        check(lib().aix_positions_total(self._h, C.byref(total)), "aix_positions_total")
        # the reference's start adjustment looks at the head of the buffer (hash.cpp:973-986); fetch as much of it as needed
        head_len, start = 1 << 16, C.c_uint64()
        while True:
            head = reads_t[:head_len].cpu().numpy()
            check(lib().aix_positions_start_k(_np_ptr(head), head.shape[0], self.k, C.byref(start)), "aix_positions_start_k")
            if head.shape[0] == reads_t.numel() or start.value + 64 < head.shape[0]:
                break
            head_len *= 16
        dev = reads_t.device
        indices = torch.empty(self.n + 1, dtype=torch.int64, device=dev)
        pos = torch.empty(max(total.value, 1), dtype=torch.int64, device=dev)[: total.value]
        with torch.cuda.device(dev):
            check(lib().aix_positions_fill_dev(self._h, vp(reads_t.data_ptr()), reads_t.numel(), start.value, vp(indices.data_ptr()),
                                               vp(pos.data_ptr()) if total.value else None, total.value, _stream_ptr(self.device)), "aix_positions_fill_dev")
        return indices, pos

    def count13_t(self, plain_t, out_t=None):
        import torch
        self._chk_dev(plain_t)
        if out_t is None:
            out_t = torch.empty(_lib.TOTAL_13MERS, dtype=torch.int64, device=plain_t.device)
        check(lib().aix_count13_dev(self._h, vp(plain_t.data_ptr()), plain_t.numel(), vp(out_t.data_ptr()), _stream_ptr(self.device)),
              "aix_count13_dev")
        return out_t

    def count23_fixed_t(self, plain_t, canon_mode: int = _lib.CANON_TRUE_RC, out_t=None):
        """Accumulates into out_t (int32[n], caller-zeroed when given)."""
        import torch
        self._chk_dev(plain_t)
        if out_t is None:
            out_t = torch.zeros(self.n, dtype=torch.int32, device=plain_t.device)
        check(lib().aix_count23_fixed_dev(self._h, vp(plain_t.data_ptr()), plain_t.numel(), canon_mode, vp(out_t.data_ptr()),
                                          _stream_ptr(self.device)), "aix_count23_fixed_dev")
        return out_t


# ---- synthetic inputs generated in HBM (mirrors aindex_amd/synth.py) ---------------------------
def synth_genome_t(seed: int, length: int, device: int = 0):
    import torch
    t = torch.empty(length, dtype=torch.uint8, device=f"cuda:{device}")
    with torch.cuda.device(device):
        check(lib().aix_synth_genome_dev(seed, length, vp(t.data_ptr()), _stream_ptr()), "aix_synth_genome_dev")
    return t


def synth_kmers_t(seed: int, n: int, k: int, device: int = 0, first: int = 0):
    import torch
    t = torch.empty(n * k + 16, dtype=torch.uint8, device=f"cuda:{device}")[: n * k]
    with torch.cuda.device(device):
        check(lib().aix_synth_kmers_dev(seed, first, n, k, vp(t.data_ptr()), _stream_ptr()), "aix_synth_kmers_dev")
    return t


def synth_reads_t(seed: int, genome_t, n_reads: int, read_len: int, rc_half: bool = False, n_rate_ppm: int = 0,
                  first_read: int = 0):
    import torch
    t = torch.empty(n_reads * (read_len + 1), dtype=torch.uint8, device=genome_t.device)
    with torch.cuda.device(genome_t.device):
        check(lib().aix_synth_reads_dev(seed, vp(genome_t.data_ptr()), genome_t.numel(), first_read, n_reads, read_len,
                                        int(rc_half), n_rate_ppm, vp(t.data_ptr()), _stream_ptr()), "aix_synth_reads_dev")
    return t


def synth_mix23_t(seed: int, genome_t, n: int, first: int = 0):
    """Q_mix: 50 % genome windows on a random strand, 50 % uniform-random 23-mers, generated in HBM."""
    import torch
    t = torch.empty(n * 23 + 16, dtype=torch.uint8, device=genome_t.device)[: n * 23]
    with torch.cuda.device(genome_t.device):
        check(lib().aix_synth_mix23_dev(seed, vp(genome_t.data_ptr()), genome_t.numel(), first, n, vp(t.data_ptr()), _stream_ptr()),
              "aix_synth_mix23_dev")
    return t
