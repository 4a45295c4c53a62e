"""Deterministic synthetic inputs: genomes, reads, k-mer queries.

Host (numpy) mirror of the device generators in csrc/aix_synth.hip — both compute exactly the
same bytes from the same seeds, so tests can build small cases on the host and the benchmark can
build the full-size ones directly in HBM. PRNG = counter-based splitmix64:

    sm64(seed, i) = finalise(seed + (i + 1) * 0x9E3779B97F4A7C15)

i.e. the i-th output of a splitmix64 stream started at `seed`.
"""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)
_ASCII_RC = np.frombuffer(b"TGCA", dtype=np.uint8)


def sm64(seed: int, idx) -> np.ndarray:
    """splitmix64 output number `idx` (array ok) of the stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        idx = np.asarray(idx, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (idx + np.uint64(1)) * GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def genome_codes(seed: int, length: int) -> np.ndarray:
    """2-bit base codes (uint8 in 0..3): base j = (sm64(seed, j // 32) >> 2*(j % 32)) & 3."""
    nwords = (length + 31) // 32
    w = sm64(seed, np.arange(nwords, dtype=np.uint64))
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    codes = ((w[:, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(-1)
    return codes[:length]


def genome_ascii(seed: int, length: int) -> np.ndarray:
    return _ASCII[genome_codes(seed, length)]


def random_kmers_ascii(seed: int, n: int, k: int, start: int = 0) -> np.ndarray:
    """n uniform random k-mers as an (n, k) uint8 ASCII array.

    k-mer i has 2-bit code sm64(seed, start + i) & (4^k - 1), first base = most significant pair.
    """
    v = sm64(seed, np.arange(start, start + n, dtype=np.uint64))
    shifts = (np.uint64(2) * np.arange(k - 1, -1, -1, dtype=np.uint64))[None, :]
    return _ASCII[((v[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)]


def reads_plain(seed: int, genome: np.ndarray, n_reads: int, read_len: int, *,
                rc_fraction_half: bool = False, n_rate_ppm: int = 0, first_read: int = 0) -> np.ndarray:
    """Plain one-read-per-line buffer: n_reads * (read_len + 1) bytes, each read followed by '\\n'.

    `genome` is the uint8 ASCII genome. Read r (global index first_read + r):
      v = sm64(seed, 2r); start = ((v >> 32) * (L - read_len + 1)) >> 32
      reverse-complemented iff rc_fraction_half and (sm64(seed, 2r + 1) & 1)
      base j replaced by 'N' iff n_rate_ppm and (sm64(seed ^ 0x5851F42D4C957F2D, r * read_len + j) >> 32) * 1_000_000 >> 32 < n_rate_ppm
    """
    L = int(genome.shape[0])
    assert L >= read_len
    r = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
    v = sm64(seed, r * np.uint64(2))
    span = np.uint64(L - read_len + 1)
    start = ((v >> np.uint64(32)) * span) >> np.uint64(32)
    idx = start[:, None] + np.arange(read_len, dtype=np.uint64)[None, :]
    fwd = genome[idx]
    out = np.empty((n_reads, read_len + 1), dtype=np.uint8)
    if rc_fraction_half:
        w = sm64(seed, r * np.uint64(2) + np.uint64(1))
        is_rc = (w & np.uint64(1)).astype(bool)
        comp = np.zeros(256, dtype=np.uint8)
        comp[:] = np.arange(256, dtype=np.uint8)
        comp[ord("A")], comp[ord("C")], comp[ord("G")], comp[ord("T")] = ord("T"), ord("G"), ord("C"), ord("A")
        rc = comp[fwd[:, ::-1]]
        out[:, :read_len] = np.where(is_rc[:, None], rc, fwd)
    else:
        out[:, :read_len] = fwd
    if n_rate_ppm:
        pos = r[:, None] * np.uint64(read_len) + np.arange(read_len, dtype=np.uint64)[None, :]
        h = sm64(seed ^ 0x5851F42D4C957F2D, pos)
        hit = (((h >> np.uint64(32)) * np.uint64(1_000_000)) >> np.uint64(32)) < np.uint64(n_rate_ppm)
        out[:, :read_len][hit] = ord("N")
    out[:, read_len] = ord("\n")
    return out.reshape(-1)


# ---------------------------------------------------------------------------------------------
# host-side 2-bit helpers (array codec; used to build key sets for tests / bench)
# ---------------------------------------------------------------------------------------------
def encode_kmers(ascii_kmers: np.ndarray) -> np.ndarray:
    """(n, k) uint8 ASCII (A/C/G/T only) -> uint64 codes, first base most significant."""
    lut = np.zeros(256, dtype=np.uint64)
    lut[ord("C")], lut[ord("G")], lut[ord("T")] = 1, 2, 3
    k = ascii_kmers.shape[1]
    code = np.zeros(ascii_kmers.shape[0], dtype=np.uint64)
    for j in range(k):
        code = (code << np.uint64(2)) | lut[ascii_kmers[:, j]]
    return code


def decode_kmers(codes: np.ndarray, k: int) -> np.ndarray:
    shifts = (np.uint64(2) * np.arange(k - 1, -1, -1, dtype=np.uint64))[None, :]
    return _ASCII[((np.asarray(codes, dtype=np.uint64)[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)]


def revcomp_codes(codes: np.ndarray, k: int) -> np.ndarray:
    codes = np.asarray(codes, dtype=np.uint64)
    r = np.zeros_like(codes)
    c = codes.copy()
    for _ in range(k):
        r = (r << np.uint64(2)) | (np.uint64(3) - (c & np.uint64(3)))
        c >>= np.uint64(2)
    return r


def rolling_codes(seq_codes: np.ndarray, k: int) -> np.ndarray:
    """All k-window codes of a 2-bit base array (len - k + 1 uint64)."""
    n = seq_codes.shape[0] - k + 1
    out = np.zeros(n, dtype=np.uint64)
    s = seq_codes.astype(np.uint64)
    for j in range(k):
        out = (out << np.uint64(2)) | s[j:j + n]
    return out


def canonical_distinct(genome_codes_arr: np.ndarray, k: int):
    """Sorted distinct true-canonical k-mer codes of a genome and their multiplicities."""
    w = rolling_codes(genome_codes_arr, k)
    c = np.minimum(w, revcomp_codes(w, k))
    keys, counts = np.unique(c, return_counts=True)
    return keys, counts.astype(np.uint32)
