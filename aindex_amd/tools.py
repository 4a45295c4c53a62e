"""Command-line replacements for the reference's native tools (same argv contracts).

    python -m aindex_amd.tools count_kmers13 <input> <pf> <out_tf.bin> [threads]
    python -m aindex_amd.tools kmer_counter  <in.fa> <k> <out> [-t N] [-m min] [--canon refx86|true|none]
    python -m aindex_amd.tools compute_mphf_seq <keys.txt> [out.pf]
    python -m aindex_amd.tools compute_index <dat> <pf> <prefix> <threads> <mock>

`threads` arguments are accepted and ignored (the work runs on the GPU). Exit status 0 on success.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib, builder, counting
from ._lib import check, lib, vp
from .engine import Index


def count_kmers13(argv) -> int:
    """count_kmers13.cpp:546-566: writes 4^13 u64 counts in mphf order."""
    if len(argv) < 3:
        print("Usage: count_kmers13 <input_file> <hash_file> <output_tf_file> [num_threads]", file=sys.stderr)
        return 1
    buf = open(argv[0], "rb").read()
    with Index.open_13(argv[1], None) as ix:
        ix.count13(buf, _lib.FMT_AUTO).tofile(argv[2])
    return 0


def kmer_counter(argv) -> int:
    """count_kmers.cpp:394-414. The reference ignores <out> and writes ./output.txt; both are written here."""
    if len(argv) < 3:
        print("Usage: kmer_counter <input_file> <k> <output_file> [-t threads] [-m min_count]", file=sys.stderr)
        return 1
    k, min_count, canon = int(argv[1]), 1, _lib.CANON_REF_X86
    i = 3
    while i < len(argv):
        if argv[i] == "-m" and i + 1 < len(argv):
            min_count = int(argv[i + 1]); i += 1
        elif argv[i] == "-t" and i + 1 < len(argv):
            i += 1
        elif argv[i] == "--canon" and i + 1 < len(argv):
            canon = {"refx86": 1, "true": 2, "none": 0}[argv[i + 1]]; i += 1
        i += 1
    keys, counts = counting.count_distinct(open(argv[0], "rb").read(), k, canon, min_count, _lib.FMT_FASTA)
    order = np.argsort(-counts.astype(np.int64), kind="stable")       # count descending (ties: key ascending)
    from .synth import decode_kmers
    kmers = decode_kmers(keys[order], k)
    lines = b"".join(bytes(a) + b"\t" + str(int(c)).encode() + b"\n" for a, c in zip(kmers, counts[order]))
    for path in {argv[2], "output.txt"}:
        with open(path, "wb") as f:
            f.write(lines)
    return 0


def compute_mphf_seq(argv) -> int:
    """compute_mphf_generic.hpp:21-30: one key per line."""
    if len(argv) < 1:
        print("Expected: compute_mphf_seq <filename> [output_filename]", file=sys.stderr)
        return 1
    keys = open(argv[0], "rb").read().split(b"\n")
    if keys and keys[-1] == b"":
        keys.pop()
    pf = builder.build_pf(keys)
    if len(argv) >= 2:
        with open(argv[1], "wb") as f:
            f.write(pf)
    return 0


def compute_index(argv) -> int:
    """compute_index.cpp:36-72: <dat> <pf> <prefix> <threads> <mock>."""
    if len(argv) < 5:
        print("Expected arguments: compute_index <dat_file> <pf_file> <output_prefix> <nthreads> <mock_flag>", file=sys.stderr)
        return 1
    mock = int(argv[4]) != 0
    rows = [ln.split() for ln in open(argv[0], "rb").read().split(b"\n") if ln]
    if any(len(r[0]) != 23 for r in rows):
        print("compute_index: every key must be a 23-mer", file=sys.stderr)
        return 1
    keys = np.frombuffer(b"".join(r[0] for r in rows), dtype=np.uint8)
    tfs = None if mock else np.array([int(r[1]) for r in rows], dtype=np.uint32)
    pf = np.frombuffer(open(argv[1], "rb").read(), dtype=np.uint8)
    n = len(rows)
    checker, tf = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint32)
    st = lib().aix_index_scatter(pf.ctypes.data_as(vp), pf.shape[0], keys.ctypes.data_as(vp), tfs.ctypes.data_as(vp) if tfs is not None else None,
                                 n, 0, checker.ctypes.data_as(vp), tf.ctypes.data_as(vp))
    if st == -12:
        print("Conflict!!", file=sys.stderr)
        return 12                                                       # reference: exit(12)
    check(st, "aix_index_scatter")
    checker.tofile(argv[2] + ".kmers.bin")
    tf.tofile(argv[2] + ".tf.bin")
    return 0


COMMANDS = {"count_kmers13": count_kmers13, "kmer_counter": kmer_counter, "compute_mphf_seq": compute_mphf_seq, "compute_index": compute_index}


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in COMMANDS:
        print(__doc__, file=sys.stderr)
        return 2
    return COMMANDS[argv[0]](argv[1:])


if __name__ == "__main__":
    sys.exit(main())
