"""Command-line replacements for the reference's native tools (same argv contracts).

    python -m aindex_amd.tools count_kmers13 <input> <pf> <out_tf.bin> [threads]
    python -m aindex_amd.tools kmer_counter  <in.fa> <k> <out> [-t N] [-m min] [--canon refx86|true|none]
    python -m aindex_amd.tools compute_mphf_seq <keys.txt> [out.pf]
    python -m aindex_amd.tools compute_index <dat> <pf> <prefix> <threads> <mock>
    python -m aindex_amd.tools compute_aindex <reads> <pf> <prefix> <threads> 23 <tf> <kmers.bin> <kmers.txt> [index.bin indices.bin]
    python -m aindex_amd.tools compute_aindex13 <reads> <pf> <tf.bin> <prefix> <threads> [pos_bin] [index.bin] [indices.bin] [--tf-u32]
    python -m aindex_amd.tools compute_reads <file1> <file2|-> <fastq|fasta|se|reads> <prefix>     (host-side, no GPU)

`threads` arguments are accepted and ignored (the work runs on the GPU). Exit status 0 on success.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib, builder, counting
from ._lib import check, lib, vp
from .engine import Index


def _write(path, a: np.ndarray):
    """ndarray -> file through the library's mapped, multi-threaded writer (aix_file_write); the positions image of a few million reads is GBs."""
    a = np.ascontiguousarray(a)
    check(lib().aix_file_write(path.encode(), a.ctypes.data_as(vp), a.nbytes), f"aix_file_write({path})")


def _mapped(path):
    """The file as a read-only memory map (the library stages it through pinned memory part by part; no host copy is made here)."""
    import os
    return np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, dtype=np.uint8)


def count_kmers13(argv) -> int:
    """count_kmers13.cpp:546-566: writes 4^13 u64 counts in mphf order."""
    if len(argv) < 3:
        print("Usage: count_kmers13 <input_file> <hash_file> <output_tf_file> [num_threads]", file=sys.stderr)
        return 1
    import os
    import time
    t0 = time.perf_counter()
    lib().aix_ingest_warm(0)                                            # library load; the staging blocks are pinned while the index opens
    t1 = time.perf_counter()
    with Index.open_13(argv[1], None) as ix:                           # the input is streamed by the library: file -> pinned parts -> HBM -> table -> file
        t2 = time.perf_counter()
        _, st = ix.count13_file(argv[0], argv[2], _lib.FMT_AUTO, want_array=False)
        t3 = time.perf_counter()
    if os.environ.get("AIX_TOOL_TIMING"):
        print(f"count_kmers13 timing: library {t1 - t0:.3f} s, index open {t2 - t1:.3f} s, count + write {t3 - t2:.3f} s "
              f"(read {st['seconds_read']:.3f}, h2d {st['seconds_h2d']:.3f}, wait {st['seconds_wait']:.3f}, compute {st['seconds_compute']:.3f}, "
              f"output {st['seconds_output']:.3f}), close {time.perf_counter() - t3:.3f} s", file=sys.stderr)
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
    lib().aix_ingest_warm(0)
    keys, counts, _ = counting.count_distinct_file(argv[0], k, canon, min_count, _lib.FMT_FASTA)      # streamed, never a whole-file bytes object
    order = np.argsort(-counts.astype(np.int64), kind="stable")       # count descending (ties: key ascending)
    ks = np.ascontiguousarray(keys[order], dtype=np.uint64)
    cs = np.ascontiguousarray(counts[order], dtype=np.uint64)
    for path in {argv[2], "output.txt"}:                               # the text is written by the library (10^8 lines are no Python loop)
        check(lib().aix_kmers_write_text(path.encode(), ks.ctypes.data_as(vp), cs.ctypes.data_as(vp), ks.shape[0], k), "aix_kmers_write_text")
    return 0


def compute_mphf_seq(argv) -> int:
    """compute_mphf_generic.hpp:21-30: one key per line."""
    if len(argv) < 1:
        print("Expected: compute_mphf_seq <filename> [output_filename]", file=sys.stderr)
        return 1
    p, n = vp(), C.c_uint64()
    check(lib().aix_pf_build_file(argv[0].encode(), C.byref(p), C.byref(n)), "aix_pf_build_file")
    try:
        if len(argv) >= 2:
            with open(argv[1], "wb") as f:
                f.write(C.string_at(p, n.value))
    finally:
        lib().aix_free(p)
    return 0


def compute_index(argv) -> int:
    """compute_index.cpp:36-72: <dat> <pf> <prefix> <threads> <mock>."""
    if len(argv) < 5:
        print("Expected arguments: compute_index <dat_file> <pf_file> <output_prefix> <nthreads> <mock_flag>", file=sys.stderr)
        return 1
    mock = int(argv[4]) != 0
    nn, kp, tp = C.c_uint64(), vp(), vp()
    st = lib().aix_dat_load(argv[0].encode(), int(mock), C.byref(nn), C.byref(kp), C.byref(tp))      # the .dat is parsed by the library
    if st == _lib.AIX_ERR_FORMAT:
        print("compute_index: every key must be a 23-mer", file=sys.stderr)
        return 1
    check(st, "aix_dat_load")
    n = nn.value
    pf = _mapped(argv[1])
    checker, tf = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint32)
    try:
        st = lib().aix_index_scatter(pf.ctypes.data_as(vp), pf.shape[0], kp, tp if not mock else None, n, 0, checker.ctypes.data_as(vp), tf.ctypes.data_as(vp))
    finally:
        lib().aix_free(kp)
        if tp:
            lib().aix_free(tp)
    if st == -12:
        print("Conflict!!", file=sys.stderr)
        return 12                                                       # reference: exit(12)
    check(st, "aix_index_scatter")
    _write(argv[2] + ".kmers.bin", checker)
    _write(argv[2] + ".tf.bin", tf)
    return 0


def compute_reads(argv) -> int:
    """compute_reads.cpp:20-216: <file1> <file2|-> <fastq|fasta|se|reads> <output_prefix> -> .reads / .ridx / .header.
    Text reformatting bound by file I/O: done by the library's host routine aix_compute_reads (memory-mapped input, one pass,
    no device work: runs without a GPU)."""
    if len(argv) < 4:
        print("Expected arguments: compute_reads <fastq_file1|fasta_file1|reads_file> <fastq_file2|-> <fastq|fasta|se|reads> <output_prefix>",
              file=sys.stderr)
        return 1
    f1, f2, mode, prefix = argv[:4]
    import os
    from ._lib import lib, AIX_ERR_ARG
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    st = lib().aix_compute_reads(f1.encode(), f2.encode() if mode == "fastq" else None, mode.encode(), prefix.encode())
    if st == AIX_ERR_ARG:
        print("Unknown format.", file=sys.stderr)
        return 2
    if st != 0:
        print(f"compute_reads: cannot read {f1}" + (f" / {f2}" if mode == "fastq" else "") + f" or write {prefix}.*", file=sys.stderr)
        return 10
    return 0


def compute_aindex(argv) -> int:
    """compute_aindex.cpp:28-116: <reads> <pf> <prefix> <threads> <k> <tf> <kmers.bin> <kmers.txt> [..index.bin] [..indices.bin]
    (the reference reads argv[10]/argv[11] for the optional names, :61-62)."""
    if len(argv) < 7:
        print("Expected arguments: compute_aindex <reads_file> <hash_file> <output_prefix> <num_threads> <k> <tf_file> <kmers_bin_file> "
              "<kmers_text_file> [index_bin] [indices_bin]", file=sys.stderr)
        return 1
    reads_file, pf, prefix, _threads, k, tf_file, kmers_bin = argv[:7]
    if int(k) != 23:
        print("compute_aindex: only k = 23 is supported (the 13-mer branch of the reference is inconsistent, SURVEY row 9)", file=sys.stderr)
        return 1
    index_bin = argv[9] if len(argv) > 9 else prefix + ".index.bin"
    indices_bin = argv[10] if len(argv) > 10 else prefix + ".indices.bin"
    with Index.open_23(pf, tf_file, kmers_bin) as ix:
        indices, pos = ix.positions_fill(_mapped(reads_file))
    _write(index_bin, pos)
    _write(indices_bin, indices)
    return 0


def compute_aindex13(argv) -> int:
    """compute_aindex13.cpp:323-409: <reads> <pf> <tf.bin> <prefix> <threads> [pos_bin] [index_bin] [indices_bin] (the optional
    names are argv[7] / argv[8] there, :343-344). N3: positions index of the 13-mers, forward strand, tf = the u64[4^13]
    table count_kmers13 wrote. DIVERGENCE FROM THE REFERENCE BINARY, on purpose: that tool reads the u64 file as u32[4^13]
    (compute_aindex13.cpp:46-47) and so indexes a scrambled table; this one reads it as what it is, and its .index.bin / .indices.bin therefore
    differ from the reference tool's for the same inputs. `--tf-u32` (anywhere on the command line) or AIX_REF_COMPAT=1 selects the
    reference's reading and reproduces its files byte for byte (tests/golden/aindex13)."""
    if len(argv) < 5:
        print("Expected arguments: compute_aindex13 <reads_file> <hash_file> <tf_file> <output_prefix> <num_threads> [pos_bin] [index_bin] [indices_bin]",
              file=sys.stderr)
        return 1
    import os
    ref_compat = "--tf-u32" in argv or os.environ.get("AIX_REF_COMPAT") == "1"
    argv = [x for x in argv if x != "--tf-u32"]
    reads_file, pf, tf_file, prefix = argv[:4]
    index_bin = argv[6] if len(argv) > 6 else prefix + ".index.bin"
    indices_bin = argv[7] if len(argv) > 7 else prefix + ".indices.bin"
    with Index.open_13(pf, None if ref_compat else tf_file) as ix:
        if ref_compat:
            # the reference binary reads the first 4^13 u32 WORDS of the tf file (compute_aindex13.cpp:46-47: the u64 file of its own
            # count_kmers13 misread); `--tf-u32` / AIX_REF_COMPAT=1 indexes exactly that table and so writes the reference's files
            ix.set_tf_13(np.fromfile(tf_file, dtype=np.uint32, count=_lib.TOTAL_13MERS).astype(np.uint64))
        indices, pos = ix.positions_fill(_mapped(reads_file))
    _write(index_bin, pos)
    _write(indices_bin, indices)
    return 0


COMMANDS = {"compute_aindex13": compute_aindex13, "count_kmers13": count_kmers13, "kmer_counter": kmer_counter, "compute_mphf_seq": compute_mphf_seq, "compute_index": compute_index,
            "compute_reads": compute_reads, "compute_aindex": compute_aindex}


def main(argv=None) -> int:
    import os
    if "torch" not in sys.modules:
        os.environ.setdefault("AIX_NO_TORCH", "1")                     # a tool run is file -> GPU -> file through the C ABI: no tensors, no torch import
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in COMMANDS:
        print(__doc__, file=sys.stderr)
        return 2
    return COMMANDS[argv[0]](argv[1:])


if __name__ == "__main__":
    sys.exit(main())
