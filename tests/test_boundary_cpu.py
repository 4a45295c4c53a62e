"""Boundary checks that need no GPU: the executable shims under bin/ (the names aindex/cli.py:run_executable spawns,
cli.py:214-278), the .pf header validation of the ABI, and the host-side reads loaders of the AindexWrapper mirror."""
import ctypes as C
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from aindex_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")
SHIMS = ("count_kmers13", "kmer_counter", "compute_mphf_seq", "compute_index", "compute_aindex", "compute_aindex13", "compute_reads")


def _run(name, *args, cwd=None):
    return subprocess.run([os.path.join(BIN, name), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=cwd, timeout=300)


@pytest.mark.parametrize("name", SHIMS)
def test_shim_is_an_executable_with_the_reference_name_and_usage_status(name):
    """cli.py:243-262 looks for <bin>/<name> and runs it; without arguments every reference tool prints its usage and exits
    non-zero (count_kmers13.cpp:546-551, count_kmers.cpp:394-399, compute_mphf_generic.hpp:21-26, compute_index.cpp:36-41,
    compute_aindex.cpp:30-35, compute_aindex13.cpp:324-338, compute_reads.cpp:24-29)."""
    path = os.path.join(BIN, name)
    assert os.path.isfile(path) and os.access(path, os.X_OK)
    r = _run(name)
    assert r.returncode == 1, r.stderr
    assert name in r.stderr.decode() or "Expected" in r.stderr.decode()


def test_compute_reads_shim_reproduces_the_reference_files(gold, tmp_path):
    d = os.path.join(gold, "compute_reads")
    prefix = str(tmp_path / "se")
    r = _run("compute_reads", os.path.join(d, "in_test_se.fastq"), "-", "se", prefix)
    assert r.returncode == 0, r.stderr
    for ext in (".reads", ".ridx"):
        assert open(prefix + ext, "rb").read() == open(os.path.join(d, "se" + ext), "rb").read()
    assert _run("compute_reads", os.path.join(d, "in_test_se.fastq"), "-", "nonsense", prefix).returncode == 2     # compute_reads.cpp:214


def test_compute_mphf_seq_shim_writes_the_reference_pf(gold, small23_prefix, tmp_path):
    """`compute_mphf_seq <keys> <out.pf>` through the shim (host MWHC builder, no GPU): byte-identical to the .pf the
    compiled reference wrote for the same keys (tests/golden/small23/small23.pf)."""
    keys = str(tmp_path / "keys.txt")
    with open(keys, "w") as f:
        for ln in open(small23_prefix + ".dat"):
            f.write(ln.split("\t")[0] + "\n")
    out = str(tmp_path / "out.pf")
    r = _run("compute_mphf_seq", keys, out)
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open(small23_prefix + ".pf", "rb").read()


def _pf_check(buf: bytes):
    hdr = (C.c_uint64 * 4)()
    a = np.frombuffer(buf, dtype=np.uint8) if buf else np.zeros(1, np.uint8)
    return _lib.lib().aix_pf_check(a.ctypes.data_as(C.c_void_p), len(buf), C.byref(hdr)), list(hdr)


def test_pf_header_validation(small23_prefix):
    """A corrupt header must be refused on the host, before any upload (D = 0x5555555555555556, B = 2 passes a
    wrapping `B == 3 * D` and would index the record table at ~2^62 on the device)."""
    good = open(small23_prefix + ".pf", "rb").read()
    st, hdr = _pf_check(good)
    n, D, seed, B = struct.unpack("<4Q", good[:32])
    assert st == 0 and hdr == [n, D, seed, B] and B == 3 * D
    fmt = -3
    assert _pf_check(good[:31])[0] == fmt                                            # shorter than the header
    assert _pf_check(good[:-8])[0] == fmt                                            # block ranks cut short
    wrap = struct.pack("<4Q", 1, 0x5555555555555556, seed, 2) + good[32:]
    assert (3 * 0x5555555555555556) % (1 << 64) == 2 and _pf_check(wrap)[0] == fmt   # 3 * D wraps around to B
    assert _pf_check(struct.pack("<4Q", n, D, seed, B + 1) + good[32:])[0] == fmt    # B != 3 D
    assert _pf_check(struct.pack("<4Q", B + 1, D, seed, B) + good[32:])[0] == fmt    # more keys than bit-pairs
    big_d = (1 << 32) // 3 + 1
    assert _pf_check(struct.pack("<4Q", 1, big_d, seed, 3 * big_d) + good[32:])[0] == fmt   # node ids beyond 32 bits
    assert _pf_check(struct.pack("<4Q", 0, 0, seed, 0))[0] == 0                      # the MPHF of an empty key set (mphf.hpp:26)
    assert _lib.lib().aix_pf_check(None, 64, None) == -1


def test_load_reads_in_memory_and_mapped_agree_with_the_reference_answers(gold, small23_prefix, tmp_path):
    """load_reads (python_wrapper.cpp:281-322, mmap) and load_reads_in_memory (:324-359, private copy) give the same reads
    access; answers = those of the compiled reference (tests/golden/small23/access.json). Host-side only."""
    from aindex_amd.wrapper import AindexWrapper
    a = json.load(open(os.path.join(gold, "small23", "access.json")))
    reads = str(tmp_path / "copy.reads")
    with open(reads, "wb") as f:
        f.write(open(small23_prefix + ".reads", "rb").read())
    with open(str(tmp_path / "copy.ridx"), "wb") as f:
        f.write(open(small23_prefix + ".ridx", "rb").read())
    for loader in ("load_reads", "load_reads_in_memory"):
        w = AindexWrapper()
        getattr(w, loader)(reads)
        assert w.n_reads == a["n_reads"] and w.reads_size == a["reads_size"] and w.get_reads_size() == a["reads_size"]
        for rid, read in a["read_by_rid"].items():
            assert w.get_read_by_rid(int(rid)) == read
        for s0, e0, rcflag, want in a["get_read"]:
            assert w.get_read(s0, e0, rcflag) == want
    # the in-memory copy is private: truncating the file afterwards does not show through (:349-351 reads it all up front)
    w = AindexWrapper()
    w.load_reads_in_memory(reads)
    first = w.get_read_by_rid(0)
    open(reads, "wb").close()
    assert w.get_read_by_rid(0) == first == a["read_by_rid"]["0"]
    w2 = AindexWrapper()
    w2.load_reads_in_memory(str(tmp_path / "missing.reads"))                          # :337-340: message, early return
    assert w2.reads_size == 0 and w2.get_read_by_rid(0) == ""


def test_host_normaliser_against_reference_written_reads_files(gold):
    """aix_normalize_reads (host side of the readers, count_kmers13.cpp:211-257) pinned directly to reference outputs: the
    `.reads` files the compiled reference's compute_reads wrote for the same inputs are the PLAIN form, one sequence per line
    (compute_reads.cpp:118-147 FASTQ line 4i+1, :170-213 FASTA records concatenated over their lines). The device kernels are
    held to the same files in tests/test_gpu_parity.py::test_normalisers_against_reference_written_reads_files."""
    from aindex_amd import counting
    d = os.path.join(gold, "compute_reads")
    for src, want, fmt in (("in_test.fasta", "fasta.reads", 1), ("../count13/synth.fa", "fasta_multi.reads", 1), ("in_test_se.fastq", "se.reads", 2)):
        raw = open(os.path.join(d, src), "rb").read()
        assert counting.normalize(raw, fmt, 0) == open(os.path.join(d, want), "rb").read(), src


def test_native_text_formats_of_the_tools(tmp_path):
    """Host routines behind the tools' text files (no GPU): the .dat parser of compute_index (`is >> kmer >> tf`, src/hash.cpp:681-702:
    any blank separates, a missing / unreadable count is 0, one beyond u32 saturates, mock reads k-mers only, empty lines skipped, a
    k-mer of another length is a format error), the k-mer list writer of kmer_counter (count_kmers.cpp:362-382) and the keys-file
    front end of the MWHC builder (== the builder called on the same keys)."""
    import ctypes as C
    import numpy as np
    from aindex_amd import _lib, builder
    from aindex_amd._lib import lib, vp
    L = lib()
    k1, k2, k3, k4 = b"ACGTACGTACGTACGTACGTACG", b"TTTTTTTTTTTTTTTTTTTTTTT", b"GATTACAGATTACAGATTACAGA", b"CCCCCCCCCCCCCCCCCCCCCCC"
    dat = tmp_path / "a.dat"
    dat.write_bytes(k1 + b"\t7\n" + b"\n" + b"  " + k2 + b" \t 4294967295\n" + k3 + b"\n" + k4 + b"\t99999999999 trailing\n" + k1[::-1] + b"\tx12")
    n, kp, tp = C.c_uint64(), vp(), vp()
    assert L.aix_dat_load(str(dat).encode(), 0, C.byref(n), C.byref(kp), C.byref(tp)) == 0 and n.value == 5
    keys = C.string_at(kp, 23 * 5)
    tfs = np.frombuffer(C.string_at(tp, 4 * 5), dtype=np.uint32).tolist()
    L.aix_free(kp); L.aix_free(tp)
    assert keys == k1 + k2 + k3 + k4 + k1[::-1] and tfs == [7, 4294967295, 0, 4294967295, 0]
    n2, kp2 = C.c_uint64(), vp()
    assert L.aix_dat_load(str(dat).encode(), 1, C.byref(n2), C.byref(kp2), None) == 0 and n2.value == 5
    assert C.string_at(kp2, 23 * 5) == keys
    L.aix_free(kp2)
    bad = tmp_path / "bad.dat"
    bad.write_bytes(k1 + b"\t1\nACGT\t2\n")
    assert L.aix_dat_load(str(bad).encode(), 0, C.byref(n), C.byref(kp), C.byref(tp)) == _lib.AIX_ERR_FORMAT
    assert L.aix_dat_load(str(tmp_path / "missing.dat").encode(), 0, C.byref(n), C.byref(kp), C.byref(tp)) != 0
    # writer: 2-bit codes, first base most significant
    codes = np.array([0, 1, (1 << 46) - 1, 0b011011], dtype=np.uint64)
    counts = np.array([5, 18446744073709551615, 1, 0], dtype=np.uint64)
    out = tmp_path / "k.txt"
    assert L.aix_kmers_write_text(str(out).encode(), codes.ctypes.data_as(vp), counts.ctypes.data_as(vp), 4, 23) == 0
    assert out.read_bytes() == b"A" * 23 + b"\t5\n" + b"A" * 22 + b"C\t18446744073709551615\n" + b"T" * 23 + b"\t1\n" + b"A" * 20 + b"CGT\t0\n"
    assert L.aix_kmers_write_text(str(out).encode(), codes.ctypes.data_as(vp), counts.ctypes.data_as(vp), 2, 3) == 0 and out.read_bytes() == b"AAA\t5\nAAC\t18446744073709551615\n"
    # keys file -> .pf == the builder on the same keys (any lengths, no final newline)
    keysf = tmp_path / "keys.txt"
    ks = [b"ACGT", b"A", b"TTGACA", b"GATTACA", k1]
    keysf.write_bytes(b"\n".join(ks))
    p, ln = vp(), C.c_uint64()
    assert L.aix_pf_build_file(str(keysf).encode(), C.byref(p), C.byref(ln)) == 0
    img = C.string_at(p, ln.value)
    L.aix_free(p)
    assert img == builder.build_pf(ks)
    # .ridx: numbers separated by any white space, reading stops at the first thing that is no number, an incomplete triple is no read
    r = tmp_path / "a.ridx"
    r.write_bytes(b"0\t0\t150\n1\t151\t301\n 2 302\n452\n3\t453\n")
    nn, pp = C.c_uint64(), vp()
    assert L.aix_ridx_load(str(r).encode(), C.byref(nn), C.byref(pp)) == 0 and nn.value == 3
    assert np.frombuffer(C.string_at(pp, 24 * 3), dtype=np.uint64).tolist() == [0, 0, 150, 1, 151, 301, 2, 302, 452]
    L.aix_free(pp)
    r.write_bytes(b"0\t0\t150\nx\t1\t2\n")
    assert L.aix_ridx_load(str(r).encode(), C.byref(nn), C.byref(pp)) == 0 and nn.value == 1
    L.aix_free(pp)
    r.write_bytes(b"")
    assert L.aix_ridx_load(str(r).encode(), C.byref(nn), C.byref(pp)) == 0 and nn.value == 0
    L.aix_free(pp)
