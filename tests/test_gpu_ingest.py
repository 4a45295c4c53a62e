"""GPU parity of the STREAMING counting path (aix_count13_file / aix_count23_fixed_file / aix_count_distinct_file and the host-buffer twins
that now run the same pipeline): parts cut at every possible byte, FASTA / FASTQ / PLAIN / CRLF, against the reference's golden outputs,
the oracle, and the device-resident entry points. Bit-exact. Nothing here reads /root/reference.
(reference: count_kmers13.cpp:166-183,211-272,277-350 streams its file through a reader thread; count_kmers.cpp:242-341 reads it whole.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as O
from aindex_amd import _lib, counting, synth
from aindex_amd.engine import Index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ix13():
    from pf13 import pf13_path
    ix = Index.open_13(pf13_path(), None)
    yield ix
    ix.close()


@pytest.fixture(scope="module")
def ix23(small23_prefix):
    ix = Index.open_23(small23_prefix + ".pf", small23_prefix + ".tf.bin", small23_prefix + ".kmers.bin")
    yield ix
    ix.close()


def sparse(counts):
    nz = np.nonzero(counts)[0]
    return nz.astype(np.uint64), counts[nz]


TINY = ["refdata_test.fasta", "refdata_test_se.fastq", "refdata_test_reads.txt", "refdata_test_unknown.txt", "refdata_test_R1.fastq"]
SYNTH = ["synth.fa", "synth.fq", "synth.txt"]


@pytest.mark.parametrize("name", TINY + SYNTH)
def test_count13_file_golden_at_every_cut(ix13, gold, name, monkeypatch, tmp_path):
    """The reference's own tests/data inputs and the synthetic FASTA / FASTQ / PLAIN files, streamed in parts of a few bytes up to the
    default: the file the tool writes and the array the call returns equal the reference's count_kmers13 output."""
    z = np.load(os.path.join(gold, "count13", "expected.npz"))
    path = os.path.join(gold, "count13", name)
    size = os.path.getsize(path)
    parts = [1, 2, 3, 5, 12, 13, 37, 64, 65] if name in TINY else [257, 4093, 65536]
    for part in parts + [0]:
        if part:
            monkeypatch.setenv("AIX_INGEST_TEST_PART", str(part))
        else:
            monkeypatch.delenv("AIX_INGEST_TEST_PART", raising=False)
        out = str(tmp_path / f"tf_{part}.bin")
        counts, st = ix13.count13_file(path, out)
        idx, cnt = sparse(counts)
        assert np.array_equal(idx, z[name + ".idx"]) and np.array_equal(cnt, z[name + ".cnt"]), (name, part)
        assert os.path.getsize(out) == 8 * 4 ** 13
        assert np.array_equal(np.fromfile(out, dtype=np.uint64), counts)
        assert st["bytes_in"] == size and st["parts"] == (-(-size // part) if part else (1 if size else 0))


def test_count13_stream_crlf_long_lines_and_empty(ix13, monkeypatch, tmp_path):
    """CRLF files ('\\r' stays in the sequence as in the reference's getline and breaks windows), a FASTA record that spans many parts,
    a FASTQ whose last line has no newline, an empty file; file path == host-buffer path == device-resident path == oracle."""
    from pf13 import pf13_path
    m = O.OracleMphf(pf13_path())
    rng = np.random.default_rng(5)
    seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), 40_000, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04]))
    fa = b">r1 long record\r\n" + b"\r\n".join(seq[i:i + 61] for i in range(0, 20_000, 61)) + b"\r\n>r2\n" + b"\n".join(seq[i:i + 70] for i in range(20_000, 40_000, 70))
    fq = b"".join(b"@q%d\n" % i + seq[i * 100:i * 100 + 100] + b"\n+\n" + b"@" * 100 + b"\n" for i in range(300))[:-1]
    txt = b"\n".join(seq[i:i + 150] for i in range(0, 40_000, 150)) + b"\r\n\r\n" + seq[:12] + b"\n" + seq[:13]
    cases = {"fa": fa, "fq": fq, "txt": txt, "empty": b"", "short": b"ACGTACGTACGT", "nl": b"\n\n\n"}
    for name, buf in cases.items():
        want = O.count13(m, buf)
        p = str(tmp_path / name)
        open(p, "wb").write(buf)
        for part in (0, 7, 1021):
            if part:
                monkeypatch.setenv("AIX_INGEST_TEST_PART", str(part))
            else:
                monkeypatch.delenv("AIX_INGEST_TEST_PART", raising=False)
            got, _ = ix13.count13_file(p)
            assert np.array_equal(got, want), (name, part, "file")
            assert np.array_equal(ix13.count13(buf), want), (name, part, "buffer")


def test_count13_stream_million_reads_memory_is_independent_of_file_size(ix13, tmp_path, monkeypatch):
    """1 M and 4 M synthetic reads (PLAIN and FASTQ on disk) streamed in 16 MiB parts: equal to the device-resident count of the same reads,
    HBM and pinned staging of the call do not grow with the file."""
    import torch
    from aindex_amd import engine
    monkeypatch.setenv("AIX_INGEST_PART_MB", "16")
    g = engine.synth_genome_t(13, 2_000_000)
    seen = {}
    for n_reads in (1_000_000, 4_000_000):
        reads = engine.synth_reads_t(14, g, n_reads, 150, rc_half=False, n_rate_ppm=1000)
        want = ix13.count13_t(reads).cpu().numpy().view(np.uint64)
        host = reads.cpu().numpy()
        p = str(tmp_path / f"r{n_reads}.txt")
        host.tofile(p)
        got, st = ix13.count13_file(p)
        assert np.array_equal(got, want)
        seen[n_reads] = st
        if n_reads == 1_000_000:
            lines = host.reshape(n_reads, 151)[:, :150]
            fq = np.empty((n_reads, 4 + 151 + 2 + 151), dtype=np.uint8)
            fq[:, :4] = np.frombuffer(b"@r1\n", dtype=np.uint8)
            fq[:, 4:154] = lines
            fq[:, 154] = 10
            fq[:, 155:157] = np.frombuffer(b"+\n", dtype=np.uint8)
            fq[:, 157:307] = ord("I")
            fq[:, 307] = 10
            pq = str(tmp_path / "r.fq")
            fq.tofile(pq)
            gq, stq = ix13.count13_file(pq)
            assert np.array_equal(gq, want)
            assert stq["plain_bytes"] == n_reads * 151
        del reads
        torch.cuda.empty_cache()
    a, b = seen[1_000_000], seen[4_000_000]
    assert b["bytes_in"] == 4 * a["bytes_in"] and b["parts"] >= 4 * a["parts"] - 3
    assert a["device_bytes"] == b["device_bytes"] and a["pinned_bytes"] == b["pinned_bytes"] == 3 * (16 << 20)


def test_count23_fixed_file_equals_buffer_and_oracle(ix23, gold, small23_prefix, monkeypatch, tmp_path):
    """config-4 histogram fed from a file: the reference pipeline's own reads (FASTA; kmer_counter rules: '>' anywhere opens a record) in
    all three canonical modes, parts of 1 ... 4 KiB bytes; also the histogram back end on small parts (AIX_COUNT23_HIST_MIN=0)."""
    orc = O.OracleIndex23.from_prefix(small23_prefix)
    path = os.path.join(gold, "small23", "reads.fa")
    buf = open(path, "rb").read()
    for canon in (0, 1, 2):
        want = orc.count23_fixed(buf, True, canon)
        for part, hist in ((0, None), (1, None), (22, None), (23, "0"), (4099, "0"), (4099, None)):
            if part:
                monkeypatch.setenv("AIX_INGEST_TEST_PART", str(part))
            else:
                monkeypatch.delenv("AIX_INGEST_TEST_PART", raising=False)
            if hist is None:
                monkeypatch.delenv("AIX_COUNT23_HIST_MIN", raising=False)
            else:
                monkeypatch.setenv("AIX_COUNT23_HIST_MIN", hist)
            if part == 1 and canon != 1:
                continue                                            # 60 K one-byte parts: once is enough
            got, st = ix23.count23_fixed_file(path, _lib.FMT_AUTO, canon)
            assert np.array_equal(got, want), (canon, part, hist)
            assert np.array_equal(ix23.count23_fixed(buf, _lib.FMT_AUTO, canon), want), (canon, part, hist, "buffer")
    # canon 1 on these reads is what kmer_counter -> compute_index stored
    assert np.array_equal(orc.count23_fixed(buf, True, 1), orc.tf_array())
    # PLAIN and FASTQ forms of the same reads
    recs = [r.split(b"\n", 1)[1].replace(b"\n", b"") for r in buf.split(b">") if r]
    txt = b"\n".join(recs) + b"\n"
    fq = b"".join(b"@x\n" + r + b"\n+\n" + b"F" * len(r) + b"\n" for r in recs)
    want = orc.count23_fixed(buf, True, 2)
    for name, data in (("r.txt", txt), ("r.fq", fq)):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        for part in (0, 313):
            if part:
                monkeypatch.setenv("AIX_INGEST_TEST_PART", str(part))
            else:
                monkeypatch.delenv("AIX_INGEST_TEST_PART", raising=False)
            got, _ = ix23.count23_fixed_file(p, _lib.FMT_AUTO, 2)
            assert np.array_equal(got, want), (name, part)


def test_count_distinct_file_pieces_and_parts(gold, monkeypatch, tmp_path):
    """kmer_counter from a file: the reference's golden sets (k = 23 / 13, min_count 1 / 2, lower-case and U input) and the oracle, with the
    input cut into parts of 1 ... 4 KiB bytes AND the PLAIN stream cut into pieces of 50 ... 5 000 windows (every piece boundary carries
    k - 1 bytes; the sorted sets of the pieces are merged by the two-way merge)."""
    path = os.path.join(gold, "kmer_counter", "mixed.fa")
    fa = open(path, "rb").read()
    for k in (23, 13):
        for mc in (1, 2):
            rows = [ln.split("\t") for ln in open(os.path.join(gold, "kmer_counter", f"mixed.k{k}.m{mc}.tsv")).read().split("\n") if ln]
            want = sorted((r[0], int(r[1])) for r in rows)
            for part, piece in ((0, 0), (1, 0), (29, 50), (997, 333), (4096, 5000), (0, 64)):
                if part == 1 and (k, mc) != (23, 1):
                    continue
                for name, v in (("AIX_INGEST_TEST_PART", part), ("AIX_DISTINCT_PIECE", piece)):
                    if v:
                        monkeypatch.setenv(name, str(v))
                    else:
                        monkeypatch.delenv(name, raising=False)
                keys, counts, st = counting.count_distinct_file(path, k, _lib.CANON_REF_X86, mc)
                got = sorted((bytes(a).decode(), int(c)) for a, c in zip(synth.decode_kmers(keys, k), counts))
                assert got == want, (k, mc, part, piece)
                if piece:
                    assert st["pieces"] > 1
                kb, cb = counting.count_distinct(fa, k, _lib.CANON_REF_X86, mc)
                assert np.array_equal(kb, keys) and np.array_equal(cb, counts)
    monkeypatch.setenv("AIX_INGEST_TEST_PART", "1500")
    monkeypatch.setenv("AIX_DISTINCT_PIECE", "700")
    for k, canon in ((23, 2), (17, 0), (31, 2), (9, 1)):
        keys, counts, _ = counting.count_distinct_file(path, k, canon, 1)
        okeys, ocnt = O.count_distinct(fa, k, canon, 1)
        assert np.array_equal(keys, okeys) and np.array_equal(counts, ocnt), (k, canon)


def test_merge_runs_equals_sort_and_reduce():
    """aix_merge_runs_dev (tree of two-way merges with summation) against numpy on runs of every shape: empty runs, one run, runs that
    share all / none of their keys, a pair split across a tile boundary of the merge kernel (2 048 entries), 64-bit counts."""
    import torch
    rng = np.random.default_rng(11)

    def check(runs, min_count=1):
        ks = np.concatenate([r[0] for r in runs]) if runs else np.zeros(0, np.uint64)
        cs = np.concatenate([r[1] for r in runs]) if runs else np.zeros(0, np.uint64)
        offs = np.cumsum([0] + [len(r[0]) for r in runs]).astype(np.uint64)
        uk, inv = np.unique(ks, return_inverse=True)
        uc = np.zeros(uk.shape[0], dtype=np.uint64)
        np.add.at(uc, inv, cs)
        keep = uc >= min_count
        kt = torch.from_numpy(ks.view(np.int64)).cuda()
        ct = torch.from_numpy(cs.view(np.int64)).cuda()
        gk, gc = counting.merge_runs_t(kt, ct, offs, min_count)
        assert np.array_equal(gk.cpu().numpy().view(np.uint64), uk[keep]) and np.array_equal(gc.cpu().numpy().view(np.uint64), uc[keep])
        gk2, gc2 = counting.merge_counts_t(kt, ct, min_count)
        assert np.array_equal(gk2.cpu().numpy().view(np.uint64), uk[keep]) and np.array_equal(gc2.cpu().numpy().view(np.uint64), uc[keep])

    def run(n, hi, big=False):
        k = np.unique(rng.integers(0, hi, size=n, dtype=np.uint64))
        c = rng.integers(1, 1 << (40 if big else 8), size=k.shape[0], dtype=np.uint64)
        return k, c

    e = (np.zeros(0, np.uint64), np.zeros(0, np.uint64))
    check([run(1000, 1 << 46)])
    check([e, run(10, 100), e])
    check([run(5000, 6000), run(5000, 6000)])                      # dense overlap
    check([run(5000, 1 << 46), run(7000, 1 << 46)], 2)             # no overlap: everything below min_count 2 except the rare pair
    a = np.arange(0, 3000, dtype=np.uint64)
    check([(a, np.ones(3000, np.uint64)), (a, np.full(3000, 1 << 40, np.uint64))])     # identical key sets: every tile boundary splits a pair
    check([(a[:2047], np.ones(2047, np.uint64)), (a[2046:2050], np.ones(4, np.uint64))])
    check([run(20000, 30000, True) for _ in range(8)], 3)
    check([run(int(rng.integers(0, 9000)), 1 << 20) for _ in range(5)] + [e])
    check([run(300_000, 1 << 46), run(300_000, 400_000), run(1, 5)])


def test_count_kmers13_tool_streams_the_file(gold, tmp_path):
    """bin/count_kmers13 <in> <pf> <out> as a process (the tool path: no torch import, file -> pinned parts -> HBM -> file): byte-equal to
    the reference's output for its own tests/data input and the synthetic FASTQ."""
    from pf13 import pf13_path
    z = np.load(os.path.join(gold, "count13", "expected.npz"))
    for name in ("refdata_test.fasta", "synth.fq"):
        out = str(tmp_path / (name + ".tf.bin"))
        env = dict(os.environ, AIX_INGEST_TEST_PART="3001")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "count_kmers13"), os.path.join(gold, "count13", name), pf13_path(), out, "4"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-400:]
        idx, cnt = sparse(np.fromfile(out, dtype=np.uint64))
        assert np.array_equal(idx, z[name + ".idx"]) and np.array_equal(cnt, z[name + ".cnt"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "count_kmers13"), str(tmp_path / "missing.fa"), pf13_path(), str(tmp_path / "x")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
