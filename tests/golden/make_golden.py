#!/usr/bin/env python3
"""Generate tests/golden/* from the COMPILED REFERENCE (oracle/_ref, built by `make -C oracle ref`
from the sources under /root/reference). Runs only in the build container; the fixtures it writes
are data (inputs + the reference's outputs), never reference source.

    python tests/golden/make_golden.py [--skip-13mer-pf]

What it pins:
  jenkins_kat.json      H1   emphf::jenkins64_hasher on 13/23-byte and odd-length strings
  codec_kat.json        C1/C2 get_dna23_bitset/get_dna13_bitset/reverseDNA incl. non-ACGT bytes
  small23/              K1 -> builder -> I1 -> Q1/Q2/Q4 + A1/A2 on a 400-read synthetic FASTA
  kmer_counter/         K1 output sets for k=23 and k=13 incl. lower-case and U
  pf13.json             H5/N1 sha256 + header of the all-13-mers .pf, 4096 sampled (13-mer, index)
  count13/              K13 sparse (index,count) lists for the reference's tests/data files and
                        a 2000-read synthetic file in FASTA / FASTQ / plain
  q13.json              Q3/Q4 13-mer query answers on the count13 synthetic table
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aindex_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
TMP = "/tmp/aix_golden"
REFDATA = "/root/reference/tests/data"


def run(cmd, cwd=None, stdin=None):
    r = subprocess.run(cmd, cwd=cwd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        sys.stderr.write(r.stderr.decode(errors="replace")[-2000:])
        raise SystemExit(f"{cmd} failed with {r.returncode}")
    return r.stdout


def probe(mode, lines, *args):
    out = run([os.path.join(REF, "ref_probe"), mode, *args], stdin=("\n".join(lines) + "\n").encode())
    return out.decode().split("\n")[:-1]


def import_ref_module():
    sys.path.insert(0, REF)
    import aindex_cpp  # the reference's pybind11 module, compiled by oracle/Makefile
    return aindex_cpp


def rng_strings(seed, n, length, alphabet=b"ACGT"):
    v = synth.sm64(seed, np.arange(n * length, dtype=np.uint64))
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return [bytes(a[(v[i * length:(i + 1) * length] % np.uint64(len(a))).astype(np.int64)]).decode() for i in range(n)]


# ---------------------------------------------------------------------------------------------
def make_jenkins():
    cases = []
    seeds = [0, 1, 0xF9E51456553305F9, 0xFFFFFFFFFFFFFFFF, 0x0123456789ABCDEF]
    strs = []
    strs += rng_strings(101, 24, 23)
    strs += rng_strings(102, 24, 13)
    strs += rng_strings(103, 8, 23, b"ACGTNacgt~")
    strs += ["A", "AC", "ACGTACG", "ACGTACGT", "ACGTACGTA", "ACGTACGTACGTACG", "ACGTACGTACGTACGT",
             "ACGTACGTACGTACGTA", "ACGTACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTACGTACGTA",
             "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTA"]
    lines = []
    for i, s in enumerate(strs):
        seed = seeds[i % len(seeds)]
        lines.append(f"{seed:x} {s}")
    outs = probe("jenkins", lines)
    for ln, o in zip(lines, outs):
        seed, s = ln.split()
        cases.append({"seed": seed, "s": s, "h": o.split()})
    json.dump(cases, open(os.path.join(GOLD, "jenkins_kat.json"), "w"), indent=0)
    print("jenkins_kat", len(cases))


def make_codec():
    s23 = rng_strings(201, 64, 23) + rng_strings(202, 32, 23, b"ACGTNacgtn") + ["A" * 23, "T" * 23, "C" * 23, "G" * 23]
    s13 = rng_strings(203, 64, 13) + rng_strings(204, 32, 13, b"ACGTNacgtn") + ["A" * 13, "T" * 13]
    e23 = [int(x) for x in probe("enc23", s23)]
    e13 = [int(x) for x in probe("enc13", s13)]
    r23 = [int(x) for x in probe("rev23", [str(x) for x in e23])]
    r13 = [int(x) for x in probe("rev13", [str(x) for x in e13])]
    json.dump({"s23": s23, "enc23": e23, "rev23": r23, "s13": s13, "enc13": e13, "rev13": r13},
              open(os.path.join(GOLD, "codec_kat.json"), "w"), indent=0)
    print("codec_kat", len(s23), len(s13))


# ---------------------------------------------------------------------------------------------
def write_fasta(path, reads, width=None, lower_every=0, u_every=0):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">read_%d some header text\n" % i)
            if lower_every and i % lower_every == 0:
                r = r.lower()
            if u_every and i % u_every == 0:
                r = r.replace(b"T", b"U")
            if width:
                for j in range(0, len(r), width):
                    f.write(r[j:j + width] + b"\n")
            else:
                f.write(r + b"\n")


def small_reads(seed, genome_len, n_reads, read_len, n_ppm):
    g = synth.genome_ascii(seed, genome_len)
    buf = synth.reads_plain(seed + 1, g, n_reads, read_len, rc_fraction_half=True, n_rate_ppm=n_ppm)
    return [bytes(x[:read_len]) for x in buf.reshape(n_reads, read_len + 1)]


def kmer_counter(fasta, k, workdir, threads=1, min_count=1):
    """reference kmer_counter always writes ./output.txt (count_kmers.cpp:359) -> run in workdir."""
    os.makedirs(workdir, exist_ok=True)
    run([os.path.join(REF, "kmer_counter"), fasta, str(k), "ignored", "-t", str(threads), "-m", str(min_count)], cwd=workdir)
    rows = [ln.split("\t") for ln in open(os.path.join(workdir, "output.txt")).read().split("\n") if ln]
    return [(a, int(b)) for a, b in rows]


def make_small23():
    d = os.path.join(GOLD, "small23")
    os.makedirs(d, exist_ok=True)
    w = os.path.join(TMP, "small23")
    shutil.rmtree(w, ignore_errors=True)
    os.makedirs(w)
    reads = small_reads(1, 3000, 400, 150, 2000)
    fa = os.path.join(d, "reads.fa")
    write_fasta(fa, reads)
    rows = kmer_counter(fa, 23, w)
    dat = os.path.join(w, "small23.dat")
    with open(dat, "w") as f:
        for kmer, c in rows:
            f.write(f"{kmer}\t{c}\n")
    keys = os.path.join(w, "keys.txt")
    with open(keys, "w") as f:
        for kmer, _ in rows:
            f.write(kmer + "\n")
    pf = os.path.join(d, "small23.pf")
    run([os.path.join(REF, "compute_mphf_seq"), keys, pf])
    run([os.path.join(REF, "compute_index"), dat, pf, os.path.join(d, "small23"), "1", "0"])
    # the .dat in the order the reference wrote it (text; input of I1 and of the builder)
    shutil.copy(dat, os.path.join(d, "small23.dat"))
    # reads file in the reference's .reads format (compute_reads, SE fasta -> one read per line)
    run([os.path.join(REF, "compute_reads"), fa, "-", "fasta", os.path.join(w, "small23")])
    reads_file = os.path.join(w, "small23.reads")
    shutil.copy(reads_file, os.path.join(d, "small23.reads"))
    # A1/A2 with one thread (slot order ascending)
    run([os.path.join(REF, "compute_aindex"), reads_file, pf, os.path.join(w, "small23"), "1", "23",
         os.path.join(d, "small23.tf.bin"), os.path.join(d, "small23.kmers.bin"), os.path.join(w, "none.txt")])
    idx = np.fromfile(os.path.join(w, "small23.index.bin"), dtype=np.uint64)
    ind = np.fromfile(os.path.join(w, "small23.indices.bin"), dtype=np.uint64)
    np.savez_compressed(os.path.join(d, "aindex.npz"), index=idx, indices=ind)

    # ---- queries through the reference's pybind module ----
    m = import_ref_module()
    wr = m.AindexWrapper()
    wr.load_from_prefix_23mer(os.path.join(d, "small23"))
    stored = [r[0] for r in rows]
    tfs = {r[0]: r[1] for r in rows}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))
    q = []
    q += stored[:300]                                   # stored strand
    q += [rc(s) for s in stored[300:600]]               # reverse strand
    q += rng_strings(301, 300, 23)                      # absent
    # N / lower-case on either strand
    for i, s in enumerate(stored[600:700]):
        p = (i * 7) % 23
        q.append(s[:p] + "N" + s[p + 1:])
        t = rc(s)
        q.append(t[:p] + "N" + t[p + 1:])
        q.append(s.lower())
        q.append(s[:p] + s[p].lower() + s[p + 1:])
    # N substituted for an A in stored / rc(stored): reproduces the N->A asymmetry
    cnt = 0
    for s in stored:
        for strand in (s, rc(s)):
            p = strand.find("A")
            if p >= 0:
                q.append(strand[:p] + "N" + strand[p + 1:])
                cnt += 1
        if cnt >= 200:
            break
    q += ["A" * 23, "T" * 23, "C" * 23, "G" * 23, "ACGT" * 5 + "ACG"]
    assert all(len(s) == 23 for s in q)
    exp = {
        "queries": q,
        "tf": list(wr.get_tf_values(q)),
        "tf_single": [wr.get_tf_value(s) for s in q[:50]],
        "total": list(wr.get_total_tf_values_23mer(q)),
        "both": [list(p) for p in wr.get_tf_both_directions_23mer_batch(q)],
        "hash": list(wr.get_hash_values(q)),
        "strand": [wr.get_strand(s) for s in q],
        "kid": [wr.get_kid_by_kmer(s) for s in q],
        "n_kmers": wr.n_kmers,
        "hash_size": wr.get_hash_size(),
    }
    # coverage (Q5): AIndex.get_sequence_coverage is pure python over get_tf_value (aindex.py:314-322)
    covs = []
    g = synth.genome_ascii(1, 3000).tobytes().decode()
    seqs = [g[100:400], rc(g[500:760]), reads[3].decode(), reads[7].decode().lower(), "ACGT" * 3, "A" * 23, "A" * 22]
    for s in seqs:
        for cutoff in (0, 3):
            cov = [0] * max(0, len(s) - 23 + 1)
            for i in range(len(s) - 23 + 1):
                tf = wr.get_tf_value(s[i:i + 23])
                if tf >= cutoff:
                    cov[i] = tf
            covs.append({"seq": s, "cutoff": cutoff, "k": 23, "cov": cov})
    exp["coverage"] = covs
    json.dump(exp, open(os.path.join(d, "queries.json"), "w"), indent=0)
    json.dump({"n": len(rows), "sum_tf": int(sum(tfs.values())), "index_len": int(idx.shape[0])},
              open(os.path.join(d, "meta.json"), "w"))
    print("small23 n =", len(rows), "queries =", len(q), "nonzero tf =", sum(1 for x in exp["tf"] if x))


def make_small23_access():
    """N2: positions / reads access of the reference's pybind module on the small23 pipeline (compute_reads ->
    compute_aindex re-run into TMP from the committed fixtures; the .ridx is committed because the mirror needs it)."""
    d = os.path.join(GOLD, "small23")
    w = os.path.join(TMP, "small23_access")
    shutil.rmtree(w, ignore_errors=True)
    os.makedirs(w)
    fa = os.path.join(d, "reads.fa")
    pf = os.path.join(d, "small23.pf")
    run([os.path.join(REF, "compute_reads"), fa, "-", "fasta", os.path.join(w, "small23")])
    assert open(os.path.join(w, "small23.reads"), "rb").read() == open(os.path.join(d, "small23.reads"), "rb").read()
    shutil.copy(os.path.join(w, "small23.ridx"), os.path.join(d, "small23.ridx"))
    run([os.path.join(REF, "compute_aindex"), os.path.join(w, "small23.reads"), pf, os.path.join(w, "small23"), "1", "23",
         os.path.join(d, "small23.tf.bin"), os.path.join(d, "small23.kmers.bin"), os.path.join(w, "none.txt")])
    m = import_ref_module()
    wr = m.AindexWrapper()
    wr.load_from_prefix_23mer(os.path.join(d, "small23"))
    wr.load_aindex_from_prefix_23mer(os.path.join(w, "small23"), 100, os.path.join(w, "small23.reads"))
    rows = [ln.split("\t") for ln in open(os.path.join(d, "small23.dat")).read().split("\n") if ln]
    stored = [r[0] for r in rows]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))
    # only k-mers get_pfid can find (hash.hpp:150-170 looks up the lexicographically smaller strand, so the STORED strand must
    # be that one): on anything else get_positions aborts the reference process
    ok = [s for s in stored if s <= rc(s)]
    kmers = ok[:80] + [rc(s) for s in ok[80:140]] + ok[-40:]
    pos = [list(wr.get_positions(s)) for s in kmers]
    probes = sorted({p for ps in pos for p in ps})[:300] + [0, 1, 149, 150, 151, 152, 10 ** 9]
    reads_size = wr.get_reads_size()
    exp = {
        "kmers": kmers, "positions": pos,
        "probes": probes, "rid": [wr.get_rid(p) for p in probes], "start": [wr.get_start(p) for p in probes],
        "n_reads": wr.n_reads, "reads_size": reads_size,
        "read_by_rid": {str(r): wr.get_read_by_rid(r) for r in (0, 1, 2, 57, 398, 399, 400, 10 ** 6)},
        "get_read": [[a, b, rcflag, wr.get_read(a, b, rcflag)] for a, b, rcflag in
                     ((0, 150, False), (0, 150, True), (151, 200, False), (300, 310, True), (5, 5, False), (10, 5, False),
                      (reads_size - 10, reads_size - 1, False), (reads_size - 1, reads_size, False), (reads_size, reads_size + 5, False))],
    }
    json.dump(exp, open(os.path.join(d, "access.json"), "w"), indent=0)
    print("small23 access: kmers", len(kmers), "positions", sum(len(x) for x in pos), "probes", len(probes))


def make_kmer_counter():
    d = os.path.join(GOLD, "kmer_counter")
    os.makedirs(d, exist_ok=True)
    reads = small_reads(11, 1500, 120, 100, 5000)
    fa = os.path.join(d, "mixed.fa")
    write_fasta(fa, reads, width=60, lower_every=3, u_every=5)
    for k in (23, 13):
        for mc in (1, 2):
            rows = kmer_counter(fa, k, os.path.join(TMP, f"kc{k}_{mc}"), threads=2, min_count=mc)
            rows.sort()
            with open(os.path.join(d, f"mixed.k{k}.m{mc}.tsv"), "w") as f:
                for kmer, c in rows:
                    f.write(f"{kmer}\t{c}\n")
            print("kmer_counter k", k, "min", mc, "distinct", len(rows))


# ---------------------------------------------------------------------------------------------
def make_pf13(skip_build):
    pf = os.path.join(ROOT, "data", "all_13mers.pf")
    os.makedirs(os.path.dirname(pf), exist_ok=True)
    if not (skip_build and os.path.exists(pf)):
        os.makedirs(TMP, exist_ok=True)
        txt = os.path.join(TMP, "all_13mers.txt")
        n = 4 ** 13
        with open(txt, "wb") as f:     # same text the reference's generate_all_13mers emits (2-bit order)
            step = 1 << 20
            for lo in range(0, n, step):
                a = synth.decode_kmers(np.arange(lo, lo + step, dtype=np.uint64), 13)
                out = np.empty((step, 14), dtype=np.uint8)
                out[:, :13] = a
                out[:, 13] = 10
                f.write(out.tobytes())
        run([os.path.join(REF, "compute_mphf_seq"), txt, pf])
        os.remove(txt)
    raw = open(pf, "rb").read()
    sha = hashlib.sha256(raw).hexdigest()
    hdr = np.frombuffer(raw[:32], dtype=np.uint64)
    codes = (synth.sm64(1313, np.arange(4096, dtype=np.uint64)) & np.uint64(4 ** 13 - 1))
    codes[:4] = [0, 1, 4 ** 13 - 1, 4 ** 13 - 2]
    kmers = [bytes(x).decode() for x in synth.decode_kmers(codes, 13)]
    idx = [int(x) for x in probe("lookup", kmers, pf)]
    json.dump({"sha256": sha, "size": len(raw), "n": int(hdr[0]), "D": int(hdr[1]), "seed": f"{int(hdr[2]):x}",
               "B": int(hdr[3]), "codes": [int(c) for c in codes], "index": idx},
              open(os.path.join(GOLD, "pf13.json"), "w"), indent=0)
    print("pf13 sha", sha, "size", len(raw))
    return pf


def sparse_counts(path):
    a = np.fromfile(path, dtype=np.uint64)
    assert a.shape[0] == 4 ** 13
    nz = np.nonzero(a)[0]
    return nz.astype(np.uint64), a[nz]


def make_count13(pf):
    d = os.path.join(GOLD, "count13")
    os.makedirs(d, exist_ok=True)
    w = os.path.join(TMP, "count13")
    os.makedirs(w, exist_ok=True)
    inputs = {}
    for name in ("test.fasta", "test_se.fastq", "test_reads.txt", "test_unknown.txt", "test_R1.fastq"):
        dst = os.path.join(d, "refdata_" + name)       # the reference's own tiny test inputs (data fixtures)
        shutil.copy(os.path.join(REFDATA, name), dst)
        inputs["refdata_" + name] = dst
    reads = small_reads(21, 20000, 2000, 150, 3000)
    # sprinkle lower-case and a '~' PE separator so every reader branch is exercised
    reads = [r.lower() if i % 11 == 0 else r for i, r in enumerate(reads)]
    p = os.path.join(d, "synth.fa")
    write_fasta(p, reads, width=70)
    inputs["synth.fa"] = p
    p = os.path.join(d, "synth.fq")
    with open(p, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    inputs["synth.fq"] = p
    p = os.path.join(d, "synth.txt")
    with open(p, "wb") as f:
        for i in range(0, len(reads), 2):
            f.write(reads[i] + b"~" + reads[i + 1] + b"\n")
        f.write(b"\nACGT\nACGTACGTACGTA")          # empty line, short line, no trailing newline
    inputs["synth.txt"] = p
    out = {}
    for name, path in inputs.items():
        tfb = os.path.join(w, name + ".tf.bin")
        run([os.path.join(REF, "count_kmers13"), path, pf, tfb, "2"])
        idx, cnt = sparse_counts(tfb)
        out[name] = (idx, cnt)
        print("count13", name, "nonzero", idx.shape[0], "total", int(cnt.sum()))
    np.savez_compressed(os.path.join(d, "expected.npz"), **{k + ".idx": v[0] for k, v in out.items()},
                        **{k + ".cnt": v[1] for k, v in out.items()})

    # ---- 13-mer queries (Q3/Q4) on the synth.fa table through the pybind module ----
    prefix = os.path.join(w, "q13")
    shutil.copy(pf, prefix + ".pf")
    shutil.copy(os.path.join(w, "synth.fa.tf.bin"), prefix + ".tf.bin")
    m = import_ref_module()
    wr = m.AindexWrapper()
    wr.load_from_prefix_13mer(prefix)
    q = []
    g = synth.genome_ascii(21, 20000).tobytes().decode()
    q += [g[i:i + 13] for i in range(0, 3000, 13)]
    q += rng_strings(401, 200, 13)
    q += [s.lower() for s in q[:20]] + ["ACGTNACGTACGT", "A" * 13, "T" * 13, "ACGT", "ACGTACGTACGTAC"]
    valid = [s for s in q if len(s) == 13 and set(s) <= set("ACGT")]
    exp = {
        "queries": q,
        "tf": list(wr.get_tf_values(q)),
        "valid": valid,
        "total": list(wr.get_total_tf_values_13mer(valid)),
        "both": [list(p) for p in wr.get_tf_both_directions_13mer_batch(valid)],
        "by_index": [[int(i), int(wr.get_tf_by_index_13mer(int(i)))] for i in out["synth.fa"][0][:64]],
    }
    covs = []
    for s in (g[50:250], g[1000:1100].lower(), "ACGTACGTACGT", reads[1].decode()):
        for cutoff in (0, 2):
            cov = [0] * max(0, len(s) - 13 + 1)
            for i in range(len(s) - 13 + 1):
                tf = wr.get_tf_value(s[i:i + 13])
                if tf >= cutoff:
                    cov[i] = tf
            covs.append({"seq": s, "cutoff": cutoff, "k": 13, "cov": cov})
    exp["coverage"] = covs
    json.dump(exp, open(os.path.join(GOLD, "q13.json"), "w"), indent=0)
    print("q13 queries", len(q), "nonzero", sum(1 for x in exp["tf"] if x))


def make_aindex13(pf):
    """N3: the reference's compute_aindex13 (1 thread) on count13/synth.txt with the tf file its own count_kmers13 wrote.
    The tool reads that u64 file as u32 (compute_aindex13.cpp:46-47), so its output is the positions index of the MISREAD
    table tf32[i] = i-th 32-bit word of the file; the tests hand that view (widened to u64) to our implementation and to the
    oracle restatement and expect the reference's files. Fixture: the positions array and a digest of the 537 MB indices file."""
    import hashlib
    d = os.path.join(GOLD, "aindex13")
    os.makedirs(d, exist_ok=True)
    w = os.path.join(TMP, "aindex13")
    os.makedirs(w, exist_ok=True)
    reads = os.path.join(GOLD, "count13", "synth.txt")
    tfb = os.path.join(w, "synth.tf.bin")
    run([os.path.join(REF, "count_kmers13"), reads, pf, tfb, "2"])
    prefix = os.path.join(w, "a13")
    run([os.path.join(REF, "compute_aindex13"), reads, pf, tfb, prefix, "1"])
    pos = np.fromfile(prefix + ".index.bin", dtype=np.uint64)
    h = hashlib.sha256()
    with open(prefix + ".indices.bin", "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    ind = np.fromfile(prefix + ".indices.bin", dtype=np.uint64)
    view = np.fromfile(tfb, dtype=np.uint32)[: 4 ** 13]
    assert int(ind[-1]) == pos.shape[0] == int(view.astype(np.uint64).sum())
    np.savez_compressed(os.path.join(d, "synth.npz"), positions=pos, indices_sha256=np.frombuffer(h.digest(), dtype=np.uint8),
                        total=np.array([pos.shape[0]], dtype=np.uint64), indices_sample=ind[:: 1 << 16])
    print("aindex13: total slots", pos.shape[0], "filled", int((pos != 0).sum()))
    for f in (tfb, prefix + ".index.bin", prefix + ".indices.bin"):
        os.remove(f)


def make_compute_reads():
    """N4: compute_reads outputs (.reads/.ridx/.header) for every input mode, on the reference's own tiny inputs."""
    d = os.path.join(GOLD, "compute_reads")
    os.makedirs(d, exist_ok=True)
    w = os.path.join(TMP, "compute_reads")
    shutil.rmtree(w, ignore_errors=True)
    os.makedirs(w)
    cases = {
        "pe": (os.path.join(REFDATA, "test_R1.fastq"), os.path.join(REFDATA, "test_R2.fastq"), "fastq"),
        "se": (os.path.join(REFDATA, "test_se.fastq"), "-", "se"),
        "fasta": (os.path.join(REFDATA, "test.fasta"), "-", "fasta"),
        "reads": (os.path.join(REFDATA, "test_reads.txt"), "-", "reads"),
        "fasta_multi": (os.path.join(GOLD, "count13", "synth.fa"), "-", "fasta"),
    }
    for name, (f1, f2, mode) in cases.items():
        for f in (f1, f2):
            if f != "-" and not f.startswith(GOLD):
                shutil.copy(f, os.path.join(d, "in_" + os.path.basename(f)))
        run([os.path.join(REF, "compute_reads"), f1, f2, mode, os.path.join(w, name)])
        for ext in (".reads", ".ridx", ".header"):
            src = os.path.join(w, name + ext)
            if os.path.exists(src):
                shutil.copy(src, os.path.join(d, name + ext))
        print("compute_reads", name, [e for e in (".reads", ".ridx", ".header") if os.path.exists(os.path.join(w, name + e))])


def make_compute_reads_edges():
    """N4, malformed / awkward inputs (ours, tiny): truncated records with and without a final newline (std::getline leaves or erases
    the string it failed to fill, and compute_reads.cpp goes on using it), a mate file that is shorter than its partner, lower case /
    N / IUPAC letters in the mate that is reverse-complemented, FASTA with empty lines, empty records, text before the first header
    and no final newline, a plain reads file with empty lines, and empty files. Inputs and the reference's outputs are committed."""
    d = os.path.join(GOLD, "compute_reads")
    w = os.path.join(TMP, "compute_reads_edges")
    shutil.rmtree(w, ignore_errors=True)
    os.makedirs(w)
    rec = lambda name, seq, q="I": "@%s\n%s\n+\n%s\n" % (name, seq, q * len(seq))
    files = {
        "edge_se_trunc_nl.fastq": rec("a", "ACGTACGT") + rec("b", "TTGACA") + "@c\n",
        "edge_se_trunc_nonl.fastq": rec("a", "ACGTACGT") + rec("b", "TTGACA") + "@header_only",
        "edge_se_noplus.fastq": rec("a", "ACGTACGT") + "@b\nGGGTTT",
        "edge_pe_R1.fastq": rec("p1", "ACGTACGTAA") + rec("p2", "CCCCGGGGTT") + rec("p3", "TTTTAAAACC"),
        "edge_pe_R2_short_nonl.fastq": rec("p1", "acgtNNRYacgtTTGA", "F") + rec("p2", "GATTACAGATTACA", "#")[:-1],
        "edge_pe_R2_iupac.fastq": rec("p1", "ACGTNRYKMSWBDHVacgtn-*") + rec("p2", "") + rec("p3", "A"),
        "edge_multi.fasta": "stray line before any header\n>first one\nACGT\n\nAC GT\n>\n>empty record above\n>third\tx\nTTTT\n>last\nGG\nCC",
        "edge_reads.txt": "ACGT\n\nAC~GT\nTTTT",
        "edge_empty.txt": "",
    }
    for name, text in files.items():
        open(os.path.join(d, name), "w").write(text)
    cases = {
        "edge_se_trunc_nl": ("edge_se_trunc_nl.fastq", "-", "se"),
        "edge_se_trunc_nonl": ("edge_se_trunc_nonl.fastq", "-", "se"),
        "edge_se_noplus": ("edge_se_noplus.fastq", "-", "se"),
        "edge_pe_short": ("edge_pe_R1.fastq", "edge_pe_R2_short_nonl.fastq", "fastq"),
        "edge_pe_iupac": ("edge_pe_R1.fastq", "edge_pe_R2_iupac.fastq", "fastq"),
        "edge_pe_r1_short": ("edge_pe_R2_short_nonl.fastq", "edge_pe_R1.fastq", "fastq"),
        "edge_fasta": ("edge_multi.fasta", "-", "fasta"),
        "edge_reads": ("edge_reads.txt", "-", "reads"),
        "edge_empty_se": ("edge_empty.txt", "-", "se"),
        "edge_empty_pe": ("edge_empty.txt", "edge_pe_R1.fastq", "fastq"),
        "edge_empty_fasta": ("edge_empty.txt", "-", "fasta"),
        "edge_empty_reads": ("edge_empty.txt", "-", "reads"),
    }
    for name, (f1, f2, mode) in cases.items():
        run([os.path.join(REF, "compute_reads"), os.path.join(d, f1), f2 if f2 == "-" else os.path.join(d, f2), mode, os.path.join(w, name)])
        got = []
        for ext in (".reads", ".ridx", ".header"):
            src = os.path.join(w, name + ext)
            if os.path.exists(src):
                shutil.copy(src, os.path.join(d, name + ext))
                got.append(ext)
        print("compute_reads", name, got)
    json.dump(cases, open(os.path.join(d, "edge_cases.json"), "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-13mer-pf", action="store_true", help="reuse data/all_13mers.pf if present")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(TMP, exist_ok=True)
    only = set(a.only.split(",")) if a.only else None
    if not only or "jenkins" in only:
        make_jenkins()
    if not only or "codec" in only:
        make_codec()
    if not only or "small23" in only:
        make_small23()
    if not only or "small23_access" in only:
        make_small23_access()
    if not only or "kmer_counter" in only:
        make_kmer_counter()
    if not only or "compute_reads" in only:
        make_compute_reads()
        make_compute_reads_edges()
    if not only or "13" in only:
        pf = make_pf13(a.skip_13mer_pf)
        make_count13(pf)
    if not only or "aindex13" in only:
        make_aindex13(os.path.join(ROOT, "data", "all_13mers.pf"))       # the all-13-mers .pf (sha256 pinned in pf13.json)
