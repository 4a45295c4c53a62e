"""N4: `compute_reads` replacement (host-side text reformatting) against the compiled reference's outputs."""
import os

import pytest

from aindex_amd import tools

CASES = {
    "pe": ("in_test_R1.fastq", "in_test_R2.fastq", "fastq"),
    "se": ("in_test_se.fastq", "-", "se"),
    "fasta": ("in_test.fasta", "-", "fasta"),
    "reads": ("in_test_reads.txt", "-", "reads"),
    "fasta_multi": ("../count13/synth.fa", "-", "fasta"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_compute_reads_matches_reference(gold, tmp_path, name):
    d = os.path.join(gold, "compute_reads")
    f1, f2, mode = CASES[name]
    f1 = os.path.join(d, f1)
    f2 = f2 if f2 == "-" else os.path.join(d, f2)
    prefix = str(tmp_path / name)
    assert tools.main(["compute_reads", f1, f2, mode, prefix]) == 0
    checked = 0
    for ext in (".reads", ".ridx", ".header"):
        want = os.path.join(d, name + ext)
        if os.path.exists(want):
            assert open(prefix + ext, "rb").read() == open(want, "rb").read(), ext
            checked += 1
        else:
            assert not os.path.exists(prefix + ext)
    assert checked >= 1


def _edge_cases(gold_dir):
    import json
    return json.load(open(os.path.join(gold_dir, "compute_reads", "edge_cases.json")))


def test_compute_reads_edge_inputs_match_reference(gold, tmp_path):
    """Malformed and awkward inputs (tests/golden/make_golden.py: make_compute_reads_edges): truncated FASTQ records with and without
    a final newline (a failed std::getline erases or keeps the string the reference's loop goes on using), a mate file shorter than
    its partner, lower case / N / IUPAC letters in the reverse-complemented mate, FASTA with empty lines, empty records, text before
    the first header and no final newline, plain reads with empty lines, empty files: byte-equal to what the compiled reference wrote."""
    d = os.path.join(gold, "compute_reads")
    cases = _edge_cases(gold)
    assert len(cases) >= 12
    for name, (f1, f2, mode) in sorted(cases.items()):
        prefix = str(tmp_path / name)
        assert tools.main(["compute_reads", os.path.join(d, f1), f2 if f2 == "-" else os.path.join(d, f2), mode, prefix]) == 0, name
        for ext in (".reads", ".ridx", ".header"):
            want = os.path.join(d, name + ext)
            if os.path.exists(want):
                assert open(prefix + ext, "rb").read() == open(want, "rb").read(), (name, ext)
            else:
                assert not os.path.exists(prefix + ext), (name, ext)


def test_compute_reads_statuses(tmp_path):
    """Unknown mode -> the reference's "Unknown format." status; unreadable input -> a non-zero status, nothing half-written is reported as success."""
    p = str(tmp_path / "x.fa")
    open(p, "w").write(">a\nACGT\n")
    assert tools.main(["compute_reads", p, "-", "bam", str(tmp_path / "o1")]) == 2
    assert tools.main(["compute_reads", str(tmp_path / "missing.fa"), "-", "fasta", str(tmp_path / "o2")]) != 0
    assert tools.main(["compute_reads", p, str(tmp_path / "missing_R2.fq"), "fastq", str(tmp_path / "o3")]) != 0
    assert tools.main(["compute_reads", p, "-", "fasta", str(tmp_path / "sub" / "dir" / "o4")]) == 0 and os.path.exists(str(tmp_path / "sub" / "dir" / "o4.reads"))


def test_tool_argument_contracts():
    """Usage errors of the argv front ends return the reference's non-zero statuses without touching a GPU
    (count_kmers13.cpp:546-566, count_kmers.cpp:394-414, compute_index.cpp:36-49, compute_aindex.cpp:30-63)."""
    from aindex_amd import tools
    assert tools.main([]) == 2 and tools.main(["no_such_tool"]) == 2
    assert tools.main(["count_kmers13", "reads.fa"]) == 1
    assert tools.main(["kmer_counter", "reads.fa", "23"]) == 1
    assert tools.main(["compute_mphf_seq"]) == 1
    assert tools.main(["compute_index", "a.dat", "a.pf", "prefix"]) == 1
    assert tools.main(["compute_aindex", "r", "pf", "prefix", "1", "23"]) == 1
    assert tools.main(["compute_aindex", "r", "pf", "prefix", "1", "13", "tf", "kmers.bin"]) == 1      # only k = 23
    assert tools.main(["compute_reads", "a.fa", "-", "fasta"]) == 1
