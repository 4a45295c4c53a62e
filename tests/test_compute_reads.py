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
