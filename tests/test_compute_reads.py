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
