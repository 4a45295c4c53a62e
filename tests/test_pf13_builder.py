"""The product builder reproduces the reference's all-13-mers .pf byte for byte (sha256 pinned in
tests/golden/pf13.json, which was computed from the reference's own compute_mphf_seq output)."""
import hashlib
import json
import os

import pytest

from aindex_amd import builder


@pytest.mark.slow
def test_all_13mers_pf_sha(gold):
    g = json.load(open(os.path.join(gold, "pf13.json")))
    img = builder.build_all_13mers_pf()
    assert len(img) == g["size"]
    assert hashlib.sha256(img).hexdigest() == g["sha256"]
