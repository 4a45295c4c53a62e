"""Locate (or regenerate with the PRODUCT builder) the all-13-mers .pf used by the 13-mer paths."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PF13 = os.path.join(ROOT, "data", "all_13mers.pf")


def pf13_path():
    from aindex_amd import builder  # product MWHC builder (C++), bit-identical to the reference's
    return builder.all_13mers_pf_path(PF13)


def pf13_sha_ok():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "pf13.json")))
    return hashlib.sha256(open(pf13_path(), "rb").read()).hexdigest() == g["sha256"]
