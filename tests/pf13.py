"""Locate (or regenerate with the PRODUCT builder) the all-13-mers .pf used by the 13-mer paths."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PF13 = os.path.join(ROOT, "data", "all_13mers.pf")


def pf13_path():
    if not os.path.exists(PF13):
        from aindex_amd import builder  # product MWHC builder (C++), bit-identical to the reference's
        os.makedirs(os.path.dirname(PF13), exist_ok=True)
        builder.build_all_13mers_pf(PF13)
    return PF13


def pf13_sha_ok():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "pf13.json")))
    return hashlib.sha256(open(pf13_path(), "rb").read()).hexdigest() == g["sha256"]
