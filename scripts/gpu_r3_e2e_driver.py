"""One process = one setting of the ingestion pipeline (the AIX_* switches are read from the environment): count13_file on a given PLAIN file,
best of three calls, one line of JSON. usage: gpu_r3_e2e_driver.py <file> [pinned]"""
import json, os, sys, time
os.environ.setdefault("AIX_NO_TORCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from aindex_amd.engine import Index
from aindex_amd.builder import all_13mers_pf_path
from aindex_amd import _lib
path = sys.argv[1]
size = os.path.getsize(path)
ix = Index.open_13(all_13mers_pf_path(), None)
best = None
out = np.empty(4 ** 13, dtype=np.uint64)
for _ in range(3):
    st = _lib.IngestStats()
    _lib.check(_lib.lib().aix_count13_file(ix._h, path.encode(), 0, None, out.ctypes.data_as(_lib.vp), _lib.C.byref(st)))
    d = st.as_dict()
    if best is None or d["seconds_total"] < best["seconds_total"]:
        best = d
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("AIX_")}, "GBps": size / best["seconds_total"] / 1e9, "sum": int(out.sum()), **{k: round(v, 4) if isinstance(v, float) else v for k, v in best.items()}}))
