"""Latency of tiny batches through the Python mirror and the ABI (the reference answers a single get_tf_value in ~0.6 us on the CPU)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aindex_amd.aindex import AIndex  # noqa: E402

prefix = os.path.join(ROOT, "tests", "golden", "small23", "small23")
ai = AIndex.load_from_prefix(prefix)
q = json.load(open(os.path.join(ROOT, "tests", "golden", "small23", "queries.json")))["queries"]
ix = ai._wrapper._ix23
res = {}
import numpy as _np
seq = "".join(q[:435])[:10000]
for name, fn, reps in (("AIndex.get_sequence_coverage(10 kbp)", lambda i: ai.get_sequence_coverage(seq, 0, 23), 2000),
                       ("AIndex.get_sequence_coverage(150 bp)", lambda i: ai.get_sequence_coverage(seq[:150], 0, 23), 5000),
                       ("AIndex[kmer] (1 query)", lambda i: ai[q[i % 1000]], 20000),
                       ("AIndex.get_tf_values(8 kmers)", lambda i: ai.get_tf_values(q[:8]), 20000),
                       ("Index.tf_ascii(1 x 23 bytes)", lambda i: ix.tf_ascii(q[i % 1000].encode()), 20000),
                       ("Index.tf_ascii(1024 x 23 bytes)", lambda i: ix.tf_ascii("".join(q[:1024]).encode()), 5000)):
    for i in range(200):
        fn(i)
    t = time.perf_counter()
    for i in range(reps):
        fn(i)
    dt = (time.perf_counter() - t) / reps
    res[name] = {"us_per_call": dt * 1e6}
    print(name, "%.1f us" % (dt * 1e6))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "latency.json"), "w"), indent=1)
