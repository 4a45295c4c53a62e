"""K1 at config-4 size: are the keys strictly ascending (no repeats), and is every one of them a key of the index the reads were drawn from?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from aindex_amd import engine, counting, _lib

ix, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
out = {}
for n_reads in (5_000_000, 60_000_000, 200_000_000):
    reads = engine.synth_reads_t(41, g, n_reads, 150, rc_half=True, n_rate_ppm=1000)
    dk, dc = counting.count_distinct_t(reads, 23, _lib.CANON_TRUE_RC)
    d = dk[1:] - dk[:-1]
    tf = ix.tf_codes_t(dk)
    missing = dk[tf == 0]
    out[n_reads] = {"distinct": int(dk.numel()), "not_ascending": int((d <= 0).sum().item()), "repeats": int((d == 0).sum().item()), "not_in_index": int(missing.numel()),
                    "first_missing": [hex(int(x)) for x in missing[:4].tolist()], "counts_of_missing": dc[tf == 0][:4].tolist(),
                    "index_keys_ascending": int(((keys[1:] - keys[:-1]) <= 0).sum().item())}
    print(n_reads, out[n_reads], flush=True)
    del reads, dk, dc, tf
print(json.dumps(out))
