"""Config 4 two ways on one index: the probe + histogram path (count23_fixed_t) against distinct k-mers of the reads first (count_distinct_t, true
canonical form) — what a 'count, then look up each distinct k-mer once' back end would cost."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from aindex_amd import engine, counting, _lib

ix, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
reads = engine.synth_reads_t(41, g, n_reads, 150, rc_half=True, n_rate_ppm=1000)


def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


t_probe, tf = timed(lambda: ix.count23_fixed_t(reads, _lib.CANON_TRUE_RC))
t_k1, (dk, dc) = timed(lambda: counting.count_distinct_t(reads, 23, _lib.CANON_TRUE_RC))
t_look, _ = timed(lambda: ix.tf_codes_t(dk), 3)
print(json.dumps({"reads": n_reads, "probe_histogram_ms": round(t_probe, 1), "distinct_ms": round(t_k1, 1), "distinct_keys": int(dk.numel()), "lookup_of_distinct_ms": round(t_look, 2),
                  "sum_tf": int(tf.to(torch.int64).sum().item()), "sum_counts": int(dc.to(torch.int64).sum().item())}))
