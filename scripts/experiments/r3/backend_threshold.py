"""count23's automatic choice between the probe path (back end 2) and "distinct k-mers first" (back end 3) at the shares of config 4 on 8, 4 and 2
GPUs (25 / 50 / 100 M reads against 5e7 keys = 64 / 128 / 256 windows per key) and below, both forced, one handle."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from aindex_amd import engine, _lib

ix, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
res = []
for n_reads in (12_500_000, 25_000_000, 50_000_000, 100_000_000):
    reads = engine.synth_reads_t(41, g, n_reads, 150, rc_half=True, n_rate_ppm=1000)
    row = {"reads": n_reads, "windows_per_key": round(n_reads * 128 / ix.n, 1)}
    for force in ("0", "1"):
        os.environ["AIX_COUNT23_VIA_K1"] = force
        ix.count23_fixed_t(reads, _lib.CANON_TRUE_RC); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ix.count23_fixed_t(reads, _lib.CANON_TRUE_RC)
        torch.cuda.synchronize()
        row["backend_%d_ms" % ix.info["count23_backend"]] = round((time.perf_counter() - t0) / 3 * 1e3, 1)
    del os.environ["AIX_COUNT23_VIA_K1"]
    res.append(row); print(row, flush=True)
    del reads
print(json.dumps(res))
