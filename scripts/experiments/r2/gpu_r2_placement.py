"""Does the Q_mix spread (2.47 ... 2.80 ms from process to process) follow WHERE the index lands in HBM? One process, one set of queries, several
handles built from the same .pf / keys / counts (each its own allocations, the earlier ones kept alive so that addresses differ): kernel time per handle."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aindex_amd import engine
from aindex_amd.engine import Index

ix0, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
N = 100_000_000
q = engine.synth_mix23_t(8, g, N)
out = torch.empty(N, dtype=torch.int32, device="cuda:0")
handles, pads, res = [ix0], [], []
for i in range(4):
    pads.append(torch.empty((37 + 61 * i) << 20, dtype=torch.uint8, device="cuda:0"))       # shift what follows by an odd number of MiB
    handles.append(Index.build_23_codes_t(pf, keys, counts32, 0))
for rep in range(2):
    for i, ix in enumerate(handles):
        ix.tf_ascii_t(q, out); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ix.tf_ascii_t(q, out)
        b.record(); torch.cuda.synchronize()
        res.append({"rep": rep, "handle": i, "kernel_ms": a.elapsed_time(b) / 5})
print(json.dumps(res))
for r in res:
    print(r, file=sys.stderr)
