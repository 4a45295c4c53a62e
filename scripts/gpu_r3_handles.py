"""Q_mix kernel time of several handles with identical contents in ONE process (each handle its own allocations; odd-sized pads in between).
argv[1] = number of handles (default 4), argv[2] = launches per handle (default 5). Prints one line per handle to stderr, JSON to stdout.
Under `rocprofv3 --kernel-trace [--pmc ...]` the dispatches of k_lookup23_ascii appear in this order: handle 0 x L, handle 1 x L, ... (after
one warm-up launch per handle)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aindex_amd import engine
from aindex_amd.engine import Index

H = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ix0, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
N = 100_000_000
q = engine.synth_mix23_t(8, g, N)
out = torch.empty(N, dtype=torch.int32, device="cuda:0")
handles, pads, res = [ix0], [], []
for i in range(H - 1):
    pads.append(torch.empty((37 + 61 * i) << 20, dtype=torch.uint8, device="cuda:0"))
    handles.append(Index.build_23_codes_t(pf, keys, counts32, 0))
for ix in handles:
    ix.tf_ascii_t(q, out)
torch.cuda.synchronize()
for i, ix in enumerate(handles):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(L):
        ix.tf_ascii_t(q, out)
    b.record(); torch.cuda.synchronize()
    res.append({"handle": i, "kernel_ms": a.elapsed_time(b) / L, "device_bytes": ix.info["device_bytes"], "bytes_per_key": ix.info["device_bytes"] / ix.n})
print(json.dumps(res))
for r in res:
    print(r, file=sys.stderr)
