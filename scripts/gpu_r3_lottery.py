"""Is a handle's Q_mix time predicted by a short self-probe on ITS OWN table (all-hit lookups of stored keys)? Eight handles with identical contents
in one process: Q_mix kernel time (100 M queries) and the time of 20 M lookups of stored codes per handle."""
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
perm = torch.randperm(keys.numel(), device="cuda:0")[:20_000_000]
probe_codes = keys[perm].contiguous()
pout = torch.empty(probe_codes.numel(), dtype=torch.int32, device="cuda:0")
handles, pads, res = [ix0], [], []
for i in range(7):
    pads.append(torch.empty((37 + 61 * i) << 20, dtype=torch.uint8, device="cuda:0"))
    handles.append(Index.build_23_codes_t(pf, keys, counts32, 0))


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


if os.environ.get("LOTTERY_NO_FILTER"):
    for ix in handles:
        ix.set_absence_filter(False)
for rep in range(2):
    for i, ix in enumerate(handles):
        res.append({"rep": rep, "handle": i, "qmix_ms": round(timed(lambda: ix.tf_ascii_t(q, out), 5), 4), "selfprobe_ms": round(timed(lambda: ix.tf_codes_t(probe_codes, pout), 5), 4)})
print(json.dumps(res))
for r in res:
    print(r, file=sys.stderr)
