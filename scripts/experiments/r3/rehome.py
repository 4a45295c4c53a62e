"""Q_mix levels: which of a handle's device arrays, moved into fresh blocks, changes its level? Six identical handles; the 0.5 ms self-probe
(20 M lookups of stored codes) tells fast from slow. For one slow and one fast handle: move each array alone, then all of them, several times."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from aindex_amd import engine, builder
from aindex_amd.engine import Index
from aindex_amd._lib import lib, check

ix0, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
handles = [ix0] + [Index.build_23_codes_t(pf, keys, counts32) for _ in range(5)]
codes = keys[torch.randperm(keys.numel(), device="cuda:0")[:20_000_000]].contiguous()
pout = torch.empty(codes.numel(), dtype=torch.int32, device="cuda:0")


def probe(ix, reps=7):
    ix.tf_codes_t(codes, pout); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ix.tf_codes_t(codes, pout)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 4)


base = [probe(h) for h in handles]
print("self-probe of six handles:", base, flush=True)
order = sorted(range(6), key=lambda i: base[i])
fast, slow = order[0], order[-1]
out = {"base": base, "fast": fast, "slow": slow, "moves": []}
for name, h in (("slow", handles[slow]), ("fast", handles[fast])):
    for mask in (1, 2, 4, 8, 16, 24, 31, 31, 31, 31):
        check(lib().aix_debug_rehome(h._h, mask), "rehome")
        t = probe(h)
        out["moves"].append({"handle": name, "mask": mask, "selfprobe_ms": t})
        print(name, "mask", mask, t, flush=True)
print(json.dumps(out))
