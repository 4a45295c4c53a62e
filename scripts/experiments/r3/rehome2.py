"""Q_mix levels: the level of a handle against the device addresses of its verification table and absence filter. One handle; the table alone,
the filter alone and both are moved into fresh blocks many times (old blocks kept, pads of varying size in between)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from aindex_amd._lib import lib, check

ix, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
codes = keys[torch.randperm(keys.numel(), device="cuda:0")[:20_000_000]].contiguous()
pout = torch.empty(codes.numel(), dtype=torch.int32, device="cuda:0")
pads = []


def probe(reps=7):
    ix.tf_codes_t(codes, pout); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ix.tf_codes_t(codes, pout)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 4)


def ptrs():
    a = (C.c_uint64 * 5)()
    check(lib().aix_debug_pointers(ix._h, a), "pointers")
    return [int(x) for x in a]


rows = []
def note(what):
    p = ptrs()
    t = probe()
    rows.append({"moved": what, "table": hex(p[3]), "filter": hex(p[4]), "selfprobe_ms": t})
    print(what, hex(p[3]), hex(p[4]), t, flush=True)

note("-")
import random
random.seed(1)
for i in range(14):
    pads.append(torch.empty(random.choice([3, 17, 64, 129, 300, 511]) << 20, dtype=torch.uint8, device="cuda:0"))
    check(lib().aix_debug_rehome(ix._h, 8), "rehome"); note("table")
for i in range(6):
    check(lib().aix_debug_rehome(ix._h, 16), "rehome"); note("filter")
print(json.dumps(rows))
