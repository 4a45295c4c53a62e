"""Placement of the verification table: is a handle slow because of WHERE its table block lies? One process. A 2 GiB block is reserved at the very
start (before anything else is allocated). Eight handles are built; then (a) the table of a SLOW handle is moved into the early block, (b) the table
of a FAST handle is moved into a block allocated late. Self-probe (20 M all-hit code lookups) before and after."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
early = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0")            # the first allocation of the process
early.zero_()
import bench
from aindex_amd import engine
from aindex_amd.engine import Index
from aindex_amd._lib import lib, check, vp
L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aindex_amd", "lib", "libaindex_hip.so"))
L.aix_debug_relocate_table.argtypes = [C.c_void_p, C.c_void_p]

ix0, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
perm = torch.randperm(keys.numel(), device="cuda:0")[:20_000_000]
codes = keys[perm].contiguous()
pout = torch.empty(codes.numel(), dtype=torch.int32, device="cuda:0")
handles, pads = [ix0], []
for i in range(7):
    pads.append(torch.empty((37 + 61 * i) << 20, dtype=torch.uint8, device="cuda:0"))
    handles.append(Index.build_23_codes_t(pf, keys, counts32, 0))


def probe(ix):
    ix.tf_codes_t(codes, pout); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ix.tf_codes_t(codes, pout)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / 5, 4)


before = [probe(ix) for ix in handles]
print("self-probe ms per handle:", before, file=sys.stderr)
slow = max(range(8), key=lambda i: before[i])
fast = min(range(8), key=lambda i: before[i])
assert handles[slow].info["buckets"] * 128 <= early.numel()
check(L.aix_debug_relocate_table(handles[slow]._h, C.c_void_p(early.data_ptr())))
check(L.aix_debug_relocate_table(handles[fast]._h, None))
after = [probe(ix) for ix in handles]
print("after: slow handle", slow, "-> early block:", before[slow], "->", after[slow], "| fast handle", fast, "-> late block:", before[fast], "->", after[fast], file=sys.stderr)
print("all after:", after, file=sys.stderr)
print(json.dumps({"before": before, "after": after, "slow": slow, "fast": fast}))
