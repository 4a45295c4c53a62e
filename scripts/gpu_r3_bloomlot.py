"""Does the self-probe time of ONE handle change when only its absence filter (100 MB) is moved to another block? Twelve relocations with pads of
varying size in between; self-probe (20 M all-hit code lookups) and Q_mix (100 M) after each."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aindex_amd import engine
from aindex_amd._lib import check
L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aindex_amd", "lib", "libaindex_hip.so"))
L.aix_debug_relocate_bloom.argtypes = [C.c_void_p, C.c_uint64]
ix, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
N = 100_000_000
q = engine.synth_mix23_t(8, g, N)
out = torch.empty(N, dtype=torch.int32, device="cuda:0")
codes = keys[torch.randperm(keys.numel(), device="cuda:0")[:20_000_000]].contiguous()
pout = torch.empty(codes.numel(), dtype=torch.int32, device="cuda:0")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 4)


res = [{"move": 0, "selfprobe_ms": timed(lambda: ix.tf_codes_t(codes, pout)), "qmix_ms": timed(lambda: ix.tf_ascii_t(q, out))}]
for m in range(1, 13):
    check(L.aix_debug_relocate_bloom(ix._h, (m * 37) << 20))
    res.append({"move": m, "selfprobe_ms": timed(lambda: ix.tf_codes_t(codes, pout)), "qmix_ms": timed(lambda: ix.tf_ascii_t(q, out))})
for r in res:
    print(r, file=sys.stderr)
print(json.dumps(res))
