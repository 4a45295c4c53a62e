"""Q_mix placement spread: does the random-read rate of a 1.6 GiB block depend on HOW / WHEN the block was allocated?
Raw hipMalloc blocks (no torch allocator in between), k_gather with 16-byte reads over each, one process:
  first   : the first allocation of the process
  late    : after 24 odd-sized blocks were allocated and every other one freed (fragmented free list)
  slab    : carved out of one 8 GiB block allocated at that point
  again   : a second plain hipMalloc of the same size while the others are alive
For every block: address, address mod 2 MiB / 1 GiB, gather rate (best of 3). One JSON line."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                   # loads the HIP runtime the library will use
from aindex_amd._lib import lib, check, vp

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]


def malloc(n):
    p = C.c_void_p()
    r = hip.hipMalloc(C.byref(p), n)
    assert r == 0 and p.value, r
    return p.value


torch.cuda.init()
sink = torch.zeros(8, dtype=torch.int64, device="cuda:0")
stream = vp(torch.cuda.current_stream().cuda_stream)
SIZE = 1600 << 20
NACC = 200_000_000


def rate(ptr, size=SIZE, elem=16):
    best = 0.0
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        check(lib().aix_bench_gather_dev(vp(ptr), size // elem, elem, 1, NACC, 99, vp(sink.data_ptr()), stream))
        b.record()
        torch.cuda.synchronize()
        best = max(best, NACC / (a.elapsed_time(b) * 1e-3))
    return best


out = []


def note(name, ptr):
    hip.hipMemset(C.c_void_p(ptr), 1, SIZE)
    torch.cuda.synchronize()
    out.append({"block": name, "addr": hex(ptr), "mod_2MiB": ptr % (2 << 20), "mod_1GiB_MiB": (ptr % (1 << 30)) >> 20, "G_per_s": round(rate(ptr) / 1e9, 2), "G_per_s_128B": round(rate(ptr, SIZE, 128) / 1e9, 2)})


first = malloc(SIZE)
note("first", first)
junk = [malloc((37 + 61 * i) << 20) for i in range(24)]
for i in range(0, 24, 2):
    hip.hipFree(C.c_void_p(junk[i]))
late = malloc(SIZE)
note("late", late)
slab = malloc(8 << 30)
note("slab+0", slab)
note("slab+3GiB+1MiB", slab + (3 << 30) + (1 << 20))
again = malloc(SIZE)
note("again", again)
note("first (re-measured)", first)
# misaligned start inside a block: does the start address matter at all?
big = malloc(SIZE + (64 << 20))
note("big+0", big)
note("big+33MiB", big + (33 << 20))
print(json.dumps(out))
for r in out:
    print(r, file=sys.stderr)
