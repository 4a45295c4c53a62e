"""Where do the 371 keys that K1 reports at 200 M reads and the index does not hold come from? Their bit patterns, and K1 on sub-ranges of the reads."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from aindex_amd import engine, counting, _lib

ix, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
n_reads = 200_000_000
reads = engine.synth_reads_t(41, g, n_reads, 150, rc_half=True, n_rate_ppm=1000)


def missing_of(buf, piece=None):
    if piece:
        os.environ["AIX_DISTINCT_PIECE"] = str(piece)
    try:
        dk, dc = counting.count_distinct_t(buf, 23, _lib.CANON_TRUE_RC)
    finally:
        os.environ.pop("AIX_DISTINCT_PIECE", None)
    tf = ix.tf_codes_t(dk)
    return dk[tf == 0], int(dk.numel())


m, n = missing_of(reads)
ms = [int(x) for x in m.tolist()]
print("all reads: distinct", n, "missing", len(ms))
print("missing keys (first 12):", [hex(x) for x in ms[:12]])
print("missing keys (last 6):", [hex(x) for x in ms[-6:]])
print("top 11 bits of the missing keys:", sorted({x >> 35 for x in ms}))
print("bits 24..34:", sorted({(x >> 24) & 0x7FF for x in ms})[:20])
# sub-ranges: which eighth of the reads produces them when counted alone?
for i in range(8):
    lo, hi = i * 25_000_000, (i + 1) * 25_000_000
    mm, nn = missing_of(reads[lo * 151: hi * 151])
    print("reads", lo, hi, "alone: distinct", nn, "missing", int(mm.numel()), flush=True)
# the same reads with other piece sizes
for piece in (1 << 29, 1 << 31):
    mm, nn = missing_of(reads, piece)
    print("piece", piece, "distinct", nn, "missing", int(mm.numel()), flush=True)
# halves
for lo, hi in ((0, 100_000_000), (100_000_000, 200_000_000), (0, 150_000_000)):
    mm, nn = missing_of(reads[lo * 151: hi * 151])
    print("reads", lo, hi, "distinct", nn, "missing", int(mm.numel()), flush=True)
