"""The 371 keys K1 reports at 200 M reads that the index does not hold, against the keys that lost a count (the two halves of the reads, counted
alone, are clean and hold the same key set): which bits differ between a wrong key and the key it should have been?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from aindex_amd import engine, counting, _lib

ix, g, keys, counts32, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
n_reads = 200_000_000
reads = engine.synth_reads_t(41, g, n_reads, 150, rc_half=True, n_rate_ppm=1000)
h = n_reads // 2 * 151
k1, c1 = counting.count_distinct_t(reads[:h], 23, _lib.CANON_TRUE_RC)
k2, c2 = counting.count_distinct_t(reads[h:], 23, _lib.CANON_TRUE_RC)
assert torch.equal(k1, k2)
want = c1.to(torch.int64) + c2.to(torch.int64)
dk, dc = counting.count_distinct_t(reads, 23, _lib.CANON_TRUE_RC)
found = ix.tf_codes_t(dk) != 0
assert torch.equal(dk[found], k1)
diff = dc[found].to(torch.int64) - want
losers = k1[diff != 0]
print("keys whose count differs:", int(losers.numel()), "differences:", sorted(set(diff[diff != 0].tolist())))
wrong = dk[~found]
print("wrong keys:", int(wrong.numel()), "their counts:", sorted(set(dc[~found].tolist())))
W = sorted(int(x) for x in wrong.tolist()); L = sorted(int(x) for x in losers.tolist())
# pair by low 24 bits, then by high bits, whichever matches uniquely
from collections import defaultdict
for bits in (24, 32, 35):
    m = (1 << bits) - 1
    idx = defaultdict(list)
    for x in L: idx[x & m].append(x)
    pairs = [(w, idx[w & m][0]) for w in W if len(idx.get(w & m, [])) == 1]
    print("paired by the low", bits, "bits:", len(pairs))
    if pairs:
        xs = sorted({w ^ l for w, l in pairs})
        print("   distinct xor patterns:", len(xs), [hex(x) for x in xs[:12]])
        for w, l in pairs[:6]:
            print("   wrong", hex(w), "should be", hex(l), "xor", hex(w ^ l))
        break
else:
    print("first wrong:", [hex(x) for x in W[:6]]); print("first losers:", [hex(x) for x in L[:6]])
