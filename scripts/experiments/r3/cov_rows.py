import torch, numpy as np
from aindex_amd import engine, counting, builder, _lib
from aindex_amd.engine import Index
g = engine.synth_genome_t(23, 2_000_000)
keys, counts = counting.count_distinct_t(g, 23, _lib.CANON_TRUE_RC)
pf = builder.build_pf_codes_t(keys, 23)
ix = Index.build_23_codes_t(pf, keys, counts.to(torch.int32))
n_seq, L, k = 1_000_000, 10_000, 23
seqs = engine.synth_reads_t(51, g, n_seq, L, rc_half=True, n_rate_ppm=1000)
offs = torch.arange(0, (n_seq + 1) * (L + 1), L + 1, dtype=torch.int64, device="cuda")
per = (L + 1) - k + 1
ooffs = torch.arange(0, (n_seq + 1) * per, per, dtype=torch.int64, device="cuda")
prof = ix.coverage_t(seqs, offs, ooffs, n_seq * per, 0).view(n_seq, per)
torch.cuda.synchronize()
rownz = torch.empty(n_seq, dtype=torch.int64, device="cuda")
for lo in range(0, n_seq, 50_000):
    rownz[lo:lo + 50_000] = (prof[lo:lo + 50_000, : L - k + 1] != 0).sum(dim=1)
r = rownz.cpu().numpy()
print("total nz", r.sum(), "frac", r.sum() / (n_seq * (L - k + 1)))
print("rows with < 50% nonzero:", int((r < 0.5 * (L - k + 1)).sum()))
bad = np.nonzero(r < 0.9 * (L - k + 1))[0]
print("rows < 90%:", bad.size, "first", bad[:10], "last", bad[-10:] if bad.size else None)
if bad.size:
    d = np.diff(bad); brk = np.nonzero(d > 1)[0]
    print("contiguous ranges:", brk.size + 1, "first range", bad[0], bad[brk[0]] if brk.size else bad[-1])
    b = int(bad[0]); row = prof[b].cpu().numpy()
    nzcols = np.nonzero(row)[0]
    print("row", b, "nonzero cols", nzcols.size, nzcols[:5], nzcols[-5:] if nzcols.size else None)
print("row nz percentiles", np.percentile(r, [0, 1, 10, 50, 90, 100]))
# slab sum the way the test does it
nz = 0
for lo in range(0, n_seq, 100_000):
    nz += int((prof[lo:lo + 100_000, : L - k + 1] != 0).sum().item())
print("slab-sum nz", nz)
nz2 = 0
for lo in range(0, n_seq, 100_000):
    nz2 += int(torch.count_nonzero(prof[lo:lo + 100_000, : L - k + 1]).item())
print("count_nonzero nz", nz2)
