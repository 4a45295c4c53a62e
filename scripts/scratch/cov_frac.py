import torch, numpy as np
from aindex_amd import engine, counting, builder, _lib
from aindex_amd.engine import Index
g = engine.synth_genome_t(23, 2_000_000)
keys, counts = counting.count_distinct_t(g, 23, _lib.CANON_TRUE_RC)
pf = builder.build_pf_codes_t(keys, 23)
ix = Index.build_23_codes_t(pf, keys, counts.to(torch.int32))
for L in (150, 1000, 10000):
    for rc in (False, True):
        for ppm in (0, 1000):
            n = 200
            seqs = engine.synth_reads_t(51, g, n, L, rc_half=rc, n_rate_ppm=ppm)
            rows = seqs.view(n, L + 1)
            nfrac = float((rows[:, :L] == ord('N')).float().mean())
            win = rows[:, :L].unfold(1, 23, 1).contiguous().view(-1)
            tf = ix.tf_ascii_t(win)
            print(L, rc, ppm, 'N frac %.5f' % nfrac, 'nonzero %.4f' % float((tf != 0).float().mean()), flush=True)
