"""world_size-3 gloo worker (CPU): a rank whose share is EMPTY (more ranks than reads) still joins every collective of the N > 1 path,
and the pieces of the multi-GPU protocols that are pure index arithmetic are pinned here, independent of the backend: the narrow (u32)
all-reduce of the 13-mer table, the K1 exchange with an empty local set, the scatter-merge slice bounds with totals that are 0, below the
world size and not divisible by it."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O                      # noqa: E402
from aindex_amd import dist as adist        # noqa: E402


def main():
    rank, world, _ = adist.init("gloo")
    assert world == 3
    prefix = os.path.join(ROOT, "tests", "golden", "small23", "small23")
    orc = O.OracleIndex23.from_prefix(prefix)
    reads = b"\n".join(open(prefix + ".reads", "rb").read().split(b"\n")[:2]) + b"\n"       # two reads for three ranks
    shares = [adist.shard_lines(reads, r, world) for r in range(world)]
    assert b"".join(shares) == reads and any(len(x) == 0 for x in shares), [len(x) for x in shares]
    mine = shares[rank]
    # counting: the empty rank contributes zeros to the one all-reduce
    tf = torch.from_numpy(orc.count23_fixed(mine, False, 1).view(np.int32).copy())
    adist.all_reduce_sum_(tf)
    assert np.array_equal(tf.numpy().view(np.uint32), orc.count23_fixed(reads, False, 1))
    # the 13-mer table: narrow when every counter stays below 2^32, wide otherwise; an all-zero share joins both
    t = torch.zeros(1000, dtype=torch.int64)
    if len(mine):
        t[rank] = (1 << 31) + 5
        t[7] = 3
    want = sum((torch.tensor([(1 << 31) + 5 if i == r else (3 if i == 7 else 0) for i in range(1000)], dtype=torch.int64) if len(shares[r]) else torch.zeros(1000, dtype=torch.int64))
               for r in range(world))
    got, bits = adist.all_reduce_sum_u64_narrow_(t.clone())
    assert bits == (32 if int(sum(((1 << 31) + 5) for r in range(world) if len(shares[r]))) < (1 << 32) else 64)
    assert torch.equal(got, want)
    big = torch.zeros(10, dtype=torch.int64)
    big[1] = (1 << 33) + rank
    got, bits = adist.all_reduce_sum_u64_narrow_(big)
    assert bits == 64 and got[1].item() == 3 * (1 << 33) + 3
    edge = torch.full((4,), (1 << 32) - 1 if rank == 0 else 0, dtype=torch.int64)                 # sum of maxima = 2^32 - 1: still narrow
    got, bits = adist.all_reduce_sum_u64_narrow_(edge)
    assert bits == 32 and got.tolist() == [(1 << 32) - 1] * 4
    got, bits = adist.all_reduce_sum_u64_narrow_(torch.zeros(0, dtype=torch.int64))
    assert got.numel() == 0
    # K1 exchange: a rank with no keys still takes part in the all-to-alls and may own keys of the others
    lines = [l for l in mine.split(b"\n") if l]
    fa = b"".join(b">r\n" + l.replace(b"~", b"\n>m\n") + b"\n" for l in lines)
    lk, lc = O.count_distinct(fa, 23, 2, 1) if lines else (np.zeros(0, np.uint64), np.zeros(0, np.uint64))
    sk, sc = adist.exchange_merge_counts(torch.from_numpy(lk.view(np.int64).copy()), torch.from_numpy(lc.astype(np.int64)), 1, keys_sorted=True)
    n_all = torch.tensor([sk.numel(), int(sc.sum())], dtype=torch.int64)
    adist.all_reduce_sum_(n_all)
    all_lines = [l for l in reads.split(b"\n") if l]
    fk, fc = O.count_distinct(b"".join(b">r\n" + l.replace(b"~", b"\n>m\n") + b"\n" for l in all_lines), 23, 2, 1)
    assert n_all.tolist() == [fk.shape[0], int(fc.sum())]
    assert torch.all(adist._owner_of(sk, world) == rank)
    # scatter-merge slice bounds and the merge itself (gloo: all-reduce + cut), totals of every awkward kind
    for total in (0, 1, 2, 3, 4, 10, 11, 1000, 1001):
        per, lo, hi = adist.scatter_slice_bounds(total, rank, world)
        assert per == -(-total // world) and 0 <= lo <= hi <= total and hi - lo <= per
        bounds = [adist.scatter_slice_bounds(total, r, world) for r in range(world)]
        assert bounds[0][1] == 0 and bounds[-1][2] == total and all(bounds[i][2] == bounds[i + 1][1] for i in range(world - 1))
        full = torch.arange(1, total + 1, dtype=torch.int64) * 7
        for merge in ("scatter", "all"):
            size = max(per * world, 1) if merge == "scatter" else max(total, 1)
            part = torch.zeros(size, dtype=torch.int64)
            idx = torch.arange(total)[rank::world]                                                  # entry i is written by rank i % world
            part[idx] = full[idx]
            got, (glo, ghi) = adist.merge_positions_(part, total, merge, rank, world)
            if merge == "scatter":
                assert (glo, ghi) == (lo, hi) and torch.equal(got, full[lo:hi]), (total, merge)
            else:
                assert (glo, ghi) == (0, total) and torch.equal(got, full), (total, merge)
    adist._raise_together(None, "nothing failed")
    adist.barrier()
    if rank == 0:
        print("DIST_EMPTY_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
