"""world_size-2 gloo worker (CPU): the N>1 host logic of counting — record-aligned read sharding +
all-reduce(sum) of per-rank tf[] — with the per-rank histogram computed by the oracle (the GPU kernel is
covered by -m gpu tests; this checks that sharded + reduced == unsharded)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O                      # noqa: E402
from aindex_amd import dist as adist, synth  # noqa: E402


def main():
    rank, world, _ = adist.init("gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    prefix = os.path.join(ROOT, "tests", "golden", "small23", "small23")
    orc = O.OracleIndex23.from_prefix(prefix)
    reads = open(prefix + ".reads", "rb").read()
    mine = adist.shard_lines(reads, rank, world)
    # shards are contiguous, record aligned, and partition the buffer
    lens = torch.tensor([len(mine)], dtype=torch.int64)
    adist.all_reduce_sum_(lens)
    assert int(lens.item()) == len(reads)
    assert mine == b"" or mine.endswith(b"\n")
    tf = torch.from_numpy(orc.count23_fixed(mine, False, 1).view(np.int32).copy())
    adist.all_reduce_sum_(tf)
    full = orc.count23_fixed(reads, False, 1)
    assert np.array_equal(tf.numpy().view(np.uint32), full)
    assert np.array_equal(full, orc.tf_array())          # == what kmer_counter -> compute_index stored
    # query sharding: contiguous ranges, concatenation of per-rank answers == unsharded answers
    q = synth.random_kmers_ascii(7, 1001, 23)
    lo, hi = adist.shard_range(q.shape[0], rank, world)
    part = torch.from_numpy(orc.tf_batch(q[lo:hi]).view(np.int32).copy())
    sizes = [adist.shard_range(q.shape[0], r, world) for r in range(world)]
    bufs = [torch.empty(h - l, dtype=torch.int32) for l, h in sizes]
    dist.all_gather(bufs, part) if len({b.numel() for b in bufs}) == 1 else None
    if len({b.numel() for b in bufs}) != 1:      # uneven split: gather through padded tensors
        m = max(b.numel() for b in bufs)
        pad = torch.zeros(m, dtype=torch.int32); pad[: part.numel()] = part
        outs = [torch.zeros(m, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(outs, pad)
        bufs = [o[: h - l] for o, (l, h) in zip(outs, sizes)]
    got = torch.cat(bufs).numpy().view(np.uint32)
    assert np.array_equal(got, orc.tf_batch(q))
    t = adist.all_reduce_max_float(float(rank + 1))
    assert t == 2.0
    adist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
