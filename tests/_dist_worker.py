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
    # K1 across ranks: per-rank distinct sets (oracle on this rank's reads) -> ONE all-to-all by key owner -> summed counts;
    # the union of the ranks' shares must equal the unsharded run, min_count applied to the GLOBAL counts
    lines = [l for l in reads.split(b"\n") if l]
    lo, hi = adist.shard_range(len(lines), rank, world)
    fasta = lambda ls: b"".join(b">r\n" + l.replace(b"~", b"\n>m\n") + b"\n" for l in ls)
    for k, mode, minc in ((23, 1, 1), (23, 2, 2), (13, 2, 3)):
        lk, lc = O.count_distinct(fasta(lines[lo:hi]), k, mode, 1)
        sk, sc = adist.exchange_merge_counts(torch.from_numpy(lk.view(np.int64).copy()), torch.from_numpy(lc.astype(np.int64)), minc)
        assert torch.all(adist._owner_of(sk, world) == rank)                   # only keys this rank owns
        assert torch.all(sk[1:] > sk[:-1]) if sk.numel() > 1 else True
        sz = torch.tensor([sk.numel()], dtype=torch.int64)
        szs = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(szs, sz)
        m = max(int(x.item()) for x in szs)
        pk = torch.full((m,), -1, dtype=torch.int64); pk[: sk.numel()] = sk
        pc = torch.zeros(m, dtype=torch.int64); pc[: sc.numel()] = sc
        gk = [torch.empty(m, dtype=torch.int64) for _ in range(world)]
        gc = [torch.empty(m, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(gk, pk); dist.all_gather(gc, pc)
        allk = torch.cat([g[: int(n.item())] for g, n in zip(gk, szs)]).numpy().view(np.uint64)
        allc = torch.cat([g[: int(n.item())] for g, n in zip(gc, szs)]).numpy().astype(np.uint64)
        o = np.argsort(allk, kind="stable")
        fk, fc = O.count_distinct(fasta(lines), k, mode, minc)
        assert fk.shape[0] > 100 and np.array_equal(allk[o], fk) and np.array_equal(allc[o], fc.astype(np.uint64)), (k, mode, minc)
    # I1 across ranks: every rank scatters a contiguous share of the keys into zeroed full-size arrays (slots from the
    # oracle MPHF here), merge = max(checker) / sum(tf) / sum(occupied); == the golden .kmers.bin / .tf.bin
    m = O.OracleMphf(prefix + ".pf")
    full_checker = np.fromfile(prefix + ".kmers.bin", dtype=np.uint64)
    full_tf = np.fromfile(prefix + ".tf.bin", dtype=np.uint32)
    keys = synth.decode_kmers(full_checker, 23)
    perm = np.random.default_rng(3).permutation(keys.shape[0])      # .dat order is not slot order
    keys, cnts = keys[perm], full_tf[perm]
    lo, hi = adist.shard_range(keys.shape[0], rank, world)
    slots = np.array([m.lookup(bytes(x)) for x in keys[lo:hi]], dtype=np.int64)
    ck = np.zeros(orc.n, dtype=np.int64); tfp = np.zeros(orc.n, dtype=np.int32); oc = np.zeros(orc.n, dtype=np.int32)
    ck[slots] = synth.encode_kmers(keys[lo:hi]).view(np.int64); tfp[slots] = cnts[lo:hi].view(np.int32); oc[slots] = 1
    ckt, tft, oct_ = torch.from_numpy(ck), torch.from_numpy(tfp), torch.from_numpy(oc)
    assert adist.merge_scatter_shards(ckt, tft, oct_) is False
    assert np.array_equal(ckt.numpy().view(np.uint64), full_checker) and np.array_equal(tft.numpy().view(np.uint32), full_tf)
    assert int(oct_.sum()) == orc.n
    dup = torch.zeros(orc.n, dtype=torch.int32); dup[0] = 1         # both ranks claim slot 0 -> collision reported on every rank
    assert adist.merge_scatter_shards(torch.zeros(orc.n, dtype=torch.int64), torch.zeros(orc.n, dtype=torch.int32), dup) is True
    t = adist.all_reduce_max_float(float(rank + 1))
    assert t == 2.0
    # helpers of the device-resident shard protocols (positions_fill_sharded_t / scatter_sharded_t), on CPU tensors here: u32 bit
    # patterns of tallies above 2^31, the rank-ordered gather that turns them into "occurrences in the shards before mine", the
    # in-place reductions, and the collective error flag that makes every rank raise together
    tall = torch.tensor([0, 5, (1 << 31) + 7, (1 << 32) - 1, (1 << 40)], dtype=torch.int64) + rank
    bits = adist._u32_bits(tall)
    assert bits.dtype == torch.int32 and (bits.to(torch.int64) & 0xFFFFFFFF).tolist() == [min(int(v), (1 << 32) - 1) for v in tall.tolist()]
    parts = adist._all_gather(bits)
    assert len(parts) == world and all((p_.to(torch.int64) & 0xFFFFFFFF).tolist() == [min(int(v) - rank + r, (1 << 32) - 1) for v in tall.tolist()] for r, p_ in enumerate(parts))
    before = sum((p_.to(torch.int64) & 0xFFFFFFFF) for p_ in parts[:rank]) if rank else torch.zeros(5, dtype=torch.int64)
    assert before.tolist() == ([0, 0, 0, 0, 0] if rank == 0 else [0, 5, (1 << 31) + 7, (1 << 32) - 1, (1 << 32) - 1])
    x = torch.tensor([rank + 1, 10 * (rank + 1)], dtype=torch.int64)
    assert adist._all_reduce_(x.clone(), "sum").tolist() == [3, 30] and adist._all_reduce_(x.clone(), "max").tolist() == [2, 20]
    adist._raise_together(None, "nothing failed")
    try:
        adist._raise_together(RuntimeError("boom") if rank == 1 else None, "one rank failed")
        raise AssertionError("no rank raised")
    except RuntimeError as e:
        assert ("boom" in str(e)) == (rank == 1) and "one rank failed" in str(e)
    adist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
