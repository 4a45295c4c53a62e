"""N>1 worker on GPU(s): every sharded entry point of aindex_amd.dist against the unsharded HIP result and the oracle.
Launched by test_gpu_parity.py with 2 gloo ranks sharing cuda:0 (the box has one GPU), and with 1 nccl rank
(AIX_FORCE_DIST=1) so that the RCCL calls themselves run. The oracle is only the checker here."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O                                    # noqa: E402
from aindex_amd import _lib, dist as adist, synth       # noqa: E402
from aindex_amd.engine import Index                        # noqa: E402


def gather_var(t):
    """all_gather of int64 tensors of different lengths (through host memory; test plumbing)."""
    world = dist.get_world_size()
    t = t.cpu()
    if world == 1:
        return t
    n = torch.tensor([t.numel()], dtype=torch.int64)
    ns = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(ns, n)
    m = max(int(x.item()) for x in ns)
    pad = torch.zeros(m, dtype=torch.int64); pad[: t.numel()] = t
    outs = [torch.empty(m, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: int(x.item())] for o, x in zip(outs, ns)])


def main():
    rank, world, _ = adist.init()
    torch.cuda.set_device(0)
    prefix = os.path.join(ROOT, "tests", "golden", "small23", "small23")
    reads = open(prefix + ".reads", "rb").read()
    orc = O.OracleIndex23.from_prefix(prefix)
    with Index.open_23(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin", device=0) as ix:
        tf = adist.count23_sharded(ix, reads, _lib.CANON_REF_X86, device=0)
        assert np.array_equal(tf.cpu().numpy().view(np.uint32), orc.tf_array())
        seqs = [l.replace(b"~", b"N") for l in reads.split(b"\n")[:41]]
        lo, hi, prof = adist.coverage_sharded(ix, seqs, 2)
        for s, p in zip(seqs[lo:hi], prof):
            assert np.array_equal(p, orc.coverage(s, 2))
        # A2 across ranks: bucket tallies -> all-gather -> shard fills with carried slot numbering -> sum == the reference's
        # 1-thread files; then a buffer whose first shard has no clean window, so that the start adjustment carries over
        z = np.load(os.path.join(ROOT, "tests", "golden", "small23", "aindex.npz"))
        ind, pos = adist.positions_fill_sharded(ix, reads)
        assert np.array_equal(ind, z["indices"]) and np.array_equal(pos, z["index"])
        lines = [l for l in reads.split(b"\n") if l]
        tricky = b"ACGTAC\n" * 2500 + lines[0][:30] + b"?" + lines[0][31:] + b"\n" + b"\n".join(lines[:40] + lines[:25]) + b"\n"
        want_ind, want_pos = orc.positions(tricky)
        ind, pos = adist.positions_fill_sharded(ix, tricky)
        assert np.array_equal(ind, want_ind) and np.array_equal(pos, want_pos) and (pos != 0).sum() > 1000
        i1, p1 = ix.positions_fill(tricky)
        assert np.array_equal(p1, want_pos)
        # the same two files with every rank's shard and all partial results resident in HBM (device-tensor protocol):
        # all-reduce merge (everybody holds the whole array) and reduce-scatter merge (rank r holds the r-th slice)
        for buf, wi, wp in ((reads, z["indices"], z["index"]), (tricky, want_ind, want_pos)):
            lo_b, hi_b = adist.shard_line_bounds(buf, rank, world)
            mine = buf[lo_b:hi_b]
            st = torch.frombuffer(bytearray(mine) if mine else bytearray(1), dtype=torch.uint8)[: len(mine)].cuda()
            ind_t, pos_t, (p_lo, p_hi) = adist.positions_fill_sharded_t(ix, st, lo_b)
            assert (p_lo, p_hi) == (0, wp.shape[0]) and ind_t.is_cuda and pos_t.is_cuda
            assert np.array_equal(ind_t.cpu().numpy().view(np.uint64), wi) and np.array_equal(pos_t.cpu().numpy().view(np.uint64), wp)
            ind_t, pos_t, (p_lo, p_hi) = adist.positions_fill_sharded_t(ix, st, lo_b, merge="scatter")
            assert p_hi - p_lo == pos_t.numel() and np.array_equal(pos_t.cpu().numpy().view(np.uint64), wp[p_lo:p_hi])
            span = torch.tensor([p_hi - p_lo], dtype=torch.int64, device="cuda:0" if dist.get_backend() == "nccl" else "cpu")
            assert int(adist.all_reduce_sum_(span).item()) == wp.shape[0]      # the slices tile the array
        cnt = torch.tensor([hi - lo], dtype=torch.int64, device="cuda:0" if dist.get_backend() == "nccl" else "cpu")
        adist.all_reduce_sum_(cnt)
        assert int(cnt.item()) == len(seqs)
    plain = reads.replace(b"~", b"\n")
    assert plain.endswith(b"\n")
    fasta = b"".join(b">r\n" + l + b"\n" for l in plain.split(b"\n") if l)
    for k, mode, minc in ((23, _lib.CANON_REF_X86, 1), (23, _lib.CANON_TRUE_RC, 2), (13, _lib.CANON_TRUE_RC, 3)):
        sk, sc = adist.count_distinct_sharded(plain, k, mode, minc, device=0)
        assert sk.is_cuda and bool(torch.all(adist._owner_of(sk, world) == rank))
        allk, allc = gather_var(sk).numpy().view(np.uint64), gather_var(sc).numpy().astype(np.uint64)
        o = np.argsort(allk, kind="stable")
        fk, fc = O.count_distinct(fasta, k, mode, minc)
        assert fk.shape[0] > 100 and np.array_equal(allk[o], fk) and np.array_equal(allc[o], fc.astype(np.uint64)), (k, mode, minc)
        mine = adist.shard_lines(plain, rank, world)                          # the same with this rank's share already resident
        mt = torch.frombuffer(bytearray(mine) if mine else bytearray(1), dtype=torch.uint8)[: len(mine)].cuda()
        tk, tc = adist.count_distinct_sharded_t(mt, k, mode, minc)
        assert torch.equal(tk, sk) and torch.equal(tc, sc)
    # I1 across ranks: shard scatter on the GPU + max/sum merges == the reference's compute_index files
    full_checker = np.fromfile(prefix + ".kmers.bin", dtype=np.uint64)
    full_tf = np.fromfile(prefix + ".tf.bin", dtype=np.uint32)
    perm = np.random.default_rng(3).permutation(full_checker.shape[0])
    keys = synth.decode_kmers(full_checker, 23)[perm]
    ck, tfv = adist.scatter_sharded(open(prefix + ".pf", "rb").read(), keys, full_tf[perm], device=0)
    assert np.array_equal(ck, full_checker) and np.array_equal(tfv, full_tf)
    bad = keys.copy(); bad[-1] = bad[0]                                   # a duplicate key: collision, reported on every rank
    try:
        adist.scatter_sharded(open(prefix + ".pf", "rb").read(), bad, full_tf[perm], device=0)
        raise AssertionError("duplicate key not detected")
    except RuntimeError as e:
        assert "conflict" in str(e)
    # the same with the key codes resident: every rank holds ITS share of the codes in HBM, the merged arrays stay there
    lo_k, hi_k = adist.shard_range(perm.shape[0], rank, world)
    codes_t = torch.from_numpy(full_checker[perm][lo_k:hi_k].view(np.int64).copy()).cuda()
    cnt_t = torch.from_numpy(full_tf[perm][lo_k:hi_k].view(np.int32).copy()).cuda()
    ck_t, tf_t = adist.scatter_sharded_t(open(prefix + ".pf", "rb").read(), codes_t, cnt_t, full_checker.shape[0])
    assert ck_t.is_cuda and np.array_equal(ck_t.cpu().numpy().view(np.uint64), full_checker) and np.array_equal(tf_t.cpu().numpy().view(np.uint32), full_tf)
    dup = codes_t.clone()
    if rank == world - 1:
        dup[-1] = int(full_checker[perm][0])                                # a key of rank 0's share again: clash across (or inside) shards
    try:
        adist.scatter_sharded_t(open(prefix + ".pf", "rb").read(), dup, cnt_t, full_checker.shape[0])
        raise AssertionError("duplicate key not detected (device twin)")
    except RuntimeError as e:
        assert "conflict" in str(e)
    from pf13 import pf13_path
    with Index.open_13(pf13_path(), None, device=0) as ix13:
        got = adist.count13_sharded(ix13, plain, device=0).cpu().numpy().view(np.uint64)
        m = O.OracleMphf(pf13_path())
        assert np.array_equal(got, O.count13(m, plain, _lib.FMT_PLAIN))
    adist.barrier()
    if rank == 0:
        print("DIST_GPU_OK", dist.get_backend(), world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
