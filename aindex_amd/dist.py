"""Multi-GPU sharding: one process per GPU, reads / queries split into contiguous ranges,
no data-path collective for lookups, one all-reduce(sum) of tf[] for counting (RCCL over xGMI when
the backend is "nccl"; "gloo" on CPU for the tests)."""
from __future__ import annotations

import os
from typing import Tuple


def rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    import torch
    import torch.distributed as dist
    rank, world, local = rank_world()
    if (world > 1 or os.environ.get("AIX_FORCE_DIST")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            backend = os.environ.get("AIX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) share of `total` items; the first `total % world` ranks get one extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_lines(buf: bytes, rank: int, world: int) -> bytes:
    """Record-aligned byte range of a PLAIN buffer (cuts only after '\\n'), contiguous per rank."""
    lo, hi = shard_line_bounds(buf, rank, world)
    return buf[lo:hi]


def shard_line_bounds(buf: bytes, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) byte offsets of shard_lines(buf, rank, world)."""
    n = len(buf)
    if world == 1:
        return 0, n

    def cut(pos):
        if pos <= 0:
            return 0
        if pos >= n:
            return n
        j = buf.find(b"\n", pos - 1)
        return n if j < 0 else j + 1

    lo, hi = shard_range(n, rank, world)
    return cut(lo), cut(hi)


def all_reduce_sum_(t):
    """In-place sum over ranks of an integer tensor (tf histograms). int32/int64 carry u32/u64 bit patterns:
    two's-complement addition is the same as unsigned addition modulo 2^32 / 2^64."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("AIX_FORCE_DIST")):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_sum_u64_narrow_(t):
    """In-place sum over ranks of a table of u64 counters carried as int64 (the 4^13 table of count_kmers13: 512 MiB), sent as u32 when
    that is exact. Every global counter is at most the sum over ranks of the ranks' largest local counter; ONE 8-byte all-reduce of
    those maxima decides: below 2^32 the table crosses the links as int32 (u32 bit patterns, 256 MiB, addition modulo 2^32 is then the
    true sum) and is widened back, otherwise it goes as int64. Returns (t, bits) with bits = 32 or 64 (0: no process group)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("AIX_FORCE_DIST"))):
        return t, 0
    via_host = t.is_cuda and dist.get_backend() != "nccl"
    m = (t.max() if t.numel() else torch.zeros((), dtype=torch.int64, device=t.device)).reshape(1).to(torch.float64)   # counts < 2^53 here: exact in f64, and the sum cannot wrap
    negative = bool((t.min() < 0).item()) if t.numel() else False                                                          # a counter above 2^63 - 1: leave it to the wide path
    flag = torch.tensor([m.item(), 1.0 if negative else 0.0], dtype=torch.float64, device="cpu" if via_host else t.device)
    dist.all_reduce(flag, op=dist.ReduceOp.SUM)
    if flag[0].item() < float(1 << 32) and flag[1].item() == 0.0:
        t32 = torch.where(t >= (1 << 31), t - (1 << 32), t).to(torch.int32)
        if via_host:
            h = t32.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t32 = h.to(t.device)
        else:
            dist.all_reduce(t32, op=dist.ReduceOp.SUM)
        t.copy_(t32.to(torch.int64) & 0xFFFFFFFF)
        return t, 32
    if via_host:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t, 64


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("AIX_FORCE_DIST")):
        dist.barrier()


def all_reduce_max_float(x: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("AIX_FORCE_DIST")):
        t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return x


# ---- sharded counting: every rank counts a contiguous, record-aligned share, one all-reduce --------
def count13_sharded(index, plain: bytes, device=None):
    """13-mer counts of a PLAIN reads buffer split over the ranks of the default process group.
    Returns a torch.int64 tensor [4^13] (mphf order, u64 bit patterns) holding the global counts on every rank."""
    import torch
    rank, world, local = rank_world()
    dev = index.device if device is None else device
    mine = shard_lines(plain, rank, world)
    t = torch.frombuffer(bytearray(mine) if mine else bytearray(1), dtype=torch.uint8)[: len(mine)].to(f"cuda:{dev}")
    out = index.count13_t(t)
    return all_reduce_sum_u64_narrow_(out)[0]


def count23_sharded(index, plain: bytes, canon_mode: int = 2, device=None):
    """tf[] histogram of a PLAIN reads buffer against the index's fixed key set, split over the ranks.
    Returns a torch.int32 tensor [n] (u32 bit patterns) holding the global histogram on every rank."""
    import torch
    rank, world, local = rank_world()
    dev = index.device if device is None else device
    mine = shard_lines(plain, rank, world)
    t = torch.frombuffer(bytearray(mine) if mine else bytearray(1), dtype=torch.uint8)[: len(mine)].to(f"cuda:{dev}")
    out = torch.zeros(index.n, dtype=torch.int32, device=f"cuda:{dev}")
    index.count23_fixed_t(t, canon_mode, out)
    return all_reduce_sum_(out)


def count13_sharded_t(index, plain_t, out_t=None):
    """Device-tensor twin of count13_sharded: `plain_t` is THIS rank's record-aligned share of the reads, already in HBM
    (uint8 tensor on the index's device) — nothing is uploaded from host memory. Counts it and all-reduces the 4^13 table
    in place (RCCL under "nccl"; as u32 — 256 MiB instead of 512 — whenever no counter can reach 2^32, all_reduce_sum_u64_narrow_).
    Returns the int64 tensor [4^13] (mphf order) with the global counts on every rank."""
    out_t = index.count13_t(plain_t, out_t)
    return all_reduce_sum_u64_narrow_(out_t)[0]


def count23_sharded_t(index, plain_t, canon_mode: int = 2, out_t=None):
    """Device-tensor twin of count23_sharded: this rank's share of the reads is already in HBM; histogram against the
    fixed key set into `out_t` (int32 [n], zeroed here) and ONE all-reduce(sum) — the only collective of the counting path
    (BASELINE configs[3]). Returns the global histogram on every rank."""
    import torch
    if out_t is None:
        out_t = torch.zeros(index.n, dtype=torch.int32, device=plain_t.device)
    else:
        out_t.zero_()
    index.count23_fixed_t(plain_t, canon_mode, out_t)
    return all_reduce_sum_(out_t)


def _raise_together(local_error, what: str, device="cpu"):
    """Collective error handling: a rank that failed locally must not leave the others blocked inside the next collective.
    Every rank contributes a 0/1 flag to ONE all-reduce and all of them raise (or none does)."""
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1 if local_error is not None else 0], dtype=torch.int32, device=device)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("AIX_FORCE_DIST")):
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
    if int(flag.item()):
        if local_error is not None:
            raise RuntimeError(f"{what}: {local_error}") from (local_error if isinstance(local_error, BaseException) else None)
        raise RuntimeError(f"{what}: failed on another rank")


def lookup_sharded(index, kmers_u8, device=None):
    """Batch tf lookup with the queries split into contiguous ranges over the ranks (index replicated); every rank
    returns only ITS range [lo, hi) as (lo, hi, uint32 numpy array) — no collective on the data path."""
    import numpy as np
    rank, world, _ = rank_world()
    a = np.ascontiguousarray(kmers_u8, dtype=np.uint8).reshape(-1, index.k)
    lo, hi = shard_range(a.shape[0], rank, world)
    return lo, hi, index.tf_ascii(a[lo:hi])


# ---- distinct k-mer discovery (K1) across ranks: the one step of the path with a real exchange ------
def _owner_of(keys, world: int):
    """Owning rank of every 2-bit key: splitmix64 finaliser of the code, modulo the world size. int64 tensors carry
    u64 bit patterns (multiplication wraps; logical shifts are spelled out because `>>` on int64 is arithmetic)."""
    import torch

    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)

    def i64(c):                                       # a u64 constant as the int64 with the same bit pattern
        return c - (1 << 64) if c >= (1 << 63) else c

    z = keys + i64(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * i64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * i64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    return lsr(z, 1) % world                          # non-negative before the modulo


def exchange_merge_counts(keys, counts, min_count: int = 1, keys_sorted: bool = False):
    """One all-to-all of (key, count) pairs: every rank sends each of its locally distinct keys to the key's owner and
    sums what it receives. keys: int64 (u64 bit patterns, any order, distinct per rank), counts: int64.
    Returns this rank's share of the GLOBAL distinct set: (keys ascending, summed counts >= min_count). The shares of
    all ranks are disjoint and their union is the unsharded result. Works on CPU tensors (gloo) and device tensors
    (nccl = RCCL; with gloo the exchange itself is staged through host memory).
    keys_sorted: this rank's keys are ascending (what count_distinct_t returns). The stable split by owner then leaves every peer's
    segment ascending, so the receiver holds one sorted run per peer and merges them (aix_merge_runs_dev: a tree of two-way merges with
    summation, the reference's single merge of per-thread maps, count_kmers.cpp:334-341) instead of sorting the concatenation."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    run_offsets = [0, int(keys.numel())]
    if world > 1 or (world == 1 and os.environ.get("AIX_FORCE_DIST") and dist.is_initialized()):
        owner = _owner_of(keys, world)
        order = torch.argsort(owner, stable=True)
        keys, counts = keys[order].contiguous(), counts[order].contiguous()
        send_sizes = torch.bincount(owner, minlength=world).to(torch.int64)
        dev = keys.device
        via_host = dist.get_backend() != "nccl" and dev.type != "cpu"
        xdev = torch.device("cpu") if via_host else dev
        send_sizes_x = send_sizes.to(xdev)
        recv_sizes_x = torch.empty(world, dtype=torch.int64, device=xdev)
        dist.all_to_all_single(recv_sizes_x, send_sizes_x)
        ins, outs = send_sizes_x.tolist(), recv_sizes_x.tolist()
        rk = torch.empty(sum(outs), dtype=torch.int64, device=xdev)
        rc = torch.empty(sum(outs), dtype=torch.int64, device=xdev)
        dist.all_to_all_single(rk, keys.to(xdev), output_split_sizes=outs, input_split_sizes=ins)
        dist.all_to_all_single(rc, counts.to(xdev), output_split_sizes=outs, input_split_sizes=ins)
        keys, counts = rk.to(dev), rc.to(dev)
        run_offsets = [0]
        for m in outs:
            run_offsets.append(run_offsets[-1] + int(m))
    # local merge: equal keys from different ranks are summed (keys compare as u64: every valid code is < 2^62)
    if keys.is_cuda:                                  # product path, inside the library: merge of the peers' sorted runs, or sort + reduce-by-key
        from .counting import merge_counts_t, merge_runs_t
        if keys_sorted:
            return merge_runs_t(keys, counts, run_offsets, min_count)
        return merge_counts_t(keys, counts, min_count)
    ukeys, inv = torch.unique(keys, sorted=True, return_inverse=True)      # CPU tensors: the gloo tests' plumbing
    sums = torch.zeros(ukeys.numel(), dtype=torch.int64, device=keys.device)
    sums.index_add_(0, inv, counts)
    if min_count > 1:
        keep = sums >= min_count
        ukeys, sums = ukeys[keep], sums[keep]
    return ukeys, sums


def count_distinct_sharded(plain: bytes, k: int, canon_mode: int = 2, min_count: int = 1, device: int = 0):
    """kmer_counter over ranks: every rank counts the distinct canonical k-mers of its record-aligned share of a PLAIN
    buffer on its GPU (window-code kernel + sort / run-length), then ONE exchange routes each key to its owner.
    Returns this rank's (keys, counts) share as device int64 tensors; min_count applies to the global counts."""
    import torch
    from . import counting
    rank, world, _ = rank_world()
    mine = shard_lines(plain, rank, world)
    t = torch.frombuffer(bytearray(mine) if mine else bytearray(1), dtype=torch.uint8)[: len(mine)].to(f"cuda:{device}")
    keys, counts = counting.count_distinct_t(t, k, canon_mode, 1)
    return exchange_merge_counts(keys, counts, min_count, keys_sorted=True)


def count_distinct_sharded_t(plain_t, k: int, canon_mode: int = 2, min_count: int = 1):
    """Device-tensor twin of count_distinct_sharded: `plain_t` (uint8, on this rank's GPU) is THIS rank's record-aligned share of
    the PLAIN reads, already in HBM — nothing is uploaded; the local distinct set and the exchanged (key, count) pairs stay there
    (one all-to-all under "nccl"). Returns this rank's share of the global distinct set as device int64 tensors."""
    from . import counting
    keys, counts = counting.count_distinct_t(plain_t, k, canon_mode, 1)
    return exchange_merge_counts(keys, counts, min_count, keys_sorted=True)


def coverage_sharded(index, seqs, cutoff: int = 0):
    """Per-position tf profiles with the SEQUENCES split into contiguous ranges over the ranks (index replicated, no
    collective on the data path). Every rank returns (lo, hi, list of uint32 arrays) for its own range."""
    rank, world, _ = rank_world()
    lo, hi = shard_range(len(seqs), rank, world)
    return lo, hi, index.coverage(seqs[lo:hi], cutoff)


# ---- I1 across ranks: every rank scatters a contiguous share of the key set, three all-reduces merge the shards -----
def merge_scatter_shards(checker, tf, occupied):
    """checker: int64 [n] (2-bit codes, zero where this rank wrote nothing), tf: int32 [n] (u32 bit patterns),
    occupied: int32 [n] in {0, 1}. In place: checker = max over ranks (codes are < 2^46, i.e. non-negative), tf = sum,
    occupied = sum. Returns True when two ranks wrote the same slot (the reference's collision, hash.cpp:703-709)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("AIX_FORCE_DIST")):
        dist.all_reduce(checker, op=dist.ReduceOp.MAX)
        dist.all_reduce(tf, op=dist.ReduceOp.SUM)
        dist.all_reduce(occupied, op=dist.ReduceOp.SUM)
    return bool((occupied > 1).any().item())


def scatter_sharded(pf_bytes: bytes, keys_u8, counts=None, device: int = 0):
    """compute_index over ranks: `keys_u8` (n x 23 ASCII, the whole key set, same on every rank) is split into contiguous
    ranges; every rank scatters its range through the MPHF on its GPU (aix_index_scatter_shard) and the shards are merged
    with max / sum all-reduces. Returns (checker uint64[n], tf uint32[n]) on every rank; raises on any collision."""
    import ctypes as C
    import numpy as np
    import torch
    from ._lib import lib, vp, AIX_ERR_CONFLICT, check
    rank, world, _ = rank_world()
    keys = np.ascontiguousarray(keys_u8, dtype=np.uint8).reshape(-1, 23)
    n = keys.shape[0]
    lo, hi = shard_range(n, rank, world)
    mine = np.ascontiguousarray(keys[lo:hi])
    cnt = None if counts is None else np.ascontiguousarray(np.asarray(counts, dtype=np.uint32)[lo:hi])
    checker = np.zeros(n, dtype=np.uint64)
    tf = np.zeros(n, dtype=np.uint32)
    occ = np.zeros((n + 31) // 32, dtype=np.uint32)
    pf = np.frombuffer(pf_bytes, dtype=np.uint8)
    st = lib().aix_index_scatter_shard(pf.ctypes.data_as(vp), pf.shape[0], mine.ctypes.data_as(vp), cnt.ctypes.data_as(vp) if cnt is not None else None,
                                       hi - lo, n, device, checker.ctypes.data_as(vp), tf.ctypes.data_as(vp), occ.ctypes.data_as(vp))
    import torch.distributed as dist
    on_gpu = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
    dev = f"cuda:{device}" if on_gpu else "cpu"
    err = None
    if st not in (0, AIX_ERR_CONFLICT):             # OOM, HIP error, malformed .pf ...: every rank raises, none is left in a collective
        try:
            check(st, "aix_index_scatter_shard")
        except Exception as e:
            err = e
    _raise_together(err, "scatter_sharded", dev)
    occ_bits = np.unpackbits(occ.view(np.uint8), bitorder="little")[:n].astype(np.int32)
    ct = torch.from_numpy(checker.view(np.int64)).to(dev)
    tt = torch.from_numpy(tf.view(np.int32)).to(dev)
    ot = torch.from_numpy(occ_bits).to(dev)
    bad = torch.tensor([1 if st == AIX_ERR_CONFLICT else 0], dtype=torch.int32, device=dev)
    clash = merge_scatter_shards(ct, tt, ot)
    all_reduce_sum_(bad)
    if clash or int(bad.item()):
        raise RuntimeError("hash conflict while scattering (a key outside the MPHF's key set, or a duplicate key)")
    return ct.cpu().numpy().view(np.uint64), tt.cpu().numpy().view(np.uint32)


# ---- A2 across ranks: buckets are global, so slots are resolved with per-rank bucket tallies ---------------
def positions_fill_sharded(index, reads: bytes):
    """compute_aindex over ranks: the reads file (same bytes on every rank) is cut after '\\n' into contiguous shards.
    Pass 1: every rank tallies its windows per bucket; an all-gather gives each rank the occurrences in the shards before
    it (the value the reference's ppositions counters would have when its single worker reaches this shard). Pass 2: every
    rank fills its shard with slot numbering starting there; the disjoint partial arrays are summed.
    Returns (indices uint64[n+1], positions uint64[sum tf]) on every rank == Index.positions_fill(reads)."""
    import ctypes as C
    import numpy as np
    import torch
    import torch.distributed as dist
    from ._lib import lib, check
    rank, world, _ = rank_world()
    active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or bool(os.environ.get("AIX_FORCE_DIST")))
    on_gpu = active and dist.get_backend() == "nccl"
    dev = f"cuda:{index.device}" if on_gpu else "cpu"
    lo, hi = shard_line_bounds(reads, rank, world)
    mine = reads[lo:hi]
    # the reference's start adjustment (hash.cpp:973-986) applies to the beginning of the FILE; it carries into the next
    # shard when a shard has no clean window at all
    st = C.c_uint64()
    a = np.frombuffer(mine, dtype=np.uint8)
    check(lib().aix_positions_start(a.ctypes.data_as(C.c_void_p) if a.shape[0] else None, a.shape[0], C.byref(st)), "aix_positions_start")
    exhausted = 1 if (len(mine) < 23 or st.value >= len(mine) - 22) else 0
    ex = torch.tensor([exhausted], dtype=torch.int32, device=dev)
    exs = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(world)]
    if active and world > 1:
        dist.all_gather(exs, ex)
    else:
        exs = [ex]
    first = all(int(e.item()) == 1 for e in exs[:rank])
    err, counts_np = None, None
    try:
        counts_np = index.positions_bucket_counts(mine, first)
    except Exception as e:                                # a failing rank must not leave the others inside the all-gather
        err = e
    _raise_together(err, "positions_fill_sharded (tally)", dev)
    counts = torch.from_numpy(counts_np.view(np.int64)).to(dev)
    if active and world > 1:
        allc = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(allc, counts)
    else:
        allc = [counts]
    before = torch.zeros_like(counts)
    for r in range(rank):
        before += allc[r]
    filled = torch.clamp(before, max=(1 << 32) - 1).cpu().numpy().astype(np.uint32)
    indices, part = None, None
    try:
        indices = index.positions_indices()
        part = index.positions_fill_shard(mine, int(indices[-1]), first, lo, filled)
    except Exception as e:
        err = e
    _raise_together(err, "positions_fill_sharded (fill)", dev)
    pt = torch.from_numpy(part.view(np.int64)).to(dev)
    all_reduce_sum_(pt)                                   # every entry is written by exactly one rank
    return indices, pt.cpu().numpy().view(np.uint64)


# ---- device-tensor twins of the I1 / A2 shard protocols: a rank's share and its partial results stay in HBM ---------------
def _active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or bool(os.environ.get("AIX_FORCE_DIST")))


def _all_reduce_(t, op="sum"):
    """In-place all-reduce of a device tensor. Under "nccl" (RCCL) the tensor never leaves HBM; under "gloo" (the one-GPU
    rehearsal of the tests) the collective is staged through host memory."""
    import torch.distributed as dist
    if not _active():
        return t
    rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX}[op]
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=rop)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=rop)
    return t


def _all_gather(t):
    """List of every rank's tensor (same shape on all ranks), on t's device; staged through host memory under "gloo"."""
    import torch
    import torch.distributed as dist
    if not _active() or dist.get_world_size() == 1:
        return [t]
    world = dist.get_world_size()
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.cpu()
        outs = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(outs, h)
        return [o.to(t.device) for o in outs]
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return outs


def scatter_sharded_t(pf_bytes: bytes, codes_t, counts_t, n_slots: int):
    """compute_index over ranks with everything resident: `codes_t` (int64, u64 bit patterns: THIS rank's share of the 2-bit key
    codes, on its GPU) and `counts_t` (int32 or None) are scattered through the MPHF into zeroed full-size device arrays
    (aix_index_scatter_shard_codes_dev); the shards are merged where they lie with all_reduce(MAX) on the checker, all_reduce(SUM)
    on tf and on a one-byte-per-slot occupancy (a slot claimed twice, inside a shard or across shards, is the reference's
    collision, hash.cpp:703-709). Returns (checker int64[n_slots], tf int32[n_slots]) device tensors on every rank."""
    import numpy as np
    import torch
    from ._lib import lib, vp, AIX_ERR_CONFLICT, check
    dev = codes_t.device
    checker = torch.empty(n_slots, dtype=torch.int64, device=dev)
    tf = torch.empty(n_slots, dtype=torch.int32, device=dev)
    occ = torch.empty((n_slots + 31) // 32, dtype=torch.int32, device=dev)
    pf = np.frombuffer(pf_bytes, dtype=np.uint8)
    nk = codes_t.numel()
    with torch.cuda.device(dev):
        st = lib().aix_index_scatter_shard_codes_dev(pf.ctypes.data_as(vp), pf.shape[0], vp(codes_t.data_ptr()) if nk else None,
                                                     vp(counts_t.data_ptr()) if counts_t is not None and nk else None, nk, n_slots, dev.index,
                                                     vp(torch.cuda.current_stream(dev).cuda_stream), vp(checker.data_ptr()), vp(tf.data_ptr()),
                                                     vp(occ.data_ptr()))
    err = None
    if st not in (0, AIX_ERR_CONFLICT):             # OOM, HIP error, malformed .pf ...: every rank raises, none is left in a collective
        try:
            check(st, "aix_index_scatter_shard_codes_dev")
        except Exception as e:
            err = e
    _raise_together(err, "scatter_sharded_t", dev if not _active() or _backend_is_nccl() else "cpu")
    bits = ((occ.unsqueeze(1) >> torch.arange(32, device=dev, dtype=torch.int32)) & 1).to(torch.int8).reshape(-1)[:n_slots].contiguous()
    bad = torch.tensor([1 if st == AIX_ERR_CONFLICT else 0], dtype=torch.int32, device=dev)
    _all_reduce_(checker, "max")
    _all_reduce_(tf, "sum")
    _all_reduce_(bits, "sum")
    _all_reduce_(bad, "sum")
    if bool((bits > 1).any().item()) or int(bad.item()):
        raise RuntimeError("hash conflict while scattering (a key outside the MPHF's key set, or a duplicate key)")
    return checker, tf


def _backend_is_nccl():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"


def positions_fill_sharded_t(index, shard_t, base_offset: int, merge: str = "all"):
    """compute_aindex over ranks with everything resident: `shard_t` (uint8, on the index's GPU) is THIS rank's record-aligned
    share of the reads file, `base_offset` its byte offset inside the file. The per-bucket tallies of every shard (u32 each) are
    all-gathered on the device, each rank fills its shard with slot numbering continuing from the shards before it, and the
    full-size partial arrays (one writer per entry, zero elsewhere) are merged in HBM (RCCL under "nccl"):
      merge="all"     all_reduce(SUM): every rank ends with the whole positions array (== the reference's .index.bin);
      merge="scatter" reduce_scatter(SUM): rank r ends with the r-th of `world` equal slices of it (half the link traffic; the
                      slice bounds are returned so that the file can be written by offset).
    Returns (indices int64[n + 1], positions int64[...], (lo, hi)): positions covers entries [lo, hi) of the whole array."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from ._lib import lib, vp, check
    rank, world, _ = rank_world()
    active = _active()
    dev = shard_t.device
    cdev = dev if (not active or _backend_is_nccl()) else "cpu"
    n, k = index.n, index.k
    stream = lambda: vp(torch.cuda.current_stream(dev).cuda_stream)
    # the reference's start adjustment (hash.cpp:973-986) belongs to the beginning of the FILE and carries into the next shard when a
    # shard holds no clean window: every rank looks at the head of its own shard
    head_len, start = 1 << 16, C.c_uint64()
    while True:
        head = shard_t[:head_len].cpu().numpy()
        check(lib().aix_positions_start_k(head.ctypes.data_as(vp) if head.shape[0] else None, head.shape[0], k, C.byref(start)), "aix_positions_start_k")
        if head.shape[0] == shard_t.numel() or start.value + 64 < head.shape[0]:
            break
        head_len *= 16
    exhausted = 1 if (shard_t.numel() < k or start.value >= shard_t.numel() - (k - 1)) else 0
    exs = _all_gather(torch.tensor([exhausted], dtype=torch.int32, device=cdev))
    first = all(int(e.item()) == 1 for e in exs[:rank])
    my_start = start.value if first else 0
    counts = torch.empty(max(n, 1), dtype=torch.int64, device=dev)[:n]
    indices = torch.empty(n + 1, dtype=torch.int64, device=dev)
    err = None
    with torch.cuda.device(dev):
        try:
            check(lib().aix_positions_bucket_counts_dev(index._h, vp(shard_t.data_ptr()) if shard_t.numel() else None, shard_t.numel(), my_start,
                                                        vp(counts.data_ptr()), stream()), "aix_positions_bucket_counts_dev")
            check(lib().aix_positions_indices_dev(index._h, vp(indices.data_ptr()), stream()), "aix_positions_indices_dev")
        except Exception as e:
            err = e
    _raise_together(err, "positions_fill_sharded_t (tally)", cdev)
    # a shard's tally of one bucket never exceeds the bucket's tf (u32): 4 bytes per bucket and rank cross the links
    allc = _all_gather(_u32_bits(counts))
    before = torch.zeros_like(counts)
    for r in range(rank):
        before += allc[r].to(torch.int64) & 0xFFFFFFFF
    filled32 = _u32_bits(before)
    total = int(indices[-1].item())
    per = (total + world - 1) // world if merge == "scatter" else total
    pos = torch.zeros(max(per * world if merge == "scatter" else total, 1), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        try:
            if total:
                check(lib().aix_positions_fill_shard_dev(index._h, vp(shard_t.data_ptr()) if shard_t.numel() else None, shard_t.numel(), my_start, base_offset,
                                                         vp(filled32.data_ptr()), vp(indices.data_ptr()), vp(pos.data_ptr()), stream()),
                      "aix_positions_fill_shard_dev")
        except Exception as e:
            err = e
    _raise_together(err, "positions_fill_sharded_t (fill)", cdev)
    mine, bounds = merge_positions_(pos, total, merge, rank, world)
    return indices, mine, bounds


def scatter_slice_bounds(total: int, rank: int, world: int):
    """merge="scatter": the positions array (total entries) is padded to per * world entries, per = ceil(total / world); rank r keeps
    entries [lo, hi) = [min(r * per, total), min(lo + per, total)) — empty for the ranks past the end."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return per, lo, min(lo + per, total)


def merge_positions_(pos, total: int, merge: str, rank: int, world: int):
    """Merge of the full-size partial positions arrays (one writer per entry, zero elsewhere). pos: max(per * world, 1) entries for
    merge="scatter", max(total, 1) for "all". Returns (this rank's entries, (lo, hi))."""
    import torch
    import torch.distributed as dist
    if merge == "scatter" and _active() and world > 1:
        per, lo, hi = scatter_slice_bounds(total, rank, world)
        if _backend_is_nccl():
            mine = torch.empty(max(per, 1), dtype=pos.dtype, device=pos.device)[:per]
            if per:
                dist.reduce_scatter_tensor(mine, pos[: per * world], op=dist.ReduceOp.SUM)
        else:                                                # gloo has no reduce-scatter: the rehearsal sums everything and cuts
            mine = _all_reduce_(pos, "sum")[rank * per:(rank + 1) * per]
        return mine[: hi - lo], (lo, hi)
    _all_reduce_(pos, "sum")                                # every entry is written by exactly one rank
    return pos[:total], (0, total)


def _u32_bits(t64):
    """int64 values (clamped to 2^32 - 1) as the int32 tensor with the same low 32 bits (u32 bit patterns)."""
    import torch
    v = t64.clamp(min=0, max=(1 << 32) - 1)
    return torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32).contiguous()
