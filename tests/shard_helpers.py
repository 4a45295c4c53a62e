"""Rank-by-rank replays of the multi-GPU shard protocols (aindex_amd.dist) inside one process, for the GPU tests."""
import ctypes as C

import numpy as np


def positions_by_shards(ix, buf: bytes, world: int):
    """dist.positions_fill_sharded without a process group: per-shard bucket tallies, exclusive sum over earlier shards,
    shard fills into full-size arrays, sum. Returns (indices, positions, first_flags)."""
    from aindex_amd import dist as adist
    from aindex_amd._lib import lib, vp
    bounds = [adist.shard_line_bounds(buf, k, world) for k in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == len(buf) and all(bounds[k][1] == bounds[k + 1][0] for k in range(world - 1))
    first, tallies = [], []
    all_exhausted = True
    for lo, hi in bounds:
        a = np.frombuffer(buf[lo:hi], dtype=np.uint8)
        st = C.c_uint64()
        assert lib().aix_positions_start(a.ctypes.data_as(vp) if a.shape[0] else None, a.shape[0], C.byref(st)) == 0
        first.append(all_exhausted)
        all_exhausted = all_exhausted and (hi - lo < 23 or st.value >= hi - lo - 22)
        tallies.append(ix.positions_bucket_counts(buf[lo:hi], first[-1]))
    indices = ix.positions_indices()
    total = int(indices[-1])
    acc = np.zeros(total, dtype=np.uint64)
    before = np.zeros(ix.n, dtype=np.uint64)
    for (lo, hi), f, t in zip(bounds, first, tallies):
        part = ix.positions_fill_shard(buf[lo:hi], total, f, lo, np.minimum(before, 2 ** 32 - 1).astype(np.uint32))
        assert not np.any((acc != 0) & (part != 0))                   # every slot has one owner
        acc += part
        before += t
    return indices, acc, first


def scatter_by_shards(pf: np.ndarray, keys: np.ndarray, counts: np.ndarray, n_slots: int, cuts):
    """aix_index_scatter_shard over the key ranges [cuts[i], cuts[i+1]); merge = max / sum / or. Returns (checker, tf, occ, statuses)."""
    from aindex_amd._lib import lib, vp
    acc_c, acc_t = np.zeros(n_slots, np.uint64), np.zeros(n_slots, np.uint32)
    acc_o = np.zeros((n_slots + 31) // 32, np.uint32)
    clash = False
    sts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        k = np.ascontiguousarray(keys[lo:hi]); c = np.ascontiguousarray(counts[lo:hi])
        oc, ot, oo = np.empty(n_slots, np.uint64), np.empty(n_slots, np.uint32), np.empty((n_slots + 31) // 32, np.uint32)
        sts.append(lib().aix_index_scatter_shard(pf.ctypes.data_as(vp), pf.shape[0], k.ctypes.data_as(vp), c.ctypes.data_as(vp), hi - lo, n_slots, 0,
                                                 oc.ctypes.data_as(vp), ot.ctypes.data_as(vp), oo.ctypes.data_as(vp)))
        clash = clash or bool(np.any(acc_o & oo))
        acc_c = np.maximum(acc_c, oc); acc_t = acc_t + ot; acc_o |= oo
    return acc_c, acc_t, acc_o, sts, clash
