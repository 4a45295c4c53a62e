"""PCIe-inclusive rates of the host-pointer ABI twins and of the list[str] Python surface."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aindex_amd import engine
from aindex_amd.wrapper import AindexWrapper

ix, g, keys, counts, pf = bench.build_index23(50_000_000, 0, 1, 0, os.path.join(bench.ROOT, ".cache"))
out = {}
N = 100_000_000
q = engine.synth_kmers_t(7, N, 23, 0).cpu().numpy()
for name, arr in (("pageable", q),):
    ix.tf_ascii(arr[: 23 * 1_000_000])
    t = time.perf_counter(); r = ix.tf_ascii(arr); dt = time.perf_counter() - t
    out[f"tf_batch_ascii_host_{name}"] = {"queries": N, "seconds": dt, "lookups_per_s": N / dt, "GBps_in": 23 * N / dt / 1e9}
pinned = torch.from_numpy(q).pin_memory().numpy()
t = time.perf_counter(); r2 = ix.tf_ascii(pinned); dt = time.perf_counter() - t
out["tf_batch_ascii_host_pinned"] = {"queries": N, "seconds": dt, "lookups_per_s": N / dt, "GBps_in": 23 * N / dt / 1e9}
assert np.array_equal(r, r2)
# python list[str] surface on the same handle
w = AindexWrapper()
w._ix23 = ix
M = 5_000_000
strs = [bytes(x).decode() for x in q[: 23 * M].reshape(-1, 23)]
t = time.perf_counter(); res = w.get_tf_values(strs); dt = time.perf_counter() - t
out["AindexWrapper.get_tf_values(list[str])"] = {"queries": M, "seconds": dt, "lookups_per_s": M / dt}
assert res == r[:M].tolist()
t = time.perf_counter(); res2 = w.get_tf_values_array(q[: 23 * M]); dt = time.perf_counter() - t
out["AindexWrapper.get_tf_values_array(uint8[N,23])"] = {"queries": M, "seconds": dt, "lookups_per_s": M / dt}
print(json.dumps(out, indent=1))
