"""N > 1 path on CPU: two gloo ranks, sharded counting + all-reduce == unsharded (see _dist_worker.py)."""
import os
import subprocess
import sys

from aindex_amd import dist as adist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for total in (0, 1, 7, 100, 101):
        for world in (1, 2, 3, 8):
            r = [adist.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_shard_lines_record_aligned():
    buf = b"".join(b"ACGT" * (i % 7 + 1) + b"\n" for i in range(100)) + b"TAIL"
    for world in (1, 2, 3, 8):
        parts = [adist.shard_lines(buf, r, world) for r in range(world)]
        assert b"".join(parts) == buf
        assert all(p == b"" or p.endswith(b"\n") or p is parts[-1] or i == world - 1 for i, p in enumerate(parts))


def test_two_rank_gloo_counting_and_queries():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29677", os.path.join(ROOT, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "DIST_OK" in out, out[-3000:]


def test_three_rank_gloo_with_an_empty_share_and_slice_arithmetic():
    """More ranks than reads: the rank with nothing to count joins every collective (counting all-reduce, the narrow 13-mer reduce,
    the K1 exchange), and the scatter-merge slice bounds hold for totals of 0, < world and not divisible by it (_dist_empty_worker.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", "29679", os.path.join(ROOT, "tests", "_dist_empty_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "DIST_EMPTY_OK" in out, out[-3000:]


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus N` outside a torchrun environment launches the N ranks itself (child processes, rendezvous on
    127.0.0.1), hands their ONE JSON line on with n_gpus = N and leaves with their status. The GPU-free `selftest` workload
    exercises exactly that plumbing here; on the GPU box the same path runs the counting workload (scripts/gpu_dist_rehearsal.sh)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for n in (2, 3):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "selftest", "--steps", "4", "--warmup", "1"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
        lines = r.stdout.decode().strip().splitlines()
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["value"] == n and d["steps"] == 4 and d["warmup"] == 1
    # a rank that dies fails the launch: non-zero status, no JSON line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "selftest"],
                       env=dict(env, AIX_SELFTEST_FAIL_RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and r.stdout.decode().strip() == ""


def test_device_tensor_sharded_counters_exist():
    """No rank of the counting path uploads from host memory: the device-tensor entry points of the N > 1 path."""
    import inspect
    for name in ("count23_sharded_t", "count13_sharded_t"):
        fn = getattr(adist, name)
        src = inspect.getsource(fn)
        assert "all_reduce_sum_" in src and "frombuffer" not in src and ".to(" not in src     # (count13: all_reduce_sum_u64_narrow_)


def test_exchange_merge_single_process_and_owner_hash():
    """Without a process group the exchange is the local merge alone; the owner hash is the splitmix64 finaliser (big-int check)."""
    import torch
    keys = torch.tensor([7, 3, 7, 2 ** 45 + 1, 3, 3], dtype=torch.int64)
    counts = torch.tensor([1, 2, 3, 4, 5, 6], dtype=torch.int64)
    k, c = adist.exchange_merge_counts(keys, counts, 1)
    assert k.tolist() == [3, 7, 2 ** 45 + 1] and c.tolist() == [13, 4, 4]
    k, c = adist.exchange_merge_counts(keys, counts, 5)
    assert k.tolist() == [3] and c.tolist() == [13]

    def sm(x):
        m = (1 << 64) - 1
        z = (x + 0x9E3779B97F4A7C15) & m
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        return (z ^ (z >> 31)) >> 1
    probe = torch.tensor([0, 1, 5, 4 ** 23 - 1, 123456789012, 2 ** 61], dtype=torch.int64)
    for world in (2, 3, 8):
        assert adist._owner_of(probe, world).tolist() == [sm(int(x)) % world for x in probe]
