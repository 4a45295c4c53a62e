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
