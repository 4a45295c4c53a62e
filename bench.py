#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X k-mer engine (contract: see the task statement).

Default workload = BASELINE.json configs[2] ("23-mer emphf perfect-hash: build index on CPU, 100M random
23-mer batch lookups on 1 MI355X, HBM GB/s vs roofline"): a ~5e7-key true-canonical 23-mer index of a
synthetic 50 Mbp genome resident in HBM; one step = one batch of 100 M uniform-random 23-mer ASCII
queries (already in HBM) through aix_tf_batch_ascii_dev. N > 1: every rank holds a replica of the index
and its own 100 M queries (weak scaling, no data-path collective).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N                      (N > 1 without a torchrun environment: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 measures what actually scales (BASELINE.json configs[3]): `--total-reads` (200 M) synthetic 150 bp reads
of seed 41 cut into N contiguous read ranges, every rank histograms its range against the fixed 23-mer MPHF index in its own
HBM, ONE RCCL all-reduce(sum) of tf[n]; total work fixed ("scaling": "strong"). The N = 1 line carries the same measurement
under secondary.count23_strong, so the curve is value(N) / secondary.count23_strong.value(1).

Other workloads (own measurements, same JSON shape): --workload count13 | count23 | lookup13 | coverage23 | ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_index23(genome_len, rank, world, device, cache_dir, gpu_builder=False):
    """Synthetic config-3 index, built without the oracle: keys/counts on the GPU, MPHF on the host CPU."""
    import torch
    from aindex_amd import builder, counting, engine, _lib
    from aindex_amd.engine import Index
    t0 = time.time()
    g = engine.synth_genome_t(23, genome_len, device)
    keys, counts = counting.count_distinct_t(g, 23, _lib.CANON_TRUE_RC)
    counts32 = counts.to(torch.int32)
    n = keys.numel()
    torch.cuda.synchronize()
    t1 = time.time()
    pf_path = os.path.join(cache_dir, f"g23_{genome_len}.pf")
    pf = None
    if gpu_builder:
        pf = builder.build_pf_codes_t(keys, 23)       # every rank builds its own (deterministic) copy in HBM
    elif rank == 0:
        if os.path.exists(pf_path):
            pf = open(pf_path, "rb").read()
            if int(np.frombuffer(pf[:8], dtype=np.uint64)[0]) != n:
                pf = None
        if pf is None:
            pf = builder.build_pf_codes(keys.cpu().numpy().view(np.uint64), 23)
            os.makedirs(cache_dir, exist_ok=True)
            with open(pf_path + ".tmp", "wb") as f:
                f.write(pf)
            os.replace(pf_path + ".tmp", pf_path)
    if (world > 1 or os.environ.get("AIX_FORCE_DIST")) and not gpu_builder:
        import torch.distributed as dist
        sz = torch.tensor([len(pf) if rank == 0 else 0], dtype=torch.int64, device=f"cuda:{device}")
        dist.broadcast(sz, 0)
        buf = torch.empty(int(sz.item()), dtype=torch.uint8, device=f"cuda:{device}")
        if rank == 0:
            buf.copy_(torch.frombuffer(bytearray(pf), dtype=torch.uint8))
        dist.broadcast(buf, 0)
        pf = buf.cpu().numpy().tobytes()
    t2 = time.time()
    ix = Index.build_23_codes_t(pf, keys, counts32, device)
    torch.cuda.synchronize()
    t3 = time.time()
    log(f"[rank {rank}] index: n={n} keys/counts {t1 - t0:.1f}s, mphf {t2 - t1:.1f}s, scatter {t3 - t2:.2f}s, "
        f"canonical_only={ix.canonical_only}, HBM {ix.info['device_bytes'] / 1e6:.0f} MB")
    return ix, g, keys, counts32, pf


def timed_steps(step_fn, steps, warmup, device, collective=True):
    """W warm-up steps, then exactly K steps bracketed by barrier + synchronize; per-launch HIP events on
    torch's current stream (the stream the kernels are launched on). collective=False: a measurement only this rank takes
    (no barrier, no max over ranks)."""
    import torch
    from aindex_amd import dist as adist
    for _ in range(warmup):
        step_fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if collective:
        adist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step_fn()
        b.record()
    torch.cuda.synchronize()
    if collective:
        adist.barrier()
    t1 = time.perf_counter()
    wall = adist.all_reduce_max_float(t1 - t0, device=f"cuda:{device}") if collective else t1 - t0
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    return wall, float(np.mean(kern_ms)), kern_ms


def clean_windows(reads_t, n_reads, read_len, k):
    """Windows of k bases without an 'N' in a buffer of fixed-length records, counted with torch in slabs (independent of the library):
    what sum(counts) of a counting workload must equal when the reads are drawn from ACGT genomes with sprinkled Ns."""
    import torch
    rows = reads_t.view(n_reads, read_len + 1)
    total = 0
    for lo in range(0, n_reads, 5_000_000):
        is_n = (rows[lo:lo + 5_000_000, :read_len] == ord("N")).to(torch.int16)
        c = torch.cumsum(is_n, dim=1)
        inside = c[:, k - 1:] - torch.cat([torch.zeros(c.shape[0], 1, dtype=torch.int16, device=c.device), c[:, :-k]], dim=1)
        total += int((inside == 0).sum().item())
        del is_n, c, inside
    return total


def scratch_dir(need_bytes, cache):
    """Where the end-to-end workloads put their input files: /dev/shm (page-cache speed: the measurement is the pipeline, not a disk)
    when it has the room, else the bench cache directory."""
    import shutil
    for d in ("/dev/shm", cache):
        try:
            os.makedirs(d, exist_ok=True)
            if shutil.disk_usage(d).free > need_bytes + (2 << 30):
                return d
        except OSError:
            pass
    return cache


def write_reads_file(path, genome_t, seed, n_reads, fmt, dev, rc_half=False, block=4_000_000):
    """n_reads synthetic 150 bp reads (the seed's stream, reads [0, n_reads)) as a PLAIN (one read per line) or FASTQ file; generated
    in HBM block by block, never held whole on the host. Returns the file size."""
    from aindex_amd import engine
    size = 0
    with open(path, "wb") as f:
        for first in range(0, n_reads, block):
            m = min(block, n_reads - first)
            r = engine.synth_reads_t(seed, genome_t, m, 150, rc_half=rc_half, n_rate_ppm=1000, first_read=first).cpu().numpy()
            if fmt == "fastq":
                fq = np.empty((m, 4 + 151 + 2 + 151), dtype=np.uint8)
                fq[:, :4] = np.frombuffer(b"@r0\n", dtype=np.uint8)
                fq[:, 4:155] = r.reshape(m, 151)
                fq[:, 155:157] = np.frombuffer(b"+\n", dtype=np.uint8)
                fq[:, 157:307] = ord("I")
                fq[:, 307] = ord("\n")
                r = fq.reshape(-1)
            r.tofile(f)
            size += r.shape[0]
    return size


def measure_e2e13(dev, cache, gigabytes, pf13, with_reference=True):
    """END TO END, the tool path: a reads FILE -> bin/count_kmers13 <file> <pf> <out> -> the 512 MiB tf file, wall clock of the whole
    process (python start, HIP init, index load, streamed ingestion, output file). Beside it: the same call inside this process
    (aix_count13_file: no process start, no index load), a FASTQ file of a quarter of the size, the reference binary on a bounded sample."""
    import subprocess
    import torch
    from aindex_amd import engine
    from aindex_amd.engine import Index
    n_reads = int(gigabytes * 1e9) // 151
    d = scratch_dir(int(gigabytes * 1.3e9) + (1 << 30), cache)
    g = engine.synth_genome_t(13, 4_000_000, dev)
    p_plain, p_fq, p_out, p_out2 = (os.path.join(d, f"aix_e2e13_{os.getpid()}{x}") for x in (".txt", ".fq", ".tf.bin", ".tf2.bin"))
    res = {"metric": "end_to_end_count_kmers13", "scratch": d, "reads": n_reads}
    try:
        t0 = time.perf_counter()
        size = write_reads_file(p_plain, g, 14, n_reads, "plain", dev)
        res["file_bytes"] = size
        res["file_written_s"] = time.perf_counter() - t0
        ix = Index.open_13(pf13, None, dev)
        # the answer, from the device-resident path, block by block (tf accumulates in the comparison only)
        want = torch.zeros(4 ** 13, dtype=torch.int64, device=f"cuda:{dev}")
        for first in range(0, n_reads, 8_000_000):
            m = min(8_000_000, n_reads - first)
            want += ix.count13_t(engine.synth_reads_t(14, g, m, 150, n_rate_ppm=1000, first_read=first))
        want = want.cpu().numpy().view(np.uint64)
        # (1) in process: the streamed call alone
        best = None
        for _ in range(2):
            _, st = ix.count13_file(p_plain, p_out2, want_array=False)
            if best is None or st["seconds_total"] < best["seconds_total"]:
                best = st
        assert np.array_equal(np.fromfile(p_out2, dtype=np.uint64), want), "streamed count differs from the device-resident count"
        res["in_process"] = {"seconds": best["seconds_total"], "GBps": size / best["seconds_total"] / 1e9, "reads_per_s": n_reads / best["seconds_total"],
                             "stats": best, "note": "aix_count13_file: file -> pinned parts -> HBM -> table -> 512 MiB output file; index already open"}
        no_out = ix.count13_file(p_plain, None, want_array=True)[1]
        res["in_process_result_to_memory"] = {"seconds": no_out["seconds_total"], "GBps": size / no_out["seconds_total"] / 1e9}
        ix.close()
        torch.cuda.empty_cache()
        # (2) the tool, as a process
        exe = os.path.join(ROOT, "bin", "count_kmers13")
        t0 = time.perf_counter()
        # (a child of a profiled run must not inherit the profiler: its preloaded tool library would attach to the tool as well)
        child_env = {k: v for k, v in os.environ.items() if not (k == "LD_PRELOAD" or k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTX")))}
        child_env["AIX_TOOL_TIMING"] = "1"
        r = subprocess.run([sys.executable, exe, p_plain, pf13, p_out, "16"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900, env=child_env)
        dt = time.perf_counter() - t0
        res["tool_phases"] = [ln for ln in r.stderr.decode().split("\n") if "timing" in ln][-1:]
        if r.returncode != 0:
            raise RuntimeError("bin/count_kmers13 failed: " + r.stderr.decode()[-300:])
        assert np.array_equal(np.fromfile(p_out, dtype=np.uint64), want), "tool output differs from the device-resident count"
        res.update({"value": n_reads / dt, "unit": "reads/s", "seconds": dt, "GBps": size / dt / 1e9,
                    "note": "wall clock of `bin/count_kmers13 <file> <pf> <out>` as a child process, output file byte-equal to the device-resident count"})
        # (3) FASTQ, a quarter of the reads: the normaliser sits in the pipeline
        nq = max(1, n_reads // 4)
        sq = write_reads_file(p_fq, g, 14, nq, "fastq", dev)
        ix = Index.open_13(pf13, None, dev)
        wq = torch.zeros(4 ** 13, dtype=torch.int64, device=f"cuda:{dev}")
        for first in range(0, nq, 8_000_000):
            m = min(8_000_000, nq - first)
            wq += ix.count13_t(engine.synth_reads_t(14, g, m, 150, n_rate_ppm=1000, first_read=first))
        gq, st_first = ix.count13_file(p_fq, None)                  # first streaming call of this handle: allocates its 2.8 GB workspace on the way
        assert np.array_equal(gq, wq.cpu().numpy().view(np.uint64)), "streamed FASTQ count differs from the device-resident count"
        gq, stq = ix.count13_file(p_fq, None)
        assert np.array_equal(gq, wq.cpu().numpy().view(np.uint64)), "streamed FASTQ count differs from the device-resident count"
        res["fastq_in_process"] = {"file_bytes": sq, "reads": nq, "seconds": stq["seconds_total"], "GBps": sq / stq["seconds_total"] / 1e9,
                                   "reads_per_s": nq / stq["seconds_total"], "first_call_seconds": st_first["seconds_total"], "stats": stq}
        ix.close()
        # (4) the reference's own binary on the head of the same file
        ref = os.path.join(ROOT, "oracle", "_ref", "count_kmers13")
        if with_reference and os.path.exists(ref):
            ns = min(n_reads, 200_000)
            p_s = p_plain + ".sample"
            with open(p_plain, "rb") as f, open(p_s, "wb") as o:
                o.write(f.read(ns * 151))
            try:
                t0 = time.perf_counter()
                subprocess.run([ref, p_s, pf13, p_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, check=True, env=child_env)
                dtr = time.perf_counter() - t0
                res["cpu_baseline"] = {"value": ns / dtr, "unit": "reads/s", "cores": os.cpu_count() or 1, "kind": "reference",
                                       "sample": f"first {ns} reads of the same file through oracle/_ref/count_kmers13 (wall clock of the process, its default threads)"}
            except Exception as e:
                log(f"e2e13: reference binary not usable ({type(e).__name__}: {e})")
            finally:
                try:
                    os.remove(p_s)
                except OSError:
                    pass
    finally:
        for f in (p_plain, p_fq, p_out, p_out2):
            try:
                os.remove(f)
            except OSError:
                pass
    return res


def measure_e2e23(ix, g, dev, cache, gigabytes):
    """END TO END for the config-4 histogram: a reads FILE (seed 41, 50 % reverse strand, 0.1 % N) -> aix_count23_fixed_file -> tf[n] in host
    memory, wall clock of the call; the result equals the device-resident count of the same reads."""
    import torch
    from aindex_amd import engine, _lib
    n_reads = int(gigabytes * 1e9) // 151
    d = scratch_dir(int(gigabytes * 1.05e9) + (1 << 30), cache)
    path = os.path.join(d, f"aix_e2e23_{os.getpid()}.txt")
    res = {"metric": "end_to_end_count23_fixed_file", "scratch": d, "reads": n_reads, "index_keys": ix.n}
    try:
        size = write_reads_file(path, g, 41, n_reads, "plain", dev, rc_half=True)
        want = torch.zeros(ix.n, dtype=torch.int32, device=f"cuda:{dev}")
        for first in range(0, n_reads, 8_000_000):
            m = min(8_000_000, n_reads - first)
            ix.count23_fixed_t(engine.synth_reads_t(41, g, m, 150, rc_half=True, n_rate_ppm=1000, first_read=first), _lib.CANON_TRUE_RC, want)
        want = want.cpu().numpy().view(np.uint32)
        best = None
        for _ in range(2):
            got, st = ix.count23_fixed_file(path, _lib.FMT_PLAIN, _lib.CANON_TRUE_RC)
            assert np.array_equal(got, want), "streamed count23 differs from the device-resident count"
            if best is None or st["seconds_total"] < best["seconds_total"]:
                best = st
        res.update({"file_bytes": size, "value": n_reads / best["seconds_total"], "unit": "reads/s", "seconds": best["seconds_total"],
                    "GBps": size / best["seconds_total"] / 1e9, "stats": best,
                    "note": "aix_count23_fixed_file: file -> pinned parts -> HBM -> probe + histogram -> tf[n] in host memory"})
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    return res


def cpu_baseline_lookup23(ix, pf, q_sample_np, gpu_sample, tmpdir):
    """CPU path timed on this host on a bounded sample of the same queries against the same index:
    the compiled reference itself (oracle/_ref/aindex_cpp, single-threaded batch get_tf_values through
    pybind11, as the reference measures it) when present, plus our C restatement (1 thread / all cores)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    os.makedirs(tmpdir, exist_ok=True)
    prefix = os.path.join(tmpdir, "bench23")
    open(prefix + ".pf", "wb").write(pf)
    ix.tf_array().tofile(prefix + ".tf.bin")
    ix.checker_array().tofile(prefix + ".kmers.bin")
    res = {}
    ncores = os.cpu_count() or 1
    orc = O.OracleIndex23.from_prefix(prefix)
    s = q_sample_np.shape[0] // 23
    t = time.perf_counter(); got = orc.tf_batch(q_sample_np, threads=1); dt = time.perf_counter() - t
    assert np.array_equal(got, gpu_sample), "CPU port and GPU disagree on the sample"
    res["port_1t"] = {"value": s / dt, "unit": "lookups/s", "cores": 1, "kind": "port", "sample": f"first {s} of the batch"}
    t = time.perf_counter(); got = orc.tf_batch(q_sample_np, threads=ncores); dt = time.perf_counter() - t
    assert np.array_equal(got, gpu_sample)
    res["port_mt"] = {"value": s / dt, "unit": "lookups/s", "cores": ncores, "kind": "port", "sample": f"first {s} of the batch"}
    del orc
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    try:
        sys.path.insert(0, ref_dir)
        import aindex_cpp
        w = aindex_cpp.AindexWrapper()
        devnull = os.open(os.devnull, os.O_WRONLY)
        so, se = os.dup(1), os.dup(2)
        os.dup2(devnull, 1); os.dup2(devnull, 2)              # the reference logs progress bars to stdout/stderr
        try:
            t = time.perf_counter(); w.load_from_prefix_23mer(prefix); t_ref_load = time.perf_counter() - t
            from aindex_amd.engine import Index as _Index
            t = time.perf_counter()
            with _Index.open_23(prefix + ".pf", prefix + ".tf.bin", prefix + ".kmers.bin", ix.device) as ix2:
                t_our_load = time.perf_counter() - t
                assert ix2.n == ix.n
            res["index_load_s"] = {"reference": t_ref_load, "ours": t_our_load,
                                   "note": "load_from_prefix_23mer of the same three files (P2: hash.cpp:367-450 reads checker and tf element by element); ours = mmap + upload + record build in HBM"}
            sref = min(s, 5_000_000)
            qs = [bytes(x).decode() for x in q_sample_np[: sref * 23].reshape(-1, 23)]
            t = time.perf_counter(); got = w.get_tf_values(qs); dt = time.perf_counter() - t
        finally:
            os.dup2(so, 1); os.dup2(se, 2)
        assert np.array_equal(np.array(got, dtype=np.uint32), gpu_sample[:sref]), "reference and GPU disagree on the sample"
        res["reference"] = {"value": sref / dt, "unit": "lookups/s", "cores": 1, "kind": "reference",
                            "sample": f"first {sref} of the batch via aindex_cpp.get_tf_values(list[str])"}
    except Exception as e:  # reference build absent on this box
        log(f"cpu_baseline: compiled reference not usable here ({type(e).__name__}: {e}); using the port")
    for f in (".pf", ".tf.bin", ".kmers.bin"):
        try:
            os.remove(prefix + f)
        except OSError:
            pass
    return res


def gather_roofline(dev, table_mib=4096, accesses=200_000_000, reps=3):
    """the chip's uniform-random read rate (16-byte reads over a 4 GiB table in HBM), measured live."""
    import torch
    from aindex_amd._lib import lib, check, vp
    nel = table_mib * (1 << 20) // 16
    table = torch.empty(nel * 2, dtype=torch.int64, device=f"cuda:{dev}")
    table.random_(0, 1 << 40)
    sink = torch.zeros(8, dtype=torch.int64, device=f"cuda:{dev}")
    sp = vp(torch.cuda.current_stream().cuda_stream)
    run = lambda: check(lib().aix_bench_gather_dev(vp(table.data_ptr()), nel, 16, 1, accesses, 99, vp(sink.data_ptr()), sp))
    run()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    ms = min(a.elapsed_time(b) for a, b in evs)
    del table
    return accesses / (ms * 1e-3)


def cpu_baseline_count13(ix, reads_t, n_sample, tmpdir, pf13):
    """count_kmers13 on the host for a bounded sample of the same reads: the compiled reference binary (all cores, its
    own "Processing completed" time) when oracle/_ref is present, and our C port; both compared with the GPU result."""
    import re
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    os.makedirs(tmpdir, exist_ok=True)
    sample_t = reads_t[: n_sample * 151]
    gpu = ix.count13_t(sample_t).cpu().numpy().view(np.uint64)
    host = sample_t.cpu().numpy().tobytes()
    ncores = os.cpu_count() or 1
    res = {}
    m = O.OracleMphf(pf13)
    for name, th in (("port_1t", 1), ("port_mt", min(ncores, 64))):
        ns = n_sample if th > 1 else max(1, n_sample // 8)
        buf = host[: ns * 151]
        t = time.perf_counter(); got = O.count13(m, buf, 0, threads=th); dt = time.perf_counter() - t
        if ns == n_sample:
            assert np.array_equal(got, gpu), "CPU port and GPU disagree on the sample"
        res[name] = {"value": ns / dt, "unit": "reads/s", "cores": th, "kind": "port", "sample": f"first {ns} reads of the batch"}
    exe = os.path.join(ROOT, "oracle", "_ref", "count_kmers13")
    if os.path.exists(exe):
        inp, outp = os.path.join(tmpdir, "sample.txt"), os.path.join(tmpdir, "sample.tf.bin")
        open(inp, "wb").write(host)
        try:
            r = subprocess.run([exe, inp, pf13, outp], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
            mm = re.search(rb"Processing completed in (\d+) ms", r.stdout)
            ref = np.fromfile(outp, dtype=np.uint64)
            assert np.array_equal(ref, gpu), "reference count_kmers13 and GPU disagree on the sample"
            if mm and int(mm.group(1)) > 0:
                res["reference"] = {"value": n_sample / (int(mm.group(1)) * 1e-3), "unit": "reads/s", "cores": ncores, "kind": "reference",
                                    "sample": f"first {n_sample} reads of the batch through oracle/_ref/count_kmers13 (its own processing time)"}
        except Exception as e:
            log(f"cpu_baseline: reference count_kmers13 not usable ({type(e).__name__}: {e})")
        for f in (inp, outp):
            try:
                os.remove(f)
            except OSError:
                pass
    return res


def cpu_baseline_distinct23(reads_t, n_sample, gpu_keys, gpu_counts, tmpdir):
    """kmer_counter on the host for a bounded sample of the same reads (FASTA, one record per read): the compiled reference
    binary (its default thread count, wall clock of the whole run: it reads text and writes ./output.txt) when oracle/_ref
    is present, plus our C port; the reference's (k-mer, count) set is compared with the GPU's REF_X86 result."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from aindex_amd import synth
    os.makedirs(tmpdir, exist_ok=True)
    host = reads_t[: n_sample * 151].cpu().numpy().reshape(-1, 151)[:, :150]
    fasta = b"".join(b">r\n" + bytes(r) + b"\n" for r in host)
    res = {}
    t = time.perf_counter(); ok, oc = O.count_distinct(fasta, 23, 1, 1); dt = time.perf_counter() - t
    assert np.array_equal(ok, gpu_keys) and np.array_equal(oc.astype(np.int64), gpu_counts.astype(np.int64)), "CPU port and GPU disagree on the sample"
    res["port_1t"] = {"value": n_sample / dt, "unit": "reads/s", "cores": 1, "kind": "port", "sample": f"first {n_sample} reads of the batch"}
    exe = os.path.join(ROOT, "oracle", "_ref", "kmer_counter")
    if os.path.exists(exe):
        inp = os.path.join(tmpdir, "sample.fa")
        open(inp, "wb").write(fasta)
        try:
            t = time.perf_counter()
            subprocess.run([exe, inp, "23", os.path.join(tmpdir, "unused.txt")], cwd=tmpdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, check=True)
            dt = time.perf_counter() - t
            rows = [ln.split("\t") for ln in open(os.path.join(tmpdir, "output.txt")).read().split("\n") if ln]
            want = dict(zip((bytes(x).decode() for x in synth.decode_kmers(gpu_keys, 23)), (int(c) for c in gpu_counts)))
            assert len(rows) == len(want) and all(want.get(k) == int(c) for k, c in rows), "reference kmer_counter and GPU disagree on the sample"
            res["reference"] = {"value": n_sample / dt, "unit": "reads/s", "cores": os.cpu_count() or 1, "kind": "reference",
                                "sample": f"first {n_sample} reads of the batch through oracle/_ref/kmer_counter (wall clock incl. its text I/O)"}
        except Exception as e:
            log(f"cpu_baseline: reference kmer_counter not usable ({type(e).__name__}: {e})")
        for f in ("sample.fa", "output.txt", "unused.txt"):
            try:
                os.remove(os.path.join(tmpdir, f))
            except OSError:
                pass
    return res


def cpu_baseline_coverage23(ix, pf, seqs_t, L, out_t, per, n_seq, tmpdir):
    """The reference's coverage path — AIndex.get_sequence_coverage's Python loop (aindex/core/aindex.py:314-322) over
    aindex_cpp.get_tf_value — on a few sequences of the batch, compared with the GPU profile."""
    os.makedirs(tmpdir, exist_ok=True)
    prefix = os.path.join(tmpdir, "cov23")
    open(prefix + ".pf", "wb").write(pf)
    ix.tf_array().tofile(prefix + ".tf.bin")
    ix.checker_array().tofile(prefix + ".kmers.bin")
    res = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
        import aindex_cpp
        w = aindex_cpp.AindexWrapper()
        devnull = os.open(os.devnull, os.O_WRONLY)
        se = os.dup(2)
        os.dup2(devnull, 2)
        try:
            w.load_from_prefix_23mer(prefix)
        finally:
            os.dup2(se, 2)
        host = seqs_t[: n_seq * (L + 1)].cpu().numpy().reshape(n_seq, L + 1)
        gpu = out_t[: n_seq * per].cpu().numpy().view(np.uint32).reshape(n_seq, per)
        t = time.perf_counter()
        for i in range(n_seq):
            s = bytes(host[i, :L]).decode()
            cov = [0] * (len(s) - 23 + 1)
            for j in range(len(s) - 23 + 1):
                tf = w.get_tf_value(s[j:j + 23])
                if tf >= 0:
                    cov[j] = tf
            if i < 4:
                assert cov == gpu[i, : L - 22].tolist(), "reference coverage loop and GPU disagree"
        dt = time.perf_counter() - t
        res = {"value": n_seq / dt, "unit": "sequences/s", "cores": 1, "kind": "reference",
               "sample": f"first {n_seq} sequences of the batch through the reference's Python loop over aindex_cpp.get_tf_value"}
    except Exception as e:
        log(f"cpu_baseline: reference coverage not usable ({type(e).__name__}: {e})")
    for f in (".pf", ".tf.bin", ".kmers.bin"):
        try:
            os.remove(prefix + f)
        except OSError:
            pass
    return res


def load_pmc_traffic(key, **must_match):
    """PMC-measured HBM-side bytes per launch (profiles/pmc_traffic.json, filled from the rocprofv3 --pmc passes of
    scripts/gpu_profile_all.sh). Attached only when the entry was measured with the same per-launch size and settings as this
    run; anything else gets None (a stale figure is worse than none)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        e = json.load(open(p)).get(key)
    except Exception:
        return None
    if not e:
        return None
    for k, v in must_match.items():
        if e.get(k) != v:
            return None
    return e


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a, argv):
    """`bench.py --gpus N` (N > 1) outside a torchrun environment: start the N ranks as a CHILD process (torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1), hand its single JSON line on and leave with its status. This process has not
    touched the GPU (device_count() does not initialise it), and nothing is exec'd."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if a.workload != "selftest":
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus and not env.get("AIX_BENCH_ONE_DEVICE"):
            # fewer devices than ranks (a 1-GPU box): rehearsal mode — every rank on device 0, gloo collectives. The JSON says so.
            log(f"bench.py: {have} device(s) for {a.gpus} ranks: rehearsal mode (AIX_BENCH_ONE_DEVICE=1, gloo)")
            env["AIX_BENCH_ONE_DEVICE"] = "1"
        if env.get("AIX_BENCH_ONE_DEVICE"):
            env.setdefault("AIX_DIST_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    log("bench.py: launching", " ".join(cmd))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        log("bench.py: the ranks printed no JSON line")
        return 1
    return r.returncode


def tf_checksum(tf_t):
    """Two order-independent digests of a tf[] histogram (device tensor of u32 bit patterns): the plain sum and a
    position-weighted sum mod 2^63. Equal digests at N = 1 and N = 8 <=> the sharded + all-reduced histogram is the unsharded one."""
    import torch
    t = tf_t.to(torch.int64) & 0xFFFFFFFF
    w = (torch.arange(t.numel(), dtype=torch.int64, device=t.device) % 1000003) + 1
    return {"sum": int(t.sum().item()), "weighted": int(((t * w).sum() & 0x7FFFFFFFFFFFFFFF).item())}


def measure_lookup23(ix, g, qset, queries, rank, world, dev, steps, warmup, gather_probe):
    """One query set of SURVEY 8(d) through aix_tf_batch_ascii_dev: Q_rand = uniform-random 23-mers (seed 7; ~all absent),
    Q_mix = 50 % genome windows on a random strand + 50 % random (seed 8). Queries are generated in HBM; every rank its own."""
    import torch
    from aindex_amd import engine
    if qset == "Q_mix":
        q = engine.synth_mix23_t(8, g, queries, first=rank * queries)
    else:
        q = engine.synth_kmers_t(7, queries, 23, dev, first=rank * queries)
    res = torch.empty(queries, dtype=torch.int32, device=f"cuda:{dev}")
    wall, kern_ms, _ = timed_steps(lambda: ix.tf_ascii_t(q, res), steps, warmup, dev)
    hf = int((res != 0).sum().item()) / queries
    info = ix.info
    # what the kernel touches (instrumented launch of the same kernel, same settings), per query: MPHF records, key records,
    # evaluations carried through to the rank, bucket lines of the verification table
    li = ix.lines_ascii_t(q).to(torch.int64)
    mphf_recs = float((li & 15).sum().item()) / queries
    key_recs = float(((li >> 4) & 15).sum().item()) / queries
    completed = float(((li >> 8) & 15).sum().item()) / queries
    filter_words = float(((li >> 12) & 15).sum().item()) / queries
    bucket_lines = float((li >> 16).sum().item()) / queries
    del li
    t = kern_ms * 1e-3
    # bytes this kernel REQUESTS: the query in and the answer out (23 + 4), 128 per bucket line, 12 per record of the early-exit
    # walk (pairs + prefix + the presence dword) or 16 per record of the parallel evaluation, 16 per key record
    rec_bytes = 12.0 if info_flag(ix, "early_exit") else 16.0
    requested = 27.0 + 8.0 * filter_words + 128.0 * bucket_lines + rec_bytes * mphf_recs + 16.0 * key_recs
    lines = filter_words + bucket_lines + mphf_recs + key_recs
    ach = requested * queries / t / 1e9
    line_gbs = (lines * 128.0 + 27.0) * queries / t / 1e9
    # the reference algorithm's bytes for the same queries (104 B forward hit, 204 B reverse hit, 200 B miss, + 27 B streamed)
    ref_bytes = 27.0 + (200.0 * (1.0 - hf) + 154.0 * hf)
    roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "kernel": "k_lookup23_ascii", "kernel_ms": kern_ms, "queries_per_launch": queries,
            "requested_bytes_per_query": requested,
            "note": "achieved = bytes the kernel requests (27 streamed + 8 per absence-filter word + 128 per bucket line + 12/16 per MPHF record + 16 per key record, as "
                    "counted by an instrumented launch) x queries / launch time; every random access moves a whole 128-byte line, see line_traffic",
            "line_traffic": {"GBps": line_gbs, "frac": line_gbs / HBM_PEAK_GBS, "lines_per_query": lines,
                             "note": "128-byte lines x accesses + streamed bytes, per second (estimate from the instrumented launch; `traffic` is the "
                                     "PMC figure: fabric-side bytes, Infinity-Cache hits included)"},
            "reference_algorithm": {"bytes_per_query": ref_bytes, "GBps": ref_bytes * queries / t / 1e9,
                                    "note": "SURVEY 8(d): what mphf::lookup x probes + checker + tf would move for these queries — for orientation, "
                                            "NOT what this kernel moves (no frac is derived from it)"}}
    if gather_probe:
        peak_acc = gather_roofline(dev)
        acc = lines * queries / t
        roof["random_read"] = {"peak_accesses_per_s": peak_acc, "achieved_accesses_per_s": acc, "frac": acc / peak_acc, "accesses_per_query": lines,
                               "note": "north_star's yardstick: peak = k_gather, uniform-random 16-byte reads over a 4 GiB table in HBM (SURVEY 8d-ii), "
                                       "measured in this run; accesses served by L2 / Infinity Cache can push the ratio above 1"}
    tr = load_pmc_traffic("lookup23:" + qset, queries_per_launch=queries, bucket_table=int(info["bucket_table"]), absence_filter=int(info["absence_filter_words"] > 0))
    if tr:
        roof["traffic"] = tr.get("bytes_per_launch")
        roof["traffic_source"] = tr.get("source")
    cfg = {"workload": f"configs[2]: 23-mer batch lookup (aix_tf_batch_ascii_dev), {qset}: " +
                       ("50 % windows of the indexed genome on a random strand + 50 % uniform-random 23-mers (seed 8)" if qset == "Q_mix"
                        else "uniform-random 23-mer ASCII queries (seed 7)") + ", resident in HBM",
           "query_set": qset, "query_seed": 8 if qset == "Q_mix" else 7, "queries_per_step_per_gpu": queries, "index_keys": ix.n,
           "hit_fraction": hf, "bucket_table": bool(info["bucket_table"]), "bucket_lanes": info["bucket_lanes"], "buckets": info["buckets"],
           "bucket_unfiled_keys": info["bucket_unfiled_keys"], "canonical_only": bool(info["canonical_only"]),
           "absence_filter_words": info["absence_filter_words"], "filter_words_per_query": filter_words,
           "bucket_lines_per_query": bucket_lines, "mphf_records_read_per_query": mphf_recs, "key_records_read_per_query": key_recs,
           "completed_evaluations_per_query": completed, "index_hbm_bytes": info["device_bytes"]}
    return {"value": world * queries * steps / wall, "ms_per_step": wall / steps * 1e3, "roofline": roof, "config": cfg, "_q": q, "_res": res}


_FLAGS = {}


def apply_ab_switches(ix, a):
    if a.no_fastpath:
        ix.set_canonical_fastpath(False)
    if a.no_fingerprint:
        ix.set_fingerprint_filter(False)
    if a.no_early_exit:
        ix.set_early_exit(False)
        _FLAGS[(id(ix), "early_exit")] = False
    globals()["_BUCKET_LANES_GIVEN"] = bool(a.bucket_lanes) or bool(os.environ.get("AIX_BUCKET_LANES"))
    if a.no_bucket_table or a.bucket_lanes:
        ix.set_bucket_table(not a.no_bucket_table, a.bucket_lanes)
    if a.no_bucket_table and not a.no_early_exit:
        ix.set_early_exit(True)                 # the round-1 walk: its presence masks are built on request since round 3 (2.5 B per key)
    if a.no_absence_filter:
        ix.set_absence_filter(False)
    if a.no_minimizer_table:
        ix.set_minimizer_table(False)


def info_flag(ix, name):
    """Host-side mirror of the A/B switches bench.py itself flipped (the library does not report them back)."""
    return _FLAGS.get((id(ix), name), True)


_BUCKET_LANES_GIVEN = False          # set by main(): --bucket-lanes / AIX_BUCKET_LANES pin the lanes per bucket line for every consumer


def count23_roofline(ix, windows, reads, kern_ms):
    """Roofline object of k_count23_fixed. `achieved` = the bytes the kernel REQUESTS (its own record sizes) per launch / the
    launch's duration: 151/128 input bytes per window (each byte is fetched once from HBM, re-reads by the neighbouring lanes
    hit L1), one probe of the verification table per valid window, one 4-byte counter RMW (4 read + 4 written, memory side).
    The reference algorithm's figure for the same windows (1.17 + 154 + 8 B) is kept beside it for orientation."""
    p = ix.probe_profile()
    if ix.info["count23_backend"] == 3:
        # the call counted the distinct k-mers of the reads first (K1) and probed each of them once: what moves per window is K1's traffic (the MSD
        # path: 1.18 B of reads in, 8 B codes written once, read and written at level 1, read twice + 4 B remainders at level 2, read again), plus
        # one probe and 24 B per DISTINCT k-mer — not a table line per window. The pipeline is bound by LDS atomics, not by HBM.
        per_window = 151.0 / 128.0 + 8.0 + 16.0 + 20.0 + 4.0
        per_key = 24.0 + p["bytes_per_hit_probe"] + 8.0
        achieved = (windows * per_window + ix.n * per_key) / (kern_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": "k_k1_split / k_k1_count / k_k1_scatter / k_k1_final / k_k1_gather per piece + k_merge_tile + k_add_counts23 (aix_count23_fixed_dev, back end 3)",
                "kernel_ms": kern_ms, "requested_bytes_per_window": per_window, "requested_bytes_per_distinct_kmer": per_key, "windows_per_launch": windows,
                "reads_per_launch": reads, "windows_per_key": windows / max(1, ix.n),
                "note": "not HBM-bound: one returning LDS atomic per window and partition level; the probe path (back end 2, AIX_COUNT23_VIA_K1=0) requests "
                        f"{151.0 / 128.0 + p['bytes_per_hit_probe'] + 8.0:.1f} B per window and ran at 617 - 679 ms on this workload (DESIGN.md 5)",
                "reference_algorithm": {"bytes_per_window": 1.17 + 154.0 + 8.0, "GBps": windows * (1.17 + 154.0 + 8.0) / (kern_ms * 1e-3) / 1e9,
                                        "note": "SURVEY 8(d): what mphf::lookup + checker + tf would move per window; not what this path moves"}}
    per_window = 151.0 / 128.0 + p["bytes_per_hit_probe"] + 8.0
    achieved = windows * per_window / (kern_ms * 1e-3) / 1e9
    lines = p["lines_per_hit_probe"]
    pmc = load_pmc_traffic("count23")                    # measured on a 10 M-read launch; `traffic` itself is only filled for that launch size
    per_window_pmc = (pmc["bytes_per_launch"] / pmc["windows_per_launch"]) if (pmc and pmc.get("windows_per_launch")) else None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "traffic_per_window_pmc": per_window_pmc,
            "kernel": "k_count23_fixed", "kernel_ms": kern_ms, "requested_bytes_per_window": per_window, "lines_per_window": lines,
            "windows_per_launch": windows, "reads_per_launch": reads,
            "probe": p["name"].replace("(8 lanes per line)", "(the counter's slot probe reads a line with 2 lanes unless --bucket-lanes says otherwise)") if not _BUCKET_LANES_GIVEN else p["name"],
            "line_traffic_estimate": {"GBps": windows * (lines * 128.0 + 151.0 / 128.0 + 128.0) / (kern_ms * 1e-3) / 1e9,
                                      "note": "128-byte lines the probes and the counter RMW move (estimate; `traffic` is the PMC figure)"},
            "reference_algorithm": {"bytes_per_window": 1.17 + 154.0 + 8.0,
                                    "GBps": windows * (1.17 + 154.0 + 8.0) / (kern_ms * 1e-3) / 1e9,
                                    "note": "SURVEY 8(d): what mphf::lookup + checker + tf would move per window; not what this kernel moves"}}


def cpu_baseline_count23(ix, g, pf, dev, cache, ns=100_000):
    """The reference has no tool for this composition (kmer_counter -> compute_index would re-derive the key set); the CPU path
    timed beside it is the C restatement of the same histogram (forward-then-rc MPHF probes per window), 1 thread, on the first
    `ns` reads of the workload, compared with the GPU histogram of the same reads."""
    import torch
    from aindex_amd import engine, _lib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    tmpd = os.path.join(cache, "cpu23c")
    os.makedirs(tmpd, exist_ok=True)
    prefix = os.path.join(tmpd, "c23")
    open(prefix + ".pf", "wb").write(pf)
    ix.tf_array().tofile(prefix + ".tf.bin")
    ix.checker_array().tofile(prefix + ".kmers.bin")
    orc = O.OracleIndex23.from_prefix(prefix)
    sample = engine.synth_reads_t(41, g, ns, 150, rc_half=True, n_rate_ppm=1000, first_read=0)
    gpu_s = torch.zeros(ix.n, dtype=torch.int32, device=f"cuda:{dev}")
    ix.count23_fixed_t(sample, _lib.CANON_TRUE_RC, gpu_s)
    t0 = time.perf_counter(); want = orc.count23_fixed(sample.cpu().numpy().tobytes(), False, 2); dt = time.perf_counter() - t0
    assert np.array_equal(want, gpu_s.cpu().numpy().view(np.uint32)), "CPU port and GPU disagree on the sample"
    del orc
    for f in (".pf", ".tf.bin", ".kmers.bin"):
        os.remove(prefix + f)
    return {"value": ns / dt, "unit": "reads/s", "cores": 1, "kind": "port", "sample": f"first {ns} reads of the workload (seed 41)"}


def measure_count23_strong(ix, g, rank, world, dev, total_reads, steps, warmup):
    """BASELINE configs[3] / SURVEY 8(d) config 4: `total_reads` reads x 150 bp (seed 41, 50 % reverse strand, 0.1 % N) cut
    into `world` contiguous read ranges; rank r generates ITS range in its own HBM (nothing comes from the host), histograms it
    against the fixed MPHF index (k_count23_fixed, true canonical form) and ONE all-reduce(sum) merges tf[n] (RCCL over xGMI
    under "nccl"). Total work is fixed as `world` grows."""
    import torch
    from aindex_amd import dist as adist, engine, _lib
    lo, hi = adist.shard_range(total_reads, rank, world)
    reads = engine.synth_reads_t(41, g, hi - lo, 150, rc_half=True, n_rate_ppm=1000, first_read=lo)
    tf = torch.zeros(ix.n, dtype=torch.int32, device=f"cuda:{dev}")
    step = lambda: adist.count23_sharded_t(ix, reads, _lib.CANON_TRUE_RC, tf)
    wall, kern_ms, _ = timed_steps(step, steps, warmup, dev)
    digest = tf_checksum(tf)
    ar_ms = 0.0
    if world > 1 or os.environ.get("AIX_FORCE_DIST"):
        scratch = tf.clone()
        w_ar, _, _ = timed_steps(lambda: adist.all_reduce_sum_(scratch), 3, 1, dev)
        ar_ms = w_ar / 3 * 1e3
        del scratch
    # kernel alone on this rank's share (no zeroing, no collective): the figure the roofline object is built from
    k_only, k_ms, _ = timed_steps(lambda: ix.count23_fixed_t(reads, _lib.CANON_TRUE_RC, tf), 2, 0, dev)
    # every window without N of every rank's reads is a key of the index (genome reads), so the merged histogram must sum to their number,
    # counted independently with torch on each rank's own reads
    clean = torch.tensor([clean_windows(reads, hi - lo, 150, 23)], dtype=torch.int64, device=reads.device)
    del reads
    import torch.distributed as dist
    backend = dist.get_backend() if (dist.is_available() and dist.is_initialized()) else "none"
    adist.all_reduce_sum_(clean)
    if int(clean.item()) != digest["sum"]:
        raise SystemExit(f"count23_strong: sum(tf) = {digest['sum']}, windows without N = {int(clean.item())}")
    windows_rank = (hi - lo) * (150 - 22)
    # N > 1: rank 0 then runs the WHOLE workload alone (same reads, same kernel, no collective) — the 1-GPU rate of this very
    # metric measured in the same process, and the histogram the sharded + all-reduced one must equal (digest of all tf[] entries)
    one = None
    if world > 1 and not os.environ.get("AIX_BENCH_NO_SINGLE"):
        if rank == 0:
            try:
                all_reads = engine.synth_reads_t(41, g, total_reads, 150, rc_half=True, n_rate_ppm=1000, first_read=0)
                tf1 = torch.zeros(ix.n, dtype=torch.int32, device=f"cuda:{dev}")
                def step1():
                    tf1.zero_()
                    ix.count23_fixed_t(all_reads, _lib.CANON_TRUE_RC, tf1)
                w1, _, _ = timed_steps(step1, 2, 1, dev, collective=False)
                d1 = tf_checksum(tf1)
                one = {"value": total_reads * 2 / w1, "unit": "reads/s", "ms_per_step": w1 / 2 * 1e3, "tf_digest": d1,
                       "equals_sharded_result": d1 == digest, "note": "rank 0 alone on all reads, after the timed region"}
                del all_reads, tf1
            except Exception as e:  # pragma: no cover
                one = {"error": f"{type(e).__name__}: {e}"}
        adist.barrier()
    return {"metric": "reads_per_sec_23mer_count_fixed_mphf", "value": total_reads * steps / wall, "unit": "reads/s",
            **({"same_workload_on_one_gpu": one} if one else {}),
            "scaling": "strong", "total_reads": total_reads, "reads_this_rank": hi - lo, "ms_per_step": wall / steps * 1e3,
            "allreduce_ms": ar_ms, "allreduce_bytes": 4 * ix.n, "collective": f"all_reduce(sum) of int32 tf[{ix.n}]" if backend != "none" else "none (1 rank)",
            "backend": backend, "collective_ranks": world if backend != "none" else 1,
            "kernel": "aix_count23_fixed_dev", "kernel_ms_this_rank": k_ms, "windows_this_rank": windows_rank,
            "counting_backend": {1: "memory-side atomics", 2: "slot stream + LDS histogram", 3: "distinct k-mers first (K1), one probe per distinct k-mer"}.get(ix.info["count23_backend"], "none"),
            "histogram_passes": ix.info["count23_passes"],
            "windows_counted_all_ranks": digest["sum"], "windows_without_N_all_ranks": int(clean.item()), "tf_digest": digest,
            "one_device_rehearsal": bool(os.environ.get("AIX_BENCH_ONE_DEVICE"))}


def run_selftest(a, real_stdout):
    """Launcher / rendezvous check that needs no GPU: `world` gloo ranks all-reduce their rank numbers; rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    from aindex_amd import dist as adist
    rank, world, _ = adist.init("gloo")
    if os.environ.get("AIX_SELFTEST_FAIL_RANK") == str(rank):      # test hook: a rank that dies must fail the whole launch
        raise SystemExit(7)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    adist.all_reduce_sum_(t)
    assert int(t.item()) == world * (world + 1) // 2
    if rank == 0:
        os.write(real_stdout, (json.dumps({"metric": "selftest_ranks", "value": world, "unit": "ranks", "n_gpus": world, "steps": a.steps,
                                           "warmup": a.warmup, "config": {"workload": "launcher self-test (gloo all-reduce on CPU tensors)"}}) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)          # the first ~9 launches after the index build run 4 % slower (profiles/r03/default_lookup_launches.txt)
    ap.add_argument("--workload", default="auto", choices=["auto", "lookup23", "lookup13", "count13", "count23", "gather", "coverage23", "coverage13", "positions23", "normalize", "distinct23", "e2e13", "e2e23", "selftest"],
                    help="auto: N = 1 -> lookup23 (BASELINE configs[2], the headline), N > 1 -> count23 --scaling strong (configs[3])")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="count23: strong = --total-reads split over the ranks (config 4); weak = --reads per rank")
    ap.add_argument("--total-reads", type=int, default=200_000_000, help="reads of the strong-scaling counting workload (config 4: 200 M)")
    ap.add_argument("--seqs", type=int, default=1_000_000, help="sequences of the coverage workloads (config 5: 1 M x 10 kbp)")
    ap.add_argument("--seq-len", type=int, default=10_000)
    ap.add_argument("--table-mib", type=int, default=4096)
    ap.add_argument("--elem", type=int, default=16, help="gather: bytes a lane reads per access (4, 8, 16, 32, 64, 128)")
    ap.add_argument("--unroll", type=int, default=1)
    ap.add_argument("--queries", type=int, default=100_000_000)
    ap.add_argument("--genome", type=int, default=50_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--cpu-sample", type=int, default=20_000_000)
    ap.add_argument("--cpu-reads", type=int, default=400_000, help="reads in the bounded CPU sample of the counting workloads")
    ap.add_argument("--positions-tf", default="reads", choices=["reads", "genome"],
                    help="positions23: tf of the index = occurrences in the reads (every found window is placed; the reference pipeline) or in the genome (tf = 1: first occurrences only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather-probe", action="store_true", help="skip the live random-read roofline measurement")
    ap.add_argument("--e2e-gb", type=float, default=8.0, help="size of the reads file of the end-to-end (file in, file out) measurements")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end tool measurements of the default workload's secondary block")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary counting measurement of the default workload")
    ap.add_argument("--reads13", type=int, default=10_000_000, help="reads per rank and step of the secondary 13-mer counting measurement (configs[1])")
    ap.add_argument("--reads23", type=int, default=10_000_000, help="reads per rank and step of the secondary counting measurement (config 4 has 25 M per GPU at N = 8)")
    ap.add_argument("--no-fastpath", action="store_true", help="force the reference's two-probe order")
    ap.add_argument("--no-fingerprint", action="store_true", help="disable the 4-bit fingerprint filter")
    ap.add_argument("--no-early-exit", action="store_true", help="disable the early-exit MPHF walk (presence masks)")
    ap.add_argument("--no-bucket-table", action="store_true", help="switch the verification table off (every probe through the MPHF records + key records)")
    ap.add_argument("--probe-path", action="store_true", help="count23: keep the per-window probe + LDS-histogram back end (AIX_COUNT23_VIA_K1=0) where the call would count distinct k-mers first")
    ap.add_argument("--no-minimizer-table", action="store_true", help="streaming consumers (count23, coverage, positions) probe the hash-keyed table: one HBM line per window")
    ap.add_argument("--no-absence-filter", action="store_true", help="switch the Bloom filter in front of the verification table off")
    ap.add_argument("--bucket-lanes", type=int, default=0, choices=[0, 1, 2, 4, 8], help="lanes that share one bucket read (0: the library's default)")
    ap.add_argument("--gpu-builder", action="store_true", help="build the MPHF on the GPU (parallel peeling) instead of the host")
    ap.add_argument("--query-mix", action="store_true", help="Q_mix: 50 %% genome windows on a random strand + 50 %% random (seed 8)")
    a = ap.parse_args()
    if a.probe_path:
        os.environ["AIX_COUNT23_VIA_K1"] = "0"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, sys.argv[1:]))        # child processes; this one never touches the GPU
    world_env = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus != world_env:
        log(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world_env}: the launcher's world size is what runs")
    if a.workload == "auto":
        a.workload = "lookup23" if world_env == 1 else "count23"
        if world_env > 1 and a.scaling is None:
            a.scaling = "strong"
    if a.scaling is None:
        a.scaling = "weak"
    # stdout carries exactly ONE JSON line (rank 0): libraries that print banners to fd 1 (RCCL does at init) go to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if a.workload == "selftest":
        return run_selftest(a, real_stdout)

    import torch
    from aindex_amd import dist as adist, engine, _lib
    rank, world, local = adist.init()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    if os.environ.get("AIX_BENCH_ONE_DEVICE"):      # rehearsal of the N>1 path on a 1-GPU box (gloo backend)
        local = 0
    torch.cuda.set_device(local)
    dev = local
    cache = os.path.join(ROOT, ".cache")
    out = {"n_gpus": world, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "data": "synthetic"}

    if a.workload == "lookup23":
        ix, g, keys, counts, pf = build_index23(a.genome, rank, world, dev, cache, a.gpu_builder)
        del keys, counts
        apply_ab_switches(ix, a)
        qset = "Q_mix" if a.query_mix else "Q_rand"
        m = measure_lookup23(ix, g, qset, a.queries, rank, world, dev, a.steps, a.warmup, not a.no_gather_probe)
        q, res = m.pop("_q"), m.pop("_res")
        value = m["value"]
        out.update({"metric": "kmer_lookups_per_sec_23mer_batch", "value": value, "unit": "lookups/s", "ms_per_step": m["ms_per_step"], "dtype": "u64",
                    "config": {**m["config"], "genome_bp": a.genome, "parallelism": f"replica x{world}"},
                    "roofline": m["roofline"]})
        # BASELINE.json.published is {}; the reference's README figure is kept for orientation only (hardware unstated,
        # measured through Python list[str] on an unshipped index), so vs_baseline stays null
        out["published_reference_rate"] = {"value": 2.3e6, "unit": "lookups/s", "source": "reference README.md:14,480,596 (BASELINE.md section 1)",
                                           "ratio": value / 2.3e6}
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            s_ = min(a.cpu_sample, a.queries)
            qs = q[: s_ * 23].cpu().numpy()
            cb = cpu_baseline_lookup23(ix, pf, qs, res[:s_].cpu().numpy().view(np.uint32), os.path.join(cache, "cpu"))
            if "index_load_s" in cb:
                out["index_load_s"] = cb.pop("index_load_s")
            out["cpu_baseline"] = cb.get("reference", cb["port_1t"])
            out["cpu_baseline_extra"] = {k: v for k, v in cb.items()}
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        del q, res
        if not a.no_secondary:
            sec = {}
            # the other query set of SURVEY 8(d): Q_mix (50 % hits) when the headline is Q_rand, and the other way round
            try:
                other = "Q_rand" if a.query_mix else "Q_mix"
                m2 = measure_lookup23(ix, g, other, a.queries, rank, world, dev, max(3, a.steps // 4), 1, False)
                m2.pop("_q"); m2.pop("_res")
                sec["lookup23_" + other] = {"metric": "kmer_lookups_per_sec_23mer_batch", "value": m2["value"], "unit": "lookups/s", "ms_per_step": m2["ms_per_step"],
                                            "config": m2["config"], "roofline": m2["roofline"]}
            except Exception as e:  # pragma: no cover
                sec["lookup23_other_query_set"] = {"error": f"{type(e).__name__}: {e}"}
            # the other half of BASELINE.json's metric: reads/s counted — config 4 at its full size (200 M reads; N ranks share them),
            # tf[] merged by one all-reduce (RCCL over xGMI at N > 1). Never allowed to take the headline number down with it.
            try:
                r = measure_count23_strong(ix, g, rank, world, dev, a.total_reads, 3, 1)
                r["roofline"] = count23_roofline(ix, r["windows_this_rank"], r["reads_this_rank"], r["kernel_ms_this_rank"])
                tr = load_pmc_traffic("count23", reads_per_launch=r["reads_this_rank"])
                if tr:
                    r["roofline"]["traffic"] = tr.get("bytes_per_launch")
                    r["roofline"]["traffic_source"] = tr.get("source")
                sec["count23_strong"] = r
            except Exception as e:  # pragma: no cover
                sec["count23_strong"] = {"error": f"{type(e).__name__}: {e}"}
            # BASELINE configs[1]: 13-mer dense 4^13 table, 10 M reads per rank, u64 table merged by one all-reduce
            try:
                from aindex_amd.engine import Index as _Index13
                from aindex_amd.builder import all_13mers_pf_path as pf13_path
                ix13 = _Index13.open_13(pf13_path(), None, dev)
                g13 = engine.synth_genome_t(13, 4_000_000, dev)
                reads13 = engine.synth_reads_t(14, g13, a.reads13, 150, n_rate_ppm=1000, first_read=rank * a.reads13)
                tf13 = torch.empty(4 ** 13, dtype=torch.int64, device=f"cuda:{dev}")
                w13, k13, _ = timed_steps(lambda: adist.count13_sharded_t(ix13, reads13, tf13), 3, 1, dev)
                total13 = int(tf13.sum().item())
                bits13 = 0
                if world > 1:                                  # the table crosses the links as u32 (256 MiB) when no counter can reach 2^32
                    probe13 = tf13 // world                    # same magnitude as one rank's table
                    bits13 = adist.all_reduce_sum_u64_narrow_(probe13)[1]
                    ar13, _, _ = timed_steps(lambda: adist.all_reduce_sum_u64_narrow_(probe13.clone()), 3, 1, dev)
                    del probe13
                else:
                    ar13 = 0.0
                sec["count13_dense"] = {"metric": "reads_per_sec_13mer_count", "value": world * a.reads13 * 3 / w13, "unit": "reads/s", "scaling": "weak",
                                        "reads_per_step_per_gpu": a.reads13, "ms_per_step": w13 / 3 * 1e3, "allreduce_ms_of_it": ar13 / 3 * 1e3,
                                        "windows_counted_all_ranks": total13,
                                        "collective": (f"all_reduce(sum) of tf[4^13] as u{bits13} ({(4 ** 13) * bits13 // 8 >> 20} MiB per rank)" if world > 1 else "none (1 rank)"),
                                        "collective_ranks": world, "backend": (torch.distributed.get_backend() if world > 1 else "none")}
                ix13.close()
                del reads13, tf13, g13
            except Exception as e:  # pragma: no cover
                sec["count13_dense"] = {"error": f"{type(e).__name__}: {e}"}
            # the tool path end to end: reads FILE -> bin/count_kmers13 -> tf file, and the config-4 histogram from a file
            if rank == 0 and world == 1 and not a.no_e2e:
                try:
                    from aindex_amd.builder import all_13mers_pf_path as pf13_path
                    sec["e2e13"] = measure_e2e13(dev, cache, a.e2e_gb, pf13_path(), with_reference=not a.no_cpu_baseline)
                except Exception as e:  # pragma: no cover
                    sec["e2e13"] = {"error": f"{type(e).__name__}: {e}"}
                try:
                    sec["e2e23"] = measure_e2e23(ix, g, dev, cache, a.e2e_gb)
                except Exception as e:  # pragma: no cover
                    sec["e2e23"] = {"error": f"{type(e).__name__}: {e}"}
            out["secondary"] = sec

    elif a.workload == "e2e13":
        from aindex_amd.builder import all_13mers_pf_path as pf13_path
        r = measure_e2e13(dev, cache, a.e2e_gb, pf13_path(), with_reference=not a.no_cpu_baseline)
        cb = r.pop("cpu_baseline", None)
        out.update({"metric": "reads_per_sec_count_kmers13_tool_end_to_end", "value": r["value"], "unit": "reads/s", "ms_per_step": r["seconds"] * 1e3, "steps": 1, "warmup": 0,
                    "dtype": "u64", "config": {"workload": f"bin/count_kmers13 on a {r['file_bytes'] / 1e9:.2f} GB PLAIN reads file (150 bp, seed 14) in {r['scratch']}: "
                                                           "process start + index load + streamed ingestion + 512 MiB output file", **{k: v for k, v in r.items() if k not in ("value", "unit")}},
                    **({"cpu_baseline": cb} if cb else {}),
                    "roofline": {"bound": "hbm", "achieved": r["in_process"]["GBps"], "peak": 64.0, "unit": "GB/s", "frac": r["in_process"]["GBps"] / 64.0, "traffic": None,
                                 "kernel": "H2D copies of the parts (the link, not a kernel, bounds this path)",
                                 "note": "peak = PCIe 5 x16 (64 GB/s raw; ~51 GB/s measured with pinned memory, profiles/r02/hostpath.json); achieved = file bytes / in-process call"}})

    elif a.workload == "e2e23":
        ix, g, keys, counts, pf = build_index23(a.genome, rank, world, dev, cache)
        del keys, counts
        r = measure_e2e23(ix, g, dev, cache, a.e2e_gb)
        out.update({"metric": "reads_per_sec_count23_fixed_file_end_to_end", "value": r["value"], "unit": "reads/s", "ms_per_step": r["seconds"] * 1e3, "steps": 1, "warmup": 0,
                    "dtype": "u64", "config": {"workload": f"aix_count23_fixed_file on a {r['file_bytes'] / 1e9:.2f} GB PLAIN reads file (150 bp, seed 41) in {r['scratch']}", **r},
                    "roofline": {"bound": "hbm", "achieved": r["GBps"], "peak": 64.0, "unit": "GB/s", "frac": r["GBps"] / 64.0, "traffic": None,
                                 "kernel": "H2D copies of the parts overlapped with k_probe23_slots + histogram",
                                 "note": "peak = PCIe 5 x16 (64 GB/s raw); the resident kernel counts ~40 GB/s of reads, so link and kernel are of one size here"}})

    elif a.workload == "lookup13":
        from aindex_amd.engine import Index
        from aindex_amd.builder import all_13mers_pf_path as pf13_path
        ix = Index.open_13(pf13_path(), None, dev)
        g = engine.synth_genome_t(13, 4_000_000, dev)
        reads = engine.synth_reads_t(14, g, 1_000_000, 150, n_rate_ppm=1000)
        tf = ix.count13_t(reads)
        torch.cuda.synchronize()
        ix.set_tf_13(tf.cpu().numpy().view(np.uint64))
        q = engine.synth_kmers_t(15, a.queries, 13, dev, first=rank * a.queries)
        res = torch.empty(a.queries, dtype=torch.int32, device=f"cuda:{dev}")
        step = lambda: ix.tf_ascii_t(q, res)
        wall, kern_ms, _ = timed_steps(step, a.steps, a.warmup, dev)
        bpq = 13.0 + 8.0 + 4.0     # what the kernel requests: the query in, ONE 8-byte read of the code-ordered table, the answer out
        achieved = bpq * a.queries / (kern_ms * 1e-3) / 1e9
        out.update({"metric": "kmer_lookups_per_sec_13mer_batch", "value": world * a.queries * a.steps / wall, "unit": "lookups/s",
                    "ms_per_step": wall / a.steps * 1e3, "dtype": "u64",
                    "config": {"workload": "13-mer dense table batch lookup, uniform-random 13-mers", "queries_per_step_per_gpu": a.queries},
                    "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "traffic": None, "kernel": "k_lookup13_ascii", "kernel_ms": kern_ms, "requested_bytes_per_query": bpq,
                                 "line_traffic": {"GBps": (128.0 + 17.0) * a.queries / (kern_ms * 1e-3) / 1e9,
                                                  "note": "one 128-byte line per table read + streamed bytes (estimate)"},
                                 "reference_algorithm": {"bytes_per_query": 117.0, "GBps": 117.0 * a.queries / (kern_ms * 1e-3) / 1e9,
                                                         "note": "SURVEY 8(d): one MPHF evaluation (92 B) + the 8-byte tf + 17 B streamed; not what this kernel moves"}}})

    elif a.workload == "count13":
        from aindex_amd.engine import Index
        from aindex_amd.builder import all_13mers_pf_path as pf13_path
        ix = Index.open_13(pf13_path(), None, dev)
        g = engine.synth_genome_t(13, 4_000_000, dev)
        reads = engine.synth_reads_t(14, g, a.reads, 150, n_rate_ppm=1000, first_read=rank * a.reads)
        tf = torch.empty(4 ** 13, dtype=torch.int64, device=f"cuda:{dev}")
        def step():
            ix.count13_t(reads, tf)
            adist.all_reduce_sum_(tf)
        wall, kern_ms, _ = timed_steps(step, a.steps, a.warmup, dev)
        windows = a.reads * (150 - 12)
        # requested by the three kernels: input once, a 2-byte payload per window written by the split and read by the histogram,
        # the 4^13 u64 table cleared and written through the permutation (4 B per slot read for the permutation)
        requested = a.reads * 151 + windows * 4.0 + (4 ** 13) * (8.0 + 8.0 + 4.0)
        achieved = requested / (kern_ms * 1e-3) / 1e9
        cb = None
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline_count13(ix, reads, min(a.cpu_reads, a.reads), os.path.join(cache, "cpu13"), pf13_path())
        out.update({"metric": "reads_per_sec_13mer_count", "value": world * a.reads * a.steps / wall, "unit": "reads/s",
                    "ms_per_step": wall / a.steps * 1e3, "dtype": "u64",
                    "config": {"workload": "configs[1]: 13-mer dense 4^13 table count of 150 bp reads + all-reduce", "reads_per_step_per_gpu": a.reads},
                    **({"cpu_baseline": cb.get("reference", cb["port_mt"]), "cpu_baseline_extra": cb} if cb else {}),
                    "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "traffic": None, "kernel": "k_c13_split_chunked + directory sort + k_c13_hist_chunked (+ memset of the table)", "kernel_ms": kern_ms,
                                 "requested_bytes_per_launch": requested, "reads_per_launch": a.reads,
                                 "note": "not HBM-bound: the split kernel is bound by its LDS atomics (DESIGN.md §5); the fraction says how far from the stream rate it is",
                                 "reference_algorithm": {"bytes_per_window": 1.09 + 8.0, "GBps": (a.reads * 151 + windows * 8.0) / (kern_ms * 1e-3) / 1e9,
                                                         "note": "SURVEY 8(d): input + one 8-byte counter RMW per window"}}})
        tr = load_pmc_traffic("count13", reads_per_launch=a.reads)
        if tr:
            out["roofline"]["traffic"] = tr.get("bytes_per_launch")
            out["roofline"]["traffic_source"] = tr.get("source")

    elif a.workload == "count23" and a.scaling == "strong":
        ix, g, keys, counts, pf = build_index23(a.genome, rank, world, dev, cache)
        del keys, counts
        apply_ab_switches(ix, a)
        r = measure_count23_strong(ix, g, rank, world, dev, a.total_reads, a.steps, a.warmup)
        out.update({"metric": r["metric"], "value": r["value"], "unit": r["unit"], "scaling": "strong", "ms_per_step": r["ms_per_step"], "dtype": "u64",
                    "config": {"workload": f"configs[3]: 23-mer counting, {a.total_reads} synthetic 150 bp reads (seed 41) in {world} contiguous read ranges, "
                                           "histogram against the fixed MPHF index + all-reduce(sum) of tf[]",
                               "total_reads": a.total_reads, "reads_this_rank": r["reads_this_rank"], "index_keys": ix.n, "genome_bp": a.genome,
                               "parallelism": f"reads sharded x{world}, index replicated", "backend": r["backend"], "collective": r["collective"],
                               "counting_backend": r["counting_backend"], "histogram_passes": r["histogram_passes"],
                               "collective_ranks": r["collective_ranks"], "one_device_rehearsal": r["one_device_rehearsal"]},
                    **({"same_workload_on_one_gpu": r["same_workload_on_one_gpu"]} if "same_workload_on_one_gpu" in r else {}),
                    "allreduce_ms": r["allreduce_ms"], "allreduce_bytes": r["allreduce_bytes"], "tf_digest": r["tf_digest"],
                    "windows_counted_all_ranks": r["windows_counted_all_ranks"],
                    "roofline": {**count23_roofline(ix, r["windows_this_rank"], r["reads_this_rank"], r["kernel_ms_this_rank"]),
                                 "calls_in_process": a.steps + a.warmup + 2}})
        tr = load_pmc_traffic("count23", reads_per_launch=r["reads_this_rank"])
        if tr:
            out["roofline"]["traffic"] = tr.get("bytes_per_launch")
            out["roofline"]["traffic_source"] = tr.get("source")
        if rank == 0 and not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_count23(ix, g, pf, dev, cache)

    elif a.workload == "count23":
        ix, g, keys, counts, pf = build_index23(a.genome, rank, world, dev, cache)
        apply_ab_switches(ix, a)
        reads = engine.synth_reads_t(41, g, a.reads, 150, rc_half=True, n_rate_ppm=1000, first_read=rank * a.reads)
        tf = torch.zeros(ix.n, dtype=torch.int32, device=f"cuda:{dev}")
        step = lambda: adist.count23_sharded_t(ix, reads, _lib.CANON_TRUE_RC, tf)
        wall, _, _ = timed_steps(step, a.steps, a.warmup, dev)
        digest = tf_checksum(tf)
        _, kern_ms, _ = timed_steps(lambda: ix.count23_fixed_t(reads, _lib.CANON_TRUE_RC, tf), 3, 0, dev)     # the kernel alone
        windows = a.reads * (150 - 22)
        backend_ran = ix.info["count23_backend"]                        # of the timed calls (the CPU-baseline leg below counts a short sample through another one)
        roof23 = count23_roofline(ix, windows, a.reads, kern_ms)
        cb23 = cpu_baseline_count23(ix, g, pf, dev, cache) if (rank == 0 and world == 1 and not a.no_cpu_baseline) else None
        out.update({**({"cpu_baseline": cb23} if cb23 else {}),
                    "metric": "reads_per_sec_23mer_count_fixed_mphf", "value": world * a.reads * a.steps / wall, "unit": "reads/s",
                    "ms_per_step": wall / a.steps * 1e3, "dtype": "u64",
                    "config": {"workload": "configs[3]-shaped: 23-mer histogram against a fixed MPHF, 150 bp reads, + all-reduce(sum) of tf[]; per-rank reads fixed (weak)",
                               "reads_per_step_per_gpu": a.reads, "index_keys": ix.n,
                               "counting_backend": {1: "memory-side atomics", 2: "slot stream + LDS histogram",
                                                    3: "distinct k-mers first (K1), one probe per distinct k-mer"}.get(backend_ran, "none")},
                    "tf_digest": digest,
                    "roofline": {**roof23,
                                 "calls_in_process": a.steps + a.warmup + 3 + (1 if cb23 else 0)}})      # for the per-call normalisation of the PMC passes
        tr = load_pmc_traffic("count23", reads_per_launch=a.reads) if backend_ran == 2 else None      # the PMC passes profile the probe path (--probe-path)
        if tr:
            out["roofline"]["traffic"] = tr.get("bytes_per_launch")
            out["roofline"]["traffic_source"] = tr.get("source")
        if not a.no_gather_probe and backend_ran != 3:                   # (back end 3 reads no table line per window)
            peak_acc = gather_roofline(dev)
            acc = out["roofline"]["lines_per_window"]
            ach = windows * acc / (kern_ms * 1e-3)
            out["roofline"]["random_read"] = {"peak_accesses_per_s": peak_acc, "achieved_accesses_per_s": ach, "frac": ach / peak_acc, "accesses_per_window": acc,
                                              "note": "peak = k_gather over a 4 GiB table; every window also issues one scattered atomic"}

    elif a.workload == "coverage23":
        ix, g, keys, counts, pf = build_index23(a.genome, rank, world, dev, cache)
        apply_ab_switches(ix, a)
        L = a.seq_len
        seqs = engine.synth_reads_t(51, g, a.seqs, L, rc_half=True, n_rate_ppm=1000, first_read=rank * a.seqs)   # records of L bases + '\n'
        offs = torch.arange(0, (a.seqs + 1) * (L + 1), L + 1, dtype=torch.int64, device=f"cuda:{dev}")
        per = (L + 1) - 23 + 1
        ooffs = torch.arange(0, (a.seqs + 1) * per, per, dtype=torch.int64, device=f"cuda:{dev}")
        outp = torch.zeros(a.seqs * per, dtype=torch.int32, device=f"cuda:{dev}")
        step = lambda: ix.coverage_t(seqs, offs, ooffs, a.seqs * per, 0, outp)
        wall, kern_ms, _ = timed_steps(step, a.steps, a.warmup, dev)
        positions = a.seqs * per
        nzf = int(torch.count_nonzero(outp).item()) / outp.numel()
        if nzf < 0.97:                                              # windows without N are keys: (1 - 0.001)^23 = 0.977; a line over a wrong profile is not a measurement
            raise SystemExit(f"coverage23: only {nzf:.4f} of the positions came back non-zero (expected ~0.98)")
        pp = ix.probe_profile()
        achieved = positions * (1.0 + pp["bytes_per_hit_probe"] + 4.0) / (kern_ms * 1e-3) / 1e9     # 1 B of sequence + one probe + the 4-byte answer per position
        cb = None
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline_coverage23(ix, pf, seqs, L, outp, per, 40, os.path.join(cache, "cpucov"))
        out.update({"metric": "sequences_per_sec_coverage_23mer", "value": world * a.seqs * a.steps / wall, "unit": "sequences/s",
                    "ms_per_step": wall / a.steps * 1e3, "dtype": "u64",
                    "config": {"workload": f"configs[4]: per-position tf profile (k=23) of {L} bp sequences drawn from the indexed genome (50 % rc, 0.1 % N)",
                               "sequences_per_step_per_gpu": a.seqs, "positions_per_step": positions, "nonzero_fraction": nzf},
                    **({"cpu_baseline": cb} if cb else {}),
                    "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "traffic": None, "kernel": "k_coverage", "kernel_ms": kern_ms, "positions_per_sec": positions / (kern_ms * 1e-3),
                                 "positions_per_launch": positions, "probe": pp["name"], "requested_bytes_per_position": 5.0 + pp["bytes_per_hit_probe"],
                                 "reference_algorithm": {"bytes_per_position": 1.0 + 154.0 + 4.0, "GBps": positions * 159.0 / (kern_ms * 1e-3) / 1e9,
                                                         "note": "SURVEY 8(d) lookup bytes per window of a present k-mer; not what this kernel moves"}}})
        tr = load_pmc_traffic("coverage23", positions_per_launch=positions)
        if tr:
            out["roofline"]["traffic"] = tr.get("bytes_per_launch")
            out["roofline"]["traffic_source"] = tr.get("source")
        if not a.no_gather_probe:
            peak_acc = gather_roofline(dev)
            nzf = out["config"]["nonzero_fraction"]
            acc = pp["lines_per_hit_probe"]                          # one bucket line (or 3 MPHF records + the key record) per position
            ach = positions * acc / (kern_ms * 1e-3)
            out["roofline"]["random_read"] = {"peak_accesses_per_s": peak_acc, "achieved_accesses_per_s": ach, "frac": ach / peak_acc, "accesses_per_position": acc,
                                              "note": "peak = k_gather over a 4 GiB table"}

    elif a.workload == "coverage13":
        from aindex_amd.engine import Index
        from aindex_amd.builder import all_13mers_pf_path as pf13_path
        ix = Index.open_13(pf13_path(), None, dev)
        g = engine.synth_genome_t(13, 4_000_000, dev)
        reads = engine.synth_reads_t(14, g, 1_000_000, 150, n_rate_ppm=1000)
        ix.set_tf_13(ix.count13_t(reads).cpu().numpy().view(np.uint64))        # the 13-mer index of config 2 (counts of 1 M reads)
        L = a.seq_len
        seqs = engine.synth_reads_t(51, g, a.seqs, L, rc_half=False, n_rate_ppm=1000, first_read=rank * a.seqs)
        offs = torch.arange(0, (a.seqs + 1) * (L + 1), L + 1, dtype=torch.int64, device=f"cuda:{dev}")
        per = (L + 1) - 13 + 1
        ooffs = torch.arange(0, (a.seqs + 1) * per, per, dtype=torch.int64, device=f"cuda:{dev}")
        outp = torch.zeros(a.seqs * per, dtype=torch.int32, device=f"cuda:{dev}")
        step = lambda: ix.coverage_t(seqs, offs, ooffs, a.seqs * per, 0, outp)
        wall, kern_ms, _ = timed_steps(step, a.steps, a.warmup, dev)
        positions = a.seqs * per
        nzf13 = int(torch.count_nonzero(outp).item()) / outp.numel()
        if nzf13 < 0.97:                                            # forward reads of the counted genome: every window without N has a count
            raise SystemExit(f"coverage13: only {nzf13:.4f} of the positions came back non-zero (expected ~0.99)")
        achieved = positions * (1.0 + 8.0 + 4.0) / (kern_ms * 1e-3) / 1e9      # one 8-byte read of the code-ordered table per position
        out.update({"metric": "sequences_per_sec_coverage_13mer", "value": world * a.seqs * a.steps / wall, "unit": "sequences/s",
                    "ms_per_step": wall / a.steps * 1e3, "dtype": "u64",
                    "config": {"workload": f"configs[4]: per-position tf profile (k=13) of {L} bp sequences drawn from the counted genome (0.1 % N)",
                               "sequences_per_step_per_gpu": a.seqs, "positions_per_step": positions, "nonzero_fraction": nzf13},
                    "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "traffic": None, "kernel": "k_coverage", "kernel_ms": kern_ms, "positions_per_sec": positions / (kern_ms * 1e-3)}})

    elif a.workload == "distinct23":
        from aindex_amd import counting
        g = engine.synth_genome_t(23, a.genome, dev)
        reads = engine.synth_reads_t(41, g, a.reads, 150, rc_half=True, n_rate_ppm=1000, first_read=rank * a.reads)
        res = {}
        step = lambda: res.__setitem__("o", counting.count_distinct_t(reads, 23, _lib.CANON_REF_X86, 1))
        wall, kern_ms, _ = timed_steps(step, a.steps, a.warmup, dev)
        keys_t, counts_t = res["o"]
        windows = a.reads * (150 - 22)
        counted, clean = int(counts_t.to(torch.int64).sum().item()), clean_windows(reads, a.reads, 150, 23)
        if counted != clean:                                        # every window without N is counted exactly once, whatever the number of pieces merged
            raise SystemExit(f"distinct23: sum(counts) = {counted}, windows without N = {clean}")
        msd = os.environ.get("AIX_K1_ROCPRIM") is None
        positions_n = a.reads * 151 - 22
        distinct_n = int(keys_t.numel())
        # requested by the kernels: the radix path writes the 8-byte codes once and moves them through 6 read+write passes; the MSD path
        # (aix_k1.hip) writes the codes (8), level 1 reads and writes them (16), level 2 reads them twice and writes 4-byte remainders
        # (20), the per-bucket stage reads those (4) and 12 bytes per distinct key go out twice
        per_pos = 8.0 * (1 + 2 * 6) if not msd else (8.0 + 16.0 + 20.0 + 4.0)
        requested = a.reads * 151 + positions_n * per_pos + (24.0 * distinct_n if msd else 12.0 * distinct_n)
        achieved = requested / (kern_ms * 1e-3) / 1e9
        cb = None
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            ns = min(a.cpu_reads, a.reads)
            sk, sc = counting.count_distinct_t(reads[: ns * 151], 23, _lib.CANON_REF_X86, 1)
            cb = cpu_baseline_distinct23(reads, ns, sk.cpu().numpy().view(np.uint64), sc.cpu().numpy(), os.path.join(cache, "cpukc"))
        out.update({"metric": "reads_per_sec_23mer_distinct_count", "value": world * a.reads * a.steps / wall, "unit": "reads/s",
                    "ms_per_step": wall / a.steps * 1e3, "dtype": "u64",
                    "config": {"workload": "K1 (kmer_counter): distinct canonical 23-mers of 150 bp reads with their counts, reads resident in HBM, per-rank sets (no exchange)",
                               "reads_per_step_per_gpu": a.reads, "windows": windows, "distinct_kmers": int(keys_t.numel()), "sum_counts": counted,
                               "sum_counts_check": "equals the windows without N counted independently with torch"},
                    **({"cpu_baseline": cb.get("reference", cb["port_1t"]), "cpu_baseline_extra": cb} if cb else {}),
                    "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                 "kernel": ("k_window_codes + k_k1_split / k_k1_count / k_k1_scatter / k_k1_final / k_k1_gather (aix_count_distinct_dev)" if msd
                                            else "k_window_codes + rocPRIM radix sort / run-length (aix_count_distinct_dev, AIX_K1_ROCPRIM=1)"),
                                 "kernel_ms": kern_ms, "reads_per_launch": a.reads, "requested_bytes_per_launch": requested,
                                 "calls_in_process": a.steps + a.warmup + (1 if cb else 0),
                                 "note": "not HBM-bound: one returning LDS atomic per window and level (DESIGN.md §4)"}})
        tr = load_pmc_traffic("distinct23", reads_per_launch=a.reads)
        if tr:
            out["roofline"]["traffic"] = tr.get("bytes_per_launch")
            out["roofline"]["traffic_source"] = tr.get("source")

    elif a.workload == "positions23":
        from aindex_amd._lib import lib, check, vp
        ix, g, keys, counts, pf = build_index23(a.genome, rank, world, dev, cache)
        reads_t = engine.synth_reads_t(41, g, a.reads, 150, rc_half=True, n_rate_ppm=1000)
        if a.positions_tf == "reads":
            # the reference pipeline (kmer_counter -> compute_index -> compute_aindex) hands compute_aindex the tf of the READS, so every
            # window of a stored k-mer is placed (positions array = 8 B per found window); with the genome's multiplicities (tf = 1 here)
            # only the first occurrence of every k-mer would be written
            from aindex_amd.engine import Index
            tf_reads = torch.zeros(ix.n, dtype=torch.int32, device=f"cuda:{dev}")
            ix.count23_fixed_t(reads_t, _lib.CANON_TRUE_RC, tf_reads)
            torch.cuda.synchronize()
            checker = ix.checker_array()
            ix.close()
            del keys, counts
            ix = Index.create_23(pf, checker, tf_reads.cpu().numpy().view(np.uint32), dev)
            del tf_reads, checker
        apply_ab_switches(ix, a)
        res = {}
        step = lambda: res.__setitem__("o", ix.positions_fill_t(reads_t))
        wall, kern_ms, _ = timed_steps(step, a.steps, a.warmup, dev)
        indices_t, pos_t = res["o"]
        # host-buffer twin of the same call (PCIe inclusive: reads in, .indices.bin / .index.bin images out), checked equal
        host_bytes = reads_t.cpu().numpy().tobytes()
        t0 = time.perf_counter()
        indices, pos = ix.positions_fill(host_bytes)
        host_dt = time.perf_counter() - t0
        assert np.array_equal(indices, indices_t.cpu().numpy().view(np.uint64)) and np.array_equal(pos, pos_t.cpu().numpy().view(np.uint64))
        cbp = None
        exe = os.path.join(ROOT, "oracle", "_ref", "compute_aindex")
        if rank == 0 and world == 1 and not a.no_cpu_baseline and os.path.exists(exe):
            # the reference's own tool (one thread: its multi-thread slot order is schedule dependent) on a sample of the same
            # reads against the same index files; its .index.bin must equal ours byte for byte. Wall clock of the whole run
            # (it loads the index first: ~1 s of the total).
            import subprocess
            tmpd = os.path.join(cache, "cpupos")
            os.makedirs(tmpd, exist_ok=True)
            prefix = os.path.join(tmpd, "p23")
            open(prefix + ".pf", "wb").write(pf)
            ix.tf_array().tofile(prefix + ".tf.bin")
            ix.checker_array().tofile(prefix + ".kmers.bin")
            ns = min(a.reads, 100_000)
            sample = host_bytes[: ns * 151]
            open(prefix + ".reads", "wb").write(sample)
            try:
                t0 = time.perf_counter()
                subprocess.run([exe, prefix + ".reads", prefix + ".pf", prefix, "1", "23", prefix + ".tf.bin", prefix + ".kmers.bin", prefix + ".none.txt"],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, check=True)
                dt_ref = time.perf_counter() - t0
                ours_ind, ours_pos = ix.positions_fill(sample)
                assert np.array_equal(np.fromfile(prefix + ".index.bin", dtype=np.uint64), ours_pos), "reference compute_aindex and GPU disagree on the sample"
                assert np.array_equal(np.fromfile(prefix + ".indices.bin", dtype=np.uint64), ours_ind)
                cbp = {"value": ns / dt_ref, "unit": "reads/s", "cores": 1, "kind": "reference",
                       "sample": f"first {ns} reads of the batch through oracle/_ref/compute_aindex, 1 thread (wall clock incl. its index load and file output)"}
            except Exception as e:
                log(f"cpu_baseline: reference compute_aindex not usable ({type(e).__name__}: {e})")
            for f in (".pf", ".tf.bin", ".kmers.bin", ".reads", ".index.bin", ".indices.bin"):
                try:
                    os.remove(prefix + f)
                except OSError:
                    pass
        windows = int(reads_t.numel() - 22)
        # bytes the fill asks for, per window: 1 in + 4 out (probe: read byte in, bucket out) + the probe's line; then the grouping of the
        # (bucket, offset) pairs: MSD path = 4 in + 8 out (level 1), 8 in (level-2 count), 8 in + 8 out (level-2 scatter), 8 in (final);
        # sort path = 4 radix passes x 16 B read + written; and 8 B per placed offset
        pp = ix.probe_profile()
        msd = os.environ.get("AIX_A2_MSD", "1") != "0" and windows >= (1 << 22)
        grouping_bytes = 44.0 if msd else 4 * 16.0
        achieved = (windows * (1.0 + 4.0 + pp["bytes_per_hit_probe"] + grouping_bytes) + 8.0 * int(indices[-1])) / (kern_ms * 1e-3) / 1e9
        out.update({**({"cpu_baseline": cbp} if cbp else {}),
                    "metric": "reads_per_sec_positions_fill_23mer", "value": world * a.reads * a.steps / wall, "unit": "reads/s",
                    "ms_per_step": wall / a.steps * 1e3, "dtype": "u64",
                    "config": {"workload": "A1+A2: positions index of 150 bp reads (resident in HBM) against the fixed 23-mer index; tf = occurrences in the "
                                           + ("reads (every found window placed)" if a.positions_tf == "reads" else "genome (first occurrences only)"),
                               "positions_tf": a.positions_tf,
                               "reads": a.reads, "windows": windows, "positions_total": int(indices[-1]), "filled": int((pos != 0).sum()),
                               "host_buffer_call_ms": host_dt * 1e3, "host_buffer_reads_per_s": a.reads / host_dt},
                    "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                 "kernel": ("k_a2_probe + k_k1_split<A2Keys> + k_k1_count + k_k1_scatter<u64> + k_a2_final (two-level MSD partition + per-bucket LDS stage)"
                                            if msd else "k_a2_probe + rocprim radix_sort_pairs + k_a2_first/k_a2_place"),
                                 "grouping": "msd" if msd else "radix sort", "requested_bytes_per_window": 1.0 + 4.0 + pp["bytes_per_hit_probe"] + grouping_bytes,
                                 "kernel_ms": kern_ms, "reads_per_launch": a.reads, "windows_per_launch": windows, "probe": pp["name"],
                                 "calls_in_process": a.steps + a.warmup + 1 + (1 if cbp else 0)}})
        tr = load_pmc_traffic("positions23", reads_per_launch=a.reads)
        if tr:
            out["roofline"]["traffic"] = tr.get("bytes_per_launch")
            out["roofline"]["traffic_source"] = tr.get("source")

    elif a.workload == "normalize":
        from aindex_amd import counting
        g = engine.synth_genome_t(23, 4_000_000, dev)
        reads = engine.synth_reads_t(41, g, a.reads, 150, n_rate_ppm=1000).cpu().numpy().reshape(-1, 151)
        fq = np.empty((a.reads, 4 + 151 + 2 + 151), dtype=np.uint8)          # "@r0\n" + seq\n + "+\n" + qual\n
        fq[:, :4] = np.frombuffer(b"@r0\n", dtype=np.uint8)
        fq[:, 4:155] = reads
        fq[:, 155:157] = np.frombuffer(b"+\n", dtype=np.uint8)
        fq[:, 157:307] = ord("I")
        fq[:, 307] = ord("\n")
        raw = torch.from_numpy(fq.reshape(-1)).to(f"cuda:{dev}")
        res = {}
        step = lambda: res.__setitem__("o", counting.normalize_t(raw, _lib.FMT_FASTQ, 0))
        wall, kern_ms, _ = timed_steps(step, a.steps, a.warmup, dev)
        assert res["o"].numel() == a.reads * 151
        nbytes = raw.numel()
        out.update({"metric": "bytes_per_sec_fastq_normalise", "value": nbytes * a.steps / wall, "unit": "B/s", "ms_per_step": wall / a.steps * 1e3, "dtype": "u8",
                    "config": {"workload": "FASTQ -> PLAIN normalisation on the device", "reads": a.reads, "raw_bytes": nbytes},
                    "roofline": {"bound": "hbm", "achieved": (2 * nbytes + a.reads * 151) / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (2 * nbytes + a.reads * 151) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "kernel": "k_norm_summarise + 2 rocprim scans + k_norm_emit", "kernel_ms": kern_ms}})

    elif a.workload == "gather":
        from aindex_amd._lib import lib, check, vp
        nel = a.table_mib * (1 << 20) // a.elem
        table = torch.empty(nel * a.elem // 8, dtype=torch.int64, device=f"cuda:{dev}")
        table.random_(0, 1 << 40)
        sink = torch.zeros(8, dtype=torch.int64, device=f"cuda:{dev}")
        sp = vp(torch.cuda.current_stream().cuda_stream)
        step = lambda: check(lib().aix_bench_gather_dev(vp(table.data_ptr()), nel, a.elem, a.unroll, a.queries, 99, vp(sink.data_ptr()), sp))
        wall, kern_ms, _ = timed_steps(step, a.steps, a.warmup, dev)
        rate = a.queries / (kern_ms * 1e-3)
        out.update({"metric": "random_gather_accesses_per_sec", "value": rate, "unit": "accesses/s", "ms_per_step": wall / a.steps * 1e3, "dtype": f"u{a.elem * 8}",
                    "config": {"workload": f"uniform-random {a.elem}-byte gather over {a.table_mib} MiB, {a.unroll} in flight per lane", "accesses": a.queries},
                    "roofline": {"bound": "hbm", "achieved": rate * 64 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s (64 B per access)", "frac": rate * 64 / 1e9 / HBM_PEAK_GBS,
                                 "traffic": None, "kernel": "k_gather", "kernel_ms": kern_ms}})

    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or os.environ.get("AIX_FORCE_DIST"):
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
