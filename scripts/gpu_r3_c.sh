#!/bin/bash
# round 3: what limits the ingestion pipeline — host reader threads, part size, copy streams (one box, one 8 GB file in /dev/shm)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
F=/dev/shm/aix_r3c_reads.txt
python - <<PY
import sys; sys.path.insert(0, ".")
import bench, torch
from aindex_amd import engine
g = engine.synth_genome_t(13, 4_000_000, 0)
print("file bytes", bench.write_reads_file("$F", g, 14, 52_980_132, "plain", 0))
PY
for th in 4 8 12 16 24; do for mb in 64 256; do
  AIX_INGEST_THREADS=$th AIX_INGEST_PART_MB=$mb timeout -k 10 120 python scripts/gpu_r3_e2e_driver.py $F >> $O/sweep.jsonl 2>> $O/sweep.err || echo "failed th=$th mb=$mb"
done; done
for cs in 2; do for mb in 32 64 128; do
  AIX_INGEST_COPY_STREAMS=$cs AIX_INGEST_PART_MB=$mb timeout -k 10 120 python scripts/gpu_r3_e2e_driver.py $F >> $O/sweep.jsonl 2>> $O/sweep.err || echo "failed cs=$cs mb=$mb"
done; done
rm -f $F
python - <<PY
import json
for ln in open("$O/sweep.jsonl"):
    d = json.loads(ln)
    print({k: v for k, v in d["env"].items() if k != "AIX_NO_TORCH"}, "%.1f GB/s total %.3f read %.3f h2d %.3f wait %.3f compute %.3f out %.3f" % (d["GBps"], d["seconds_total"], d["seconds_read"], d["seconds_h2d"], d["seconds_wait"], d["seconds_compute"], d["seconds_output"]))
PY
