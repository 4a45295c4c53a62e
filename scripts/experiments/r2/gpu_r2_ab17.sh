#!/bin/bash
# K1: u64 counts written by the gather (no widening pass, no concatenation copy for a single piece): parity, then timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/stats; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "kmer_counter or distinct or tools or normalise or sharded_entry" > $O/pytest_k1.log 2>&1; rc=$?; tail -2 $O/pytest_k1.log
[ $rc -eq 0 ] || { tail -30 $O/pytest_k1.log; exit 3; }
Q="--no-cpu-baseline --no-secondary --no-gather-probe"
timeout -k 10 600 python bench.py --workload distinct23 --reads 5000000 --steps 5 --warmup 1 > $O/bench_distinct23reads5000000.json 2> $O/b4.err || { tail -5 $O/b4.err; exit 5; }
export TMPDIR=/tmp; cd /tmp
rm -rf $O/stats/dist23
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/dist23 -- python3 $R/bench.py --workload distinct23 --reads 5000000 --steps 5 --warmup 1 $Q > $O/stats/dist23.json 2> $O/stats/dist23.err || exit 8
f=$(ls -t $O/stats/dist23/*/*kernel_stats.csv | head -1); head -14 $f | cut -d, -f1-4 | sed 's/(.*",/",/'
python - <<PY
import json; d=json.load(open("$O/bench_distinct23reads5000000.json")); print("distinct23 ms_per_step", d["ms_per_step"], d["value"])
PY
