#!/bin/bash
# Round-2 A/B pass 3: adaptive absence filter, kernel trace of the count23 pipeline.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2ab3; mkdir -p $O; cd $R
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
step "targeted tests"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "bucket or early_exit or fuzz_queries or inconsistent or q23 or coverage or canonical or codes" > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 3
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"], r["frac"]))
PY
}
run qrand --workload lookup23 $B || exit 5
run qmix --workload lookup23 --query-mix $B || exit 5
run qmix_nofilter --workload lookup23 --query-mix --no-absence-filter $B || exit 5
run cov --workload coverage23 $B || exit 5
run count23 --workload count23 --reads 10000000 $B || exit 5
export TMPDIR=/tmp; cd /tmp
step "rocprofv3 kernel trace: count23"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_count23 -- python3 $R/bench.py --workload count23 --reads 10000000 --steps 5 --warmup 1 --no-cpu-baseline --no-gather-probe > $O/prof_count23.out 2> $O/prof_count23.err || exit 8
f=$(find $O/prof_count23 -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
step "done"
