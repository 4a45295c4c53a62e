#!/bin/bash
# Round-2 A/B pass 2: absence filter + histogram back end of count23, then the default bench line and the 2-rank rehearsal.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2ab2; mkdir -p $O; cd $R
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
step "pytest gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 3
fi
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"], r["frac"]))
PY
}
step "count23 (10 M reads): histogram back end vs atomics"
run count23_hist --workload count23 --reads 10000000 $B || exit 4
AIX_COUNT23_ATOMICS=1 run count23_atomics --workload count23 --reads 10000000 $B || exit 4
run count23_hist_nobk --workload count23 --reads 10000000 --no-bucket-table $B || exit 4
step "lookup23: absence filter"
run qrand_filter --workload lookup23 $B || exit 5
run qrand_nofilter --workload lookup23 --no-absence-filter $B || exit 5
run qmix_filter --workload lookup23 --query-mix $B || exit 5
run qmix_nofilter --workload lookup23 --query-mix --no-absence-filter $B || exit 5
AIX_BLOOM_BITS=12 run qrand_filter12 --workload lookup23 $B || exit 5
AIX_BLOOM_BITS=12 run qmix_filter12 --workload lookup23 --query-mix $B || exit 5
AIX_BLOOM_BITS=24 run qrand_filter24 --workload lookup23 $B || exit 5
step "coverage23"
run cov_filter --workload coverage23 $B || exit 7
run cov_nofilter --workload coverage23 --no-absence-filter $B || exit 7
step "default bench (N = 1, all secondaries)"
ts=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 8; }
echo "bench default wall $(( $(date +%s) - ts )) s" | tee -a $O/progress.txt; cat $O/bench_default.json
step "2-rank rehearsal on one device (gloo): bench.py --gpus 2 starts its own ranks"
timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --total-reads 40000000 > $O/bench_gpus2.json 2> $O/bench_gpus2.err || { tail -20 $O/bench_gpus2.err; exit 9; }
cat $O/bench_gpus2.json
step "done"
