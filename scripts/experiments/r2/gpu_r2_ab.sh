#!/bin/bash
# Round-2 A/B pass: GPU tests first, then the verification table on/off and its lane widths on the kernels it serves.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2ab; mkdir -p $O; cd $R
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
step "pytest gpu"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 3
fi
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"], r["frac"]))
PY
}
step "count23 (10 M reads)"
for l in 8 4 2 1; do run count23_l$l --workload count23 --reads 10000000 --bucket-lanes $l $B || exit 4; done
run count23_off --workload count23 --reads 10000000 --no-bucket-table $B || exit 4
step "lookup23 Q_mix"
for l in 8 4 2 1; do run qmix_l$l --workload lookup23 --query-mix --bucket-lanes $l $B || exit 5; done
run qmix_off --workload lookup23 --query-mix --no-bucket-table $B || exit 5
step "lookup23 Q_rand"
for l in 8 2 1; do run qrand_l$l --workload lookup23 --bucket-lanes $l $B || exit 6; done
run qrand_off --workload lookup23 --no-bucket-table $B || exit 6
step "coverage23 / positions23"
run cov_l8 --workload coverage23 $B || exit 7
run cov_off --workload coverage23 --no-bucket-table $B || exit 7
run pos_l8 --workload positions23 --reads 5000000 $B || exit 7
run pos_off --workload positions23 --reads 5000000 --no-bucket-table $B || exit 7
step "done"
