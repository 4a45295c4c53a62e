#!/bin/bash
# Round-2 A/B pass 7: table keyed by a mix of the code (no Jenkins on the table path): full GPU suite, then the benches it moves.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2ab7; mkdir -p $O; cd $R
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
step "pytest gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -4 $O/pytest_gpu.log
[ $rc -eq 0 ] || { grep -n "^E " $O/pytest_gpu.log | head -20; exit 3; }
fi
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"], r["frac"]))
PY
}
run qrand --workload lookup23 $B || exit 5
run qmix --workload lookup23 --query-mix $B || exit 5
run count23 --workload count23 --reads 10000000 $B || exit 5
for l in 4 2; do run count23_l$l --workload count23 --reads 10000000 --bucket-lanes $l $B || exit 5; done
run cov --workload coverage23 --seqs 100000 $B || exit 5
run pos --workload positions23 --reads 5000000 $B || exit 5
run dist --workload distinct23 --reads 5000000 $B || exit 5
step "done"
