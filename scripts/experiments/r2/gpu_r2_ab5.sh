#!/bin/bash
# Round-2 A/B pass 5: minimizer-keyed table on the streaming consumers.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2ab5; mkdir -p $O; cd $R
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
step "targeted tests"
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "bucket or histogram or count23_fixed or positions_fill_equals" > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log
[ $rc -eq 0 ] || { grep -n "^E " $O/pytest_gpu.log | head -20; exit 3; }
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"], r["frac"]))
PY
}
AIX_MINIMIZER_TABLE=1 run count23_mk --workload count23 --reads 10000000 $B || exit 5
run count23_nomk --workload count23 --reads 10000000 --no-minimizer-table $B || exit 5
AIX_MINIMIZER_TABLE=1 run cov_mk --workload coverage23 --seqs 100000 $B || exit 5
run cov_nomk --workload coverage23 --seqs 100000 --no-minimizer-table $B || exit 5
AIX_MINIMIZER_TABLE=1 run pos_mk --workload positions23 --reads 5000000 $B || exit 5
run pos_nomk --workload positions23 --reads 5000000 --no-minimizer-table $B || exit 5
AIX_MINIMIZER_TABLE=1 AIX_MINIMIZER_LOAD=4 run count23_mk_load4b --workload count23 --reads 10000000 $B || exit 5
AIX_MINIMIZER_TABLE=1 AIX_MINIMIZER_LOAD=8 run count23_mk_load8 --workload count23 --reads 10000000 $B || exit 5
grep -h "index:" $O/count23_mk.err | tail -1
export TMPDIR=/tmp; cd /tmp
step "rocprofv3 kernel trace: count23"
AIX_MINIMIZER_TABLE=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c23 -- python3 $R/bench.py --workload count23 --reads 10000000 --steps 5 --warmup 1 --no-cpu-baseline --no-gather-probe > $O/prof_c23.out 2> $O/prof_c23.err || exit 8
f=$(find $O/prof_c23 -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
for i, r in enumerate(csv.DictReader(open("$f"))):
    if i < 8: print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
step "done"
