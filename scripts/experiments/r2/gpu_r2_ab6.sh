#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2ab6; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "histogram or bucket" 2>&1 | tail -2
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"], r["frac"]))
PY
  grep -h "index:" $O/$n.err | tail -1
}
for l in 6 3; do AIX_MINIMIZER_TABLE=1 AIX_MINIMIZER_LOAD=$l run count23_mk_load$l --workload count23 --reads 10000000 $B || exit 5; done
export TMPDIR=/tmp; cd /tmp
AIX_MINIMIZER_TABLE=1 AIX_MINIMIZER_LOAD=2 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c23 -- python3 $R/bench.py --workload count23 --reads 10000000 --steps 5 --warmup 1 --no-cpu-baseline --no-gather-probe > $O/prof_c23.out 2> $O/prof_c23.err || exit 8
f=$(find $O/prof_c23 -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
for i, r in enumerate(csv.DictReader(open("$f"))):
    if i < 5: print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
