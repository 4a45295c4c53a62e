#!/bin/bash
# split kernel of count13 / count23: number of (persistent) workgroups, same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab25; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json"))
print("%-28s %10.4g %s  ms_per_step %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"]))
PY
}
for rep in 1 2; do for g in 256 512 768 1024; do
AIX_C13_GRID=$g run c13_g${g}_$rep --workload count13 $B || exit 5
AIX_C13_GRID=$g run c23_g${g}_$rep --workload count23 --reads 10000000 $B || exit 5
done; done
