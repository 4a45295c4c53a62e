#!/bin/bash
# count23: piece size of the probe / histogram pipeline (2^27, 2^28 default, 2^29 windows), same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab23; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json"))
print("%-28s %10.4g %s  ms_per_step %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"]))
PY
}
for rep in 1 2; do
for p in 536870912 1073741824 2147483648; do
AIX_COUNT23_PIECE=$p run c23_p${p}_$rep --workload count23 --reads 10000000 $B || exit 5
done; done
for p in 536870912 1073741824 2147483648; do
AIX_COUNT23_PIECE=$p run strong_p$p --workload count23 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline || exit 5
done
