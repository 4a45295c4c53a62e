#!/bin/bash
# count23 at 200 M reads: piece size, two rounds on one box
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab24; mkdir -p $O; cd $R
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json"))
print("%-28s %10.4g %s  ms_per_step %.3f digest %s" % ("$n", d["value"], d["unit"], d["ms_per_step"], d["tf_digest"]["weighted"]))
PY
}
for rep in 1 2; do
for p in 268435456 536870912 1073741824 2147483648; do
AIX_COUNT23_PIECE=$p run strong_p${p}_$rep --workload count23 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline || exit 5
done; done
