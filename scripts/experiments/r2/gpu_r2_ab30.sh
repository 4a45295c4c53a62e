#!/bin/bash
# positions fill: lanes per bucket line of k_a2_probe (it was fixed at 8; count23's slot probe gained 5.6 % with 2), same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab30; mkdir -p $O; cd $R
AIX_A2_PROBE_LANES=2 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "positions_fill" > $O/pytest2.log 2>&1; rc=$?; tail -2 $O/pytest2.log; [ $rc -eq 0 ] || exit 3
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 6 --warmup 2 --workload positions23 --reads 5000000"
run () { n=$1; shift; timeout -k 10 400 python bench.py $B > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json"))
print("%-16s %10.4g %s  ms_per_step %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"]))
PY
}
for rep in 1 2 3; do for l in 8 4 2; do
AIX_A2_PROBE_LANES=$l run l${l}_$rep || exit 5
done; done
