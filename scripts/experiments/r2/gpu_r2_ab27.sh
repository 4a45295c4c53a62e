#!/bin/bash
# count23: probe stream restricted to a share of the CUs (hipExtStreamCreateWithCUMask), partition + histogram on a stream of its own; same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab27; mkdir -p $O; cd $R
AIX_COUNT23_CUMASK=FFFFFF00 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "count23 or config4" > $O/pytest_mask.log 2>&1; rc=$?; tail -3 $O/pytest_mask.log; [ $rc -eq 0 ] || exit 3
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 4 --warmup 1 --workload count23 --reads 40000000"
run () { n=$1; shift; timeout -k 10 400 python bench.py $B > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json"))
print("%-28s %10.4g %s  ms_per_step %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"]))
PY
}
for rep in 1 2; do
run base_$rep || exit 5
AIX_COUNT23_CUMASK=FFFFFFFF run m100_$rep || exit 5
AIX_COUNT23_CUMASK=FFFFFFF0 run m88_$rep || exit 5
AIX_COUNT23_CUMASK=FFFFFF00 run m75_$rep || exit 5
AIX_COUNT23_CUMASK=FFFFF000 run m62_$rep || exit 5
AIX_COUNT23_CUMASK=FFFF0000 run m50_$rep || exit 5
done
