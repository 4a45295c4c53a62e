#!/bin/bash
# count23: can the split of piece i share CUs with the probe of piece i+1? (small split shape + an LDS handle on the probe's occupancy), same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab26; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 4 --warmup 1 --workload count23 --reads 40000000"
run () { n=$1; shift; timeout -k 10 400 python bench.py $B > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json"))
print("%-28s %10.4g %s  ms_per_step %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"]))
PY
}
for rep in 1 2; do
run base_$rep || exit 5
AIX_C13_SHAPE=small run small_$rep || exit 5
AIX_C13_SHAPE=small AIX_PROBE_LDS=15360 run small_p15_$rep || exit 5
AIX_C13_SHAPE=small AIX_PROBE_LDS=18432 run small_p18_$rep || exit 5
AIX_C13_SHAPE=small AIX_PROBE_LDS=18432 AIX_C13_GRID=256 run small_p18_g256_$rep || exit 5
AIX_C13_SHAPE=small AIX_PROBE_LDS=22528 AIX_C13_GRID=256 run small_p22_g256_$rep || exit 5
AIX_PROBE_LDS=18432 run big_p18_$rep || exit 5
done
