#!/bin/bash
# positions23: does the scratch cache limit (16 GB default) cost allocations per call? same box, alternating
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab18; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  ms_per_step %.3f kernel_ms %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"]))
PY
}
for rep in 1 2; do
run pos_c16_$rep --workload positions23 --reads 5000000 $B || exit 5
AIX_SCRATCH_CACHE_GB=48 run pos_c48_$rep --workload positions23 --reads 5000000 $B || exit 5
AIX_POSITIONS_PIECE=402653184 run pos_piece_$rep --workload positions23 --reads 5000000 $B || exit 5
done
