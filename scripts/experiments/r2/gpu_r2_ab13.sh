#!/bin/bash
# count23's slot probe: two lanes per bucket line (new default) against eight, alternating on one box; sharded entry points once more
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab13; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "count23 or sharded_entry_points or bucket_table" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 3
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  ms_per_step %.3f kernel_ms %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"]))
PY
}
for rep in 1 2 3; do
run c23_def_$rep --workload count23 --reads 10000000 $B || exit 5
run c23_l8_$rep --workload count23 --reads 10000000 --bucket-lanes 8 $B || exit 5
run c23_l4_$rep --workload count23 --reads 10000000 --bucket-lanes 4 $B || exit 5
done
run strong_def --workload count23 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline || exit 5
run strong_l8 --workload count23 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline --bucket-lanes 8 || exit 5
