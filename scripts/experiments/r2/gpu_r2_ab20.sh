#!/bin/bash
# same-box A/B: non-temporal stores for the probe kernels' output streams (variant_B) against plain stores (variant_A)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab20; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  ms_per_step %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"]))
PY
}
for rep in 1 2; do for v in A B; do
cp aindex_amd/lib/variant_$v.so aindex_amd/lib/libaindex_hip.so
run ${v}_c23_$rep --workload count23 --reads 10000000 $B || exit 5
run ${v}_pos_$rep --workload positions23 --reads 5000000 $B || exit 5
done; done
