#!/bin/bash
# same-box A/B: A2 final stage at 5 (variant_A) and 6 (variant_B: launch bound) workgroups per CU, bucket size target, count23 with one lane per line
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab14; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  ms_per_step %.3f kernel_ms %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"]))
PY
}
for rep in 1 2; do
for v in A B; do
cp aindex_amd/lib/variant_$v.so aindex_amd/lib/libaindex_hip.so
run ${v}_pos_$rep --workload positions23 --reads 5000000 $B || exit 5
done
done
cp aindex_amd/lib/variant_B.so aindex_amd/lib/libaindex_hip.so
for tg in 512 2048 3000; do AIX_A2_TARGET=$tg run B_pos_t$tg --workload positions23 --reads 5000000 $B || exit 5; done
for rep in 1 2; do
run c23_def_$rep --workload count23 --reads 10000000 $B || exit 5
run c23_l1_$rep --workload count23 --reads 10000000 --bucket-lanes 1 $B || exit 5
done
