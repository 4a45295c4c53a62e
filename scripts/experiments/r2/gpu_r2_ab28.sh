#!/bin/bash
# Q_rand / Q_mix: absence-filter word read with a non-temporal load (variant_B.so) against the default (variant_A.so), same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab28; mkdir -p $O; cd $R
cp aindex_amd/lib/libaindex_hip.so $O/keep.so
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"]))
PY
}
for v in A B A B; do
cp aindex_amd/lib/variant_$v.so aindex_amd/lib/libaindex_hip.so
run ${v}_qrand --workload lookup23 $B || exit 5
run ${v}_qmix --workload lookup23 --query-mix $B || exit 5
done
cp $O/keep.so aindex_amd/lib/libaindex_hip.so; rm -f $O/keep.so
