#!/bin/bash
# Same-box A/B of two library builds (aindex_amd/lib/variant_A.so = Jenkins-keyed table, variant_B.so = table keyed by a mix of the
# code) at 8 / 4 / 2 lanes per bucket line: the benches the probe moves. Box-to-box spread is ~7 %, so only same-run numbers compare.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab12; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"]))
PY
}
for v in A B A; do
cp aindex_amd/lib/variant_$v.so aindex_amd/lib/libaindex_hip.so
for l in 8 4 2; do
run ${v}_count23_l$l --workload count23 --reads 10000000 --bucket-lanes $l $B || exit 5
run ${v}_qmix_l$l --workload lookup23 --query-mix --bucket-lanes $l $B || exit 5
run ${v}_cov_l$l --workload coverage23 --seqs 100000 --bucket-lanes $l $B || exit 5
run ${v}_pos_l$l --workload positions23 --reads 5000000 --bucket-lanes $l $B || exit 5
done
run ${v}_qrand --workload lookup23 $B || exit 5
done
