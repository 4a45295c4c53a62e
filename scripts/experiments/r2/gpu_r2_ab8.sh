#!/bin/bash
# lane width of the bucket read on the streaming consumers
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2ab8; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"], r["frac"]))
PY
}
for l in 8 4 2 1; do
run count23_l$l --workload count23 --reads 10000000 --bucket-lanes $l $B || exit 5
run cov_l$l --workload coverage23 --seqs 100000 --bucket-lanes $l $B || exit 5
done
run qmix_l8 --workload lookup23 --query-mix $B || exit 5
