#!/bin/bash
# grid cap of the grid-stride kernels (workgroups per CU), same box: lookups, count23, coverage
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab31; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 8 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json"))
print("%-20s %10.4g %s  ms_per_step %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"]))
PY
}
for g in 32 64 128 256 512 1024 64 32; do
AIX_GRID_PER_CU=$g run qrand_g$g --workload lookup23 $B || exit 5
AIX_GRID_PER_CU=$g run qmix_g$g --workload lookup23 --query-mix $B || exit 5
AIX_GRID_PER_CU=$g run c23_g$g --workload count23 --reads 10000000 $B || exit 5
AIX_GRID_PER_CU=$g run cov_g$g --workload coverage23 --seqs 100000 $B || exit 5
done
