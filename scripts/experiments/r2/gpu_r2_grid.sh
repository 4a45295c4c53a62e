#!/bin/bash
# finer grids (256 workgroups per CU, >= 4 trips per wave) as the default: full GPU suite + smoke, then old (AIX_GRID_PER_CU=32) against new on one box
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O $R/gpurun_out/ab32; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -eq 0 ] || { tail -30 $O/pytest_gpu.log; exit 3; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || exit 4; tail -1 $O/smoke.log
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 8 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $R/gpurun_out/ab32/$n.json 2> $R/gpurun_out/ab32/$n.err || { echo "$n failed"; tail -8 $R/gpurun_out/ab32/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$R/gpurun_out/ab32/$n.json"))
print("%-20s %10.4g %s  ms_per_step %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"]))
PY
}
for g in 32 256 32 256; do
AIX_GRID_PER_CU=$g run qrand_g$g --workload lookup23 $B || exit 5
AIX_GRID_PER_CU=$g run qmix_g$g --workload lookup23 --query-mix $B || exit 5
AIX_GRID_PER_CU=$g run q10m_g$g --workload lookup23 --query-mix --queries 10000000 $B || exit 5
AIX_GRID_PER_CU=$g run pos_g$g --workload positions23 --reads 5000000 $B || exit 5
AIX_GRID_PER_CU=$g run l13_g$g --workload lookup13 $B || exit 5
done
