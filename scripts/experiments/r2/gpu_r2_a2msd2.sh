#!/bin/bash
# A2 final stage: bucket size sweep + kernel stats
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/a2msd2; mkdir -p $O; cd $R
AIX_A2_MSD=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "positions_fill" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 3
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  ms_per_step %.3f kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"], r["frac"]))
PY
}
for tg in 1024 2048; do AIX_A2_TARGET=$tg run pos23_t$tg --workload positions23 --reads 5000000 $B || exit 5; done
export TMPDIR=/tmp; cd /tmp
for tg in 1024; do
AIX_A2_TARGET=$tg timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$tg -- python3 $R/bench.py --workload positions23 --reads 5000000 $B > $O/stats$tg.json 2> $O/stats$tg.err || { tail -5 $O/stats$tg.err; exit 8; }
f=$(ls $O/stats$tg/*/*kernel_stats.csv | head -1); grep -E "k_a2|k_k1_(split|count|scatter)" $f | cut -d, -f1-4 | sed 's/(.*",/",/' 
done
