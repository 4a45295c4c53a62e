#!/bin/bash
# 13-mer split kernel: parity tests, bench count13 / count23, kernel stats
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c13; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "count13 or count23 or fuzz" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 3
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  ms_per_step %.3f kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"], r["frac"]))
PY
}
run count13 --workload count13 $B || exit 5
run count23 --workload count23 --reads 10000000 $B || exit 5
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload count13 $B > $O/stats.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 8; }
f=$(ls -t $O/stats/*/*kernel_stats.csv | head -1); grep -E "k_c13" $f | cut -d, -f1-4 | sed 's/(.*",/",/'
