#!/bin/bash
# count23: probe of piece i + 1 on a second stream while piece i is partitioned and added (default) against one stream (AIX_COUNT23_OVERLAP=0):
# parity first (pieces forced small so that several are in flight), then same-box timings at 10 M and 200 M reads
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab22; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "count23 or config4 or fuzz_queries or sharded_entry" > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc -eq 0 ] || { tail -30 $O/pytest.log; exit 3; }
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  ms_per_step %.3f  digest %s" % ("$n", d["value"], d["unit"], d["ms_per_step"], d.get("tf_digest")))
PY
}
for rep in 1 2 3; do
run ov_$rep --workload count23 --reads 10000000 $B || exit 5
AIX_COUNT23_OVERLAP=0 run one_$rep --workload count23 --reads 10000000 $B || exit 5
done
run strong_ov --workload count23 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline || exit 5
AIX_COUNT23_OVERLAP=0 run strong_one --workload count23 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline || exit 5
