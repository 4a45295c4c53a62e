#!/bin/bash
set -e
mkdir -p gpurun_out/ab
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab/prof_c13 -o c13 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload count13 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/ab/prof_c13/c13_kernel_stats.csv')))[:6]:
    print(r['Name'][:70].replace('\n',' '), r['Calls'], '%.3f ms avg' % (float(r['AverageNs'])/1e6))
PY
