#!/bin/bash
# quick A/B pass for the 13-mer counting kernels: parity first, then the bench and its per-kernel times
set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "13" > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -2 gpurun_out/ab/pytest.log
timeout -k 10 300 python bench.py --workload count13 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab/count13.json 2> gpurun_out/ab/count13.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab/prof_c13 -o c13 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload count13 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cut -c1-330 gpurun_out/ab/count13.json
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/ab/prof_c13/c13_kernel_stats.csv')))[:9]:
    print(r['Name'][:70].replace('\n',' '), r['Calls'], '%.3f ms avg' % (float(r['AverageNs'])/1e6))
PY
