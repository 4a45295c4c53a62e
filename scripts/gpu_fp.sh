#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log
if [ $rc -ne 0 ]; then exit 3; fi
for v in "" "--no-fingerprint" "--no-fastpath" "--no-fastpath --no-fingerprint"; do
  n=$(echo "fp$v" | tr -d ' -')
  timeout -k 10 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $v > $O/bench_l23_$n.json 2> $O/bench_l23_$n.err || { echo failed $v; tail -20 $O/bench_l23_$n.err; exit 4; }
  python - <<PY
import json; d=json.load(open("$O/bench_l23_$n.json")); print("$n", "%.2f G lookups/s" % (d["value"]/1e9), "kernel %.3f ms" % d["roofline"]["kernel_ms"])
PY
done
timeout -k 10 600 python bench.py --workload coverage23 --steps 3 --warmup 1 > $O/bench_coverage23.json 2> $O/bench_coverage23.err && python -c "
import json; d=json.load(open('$O/bench_coverage23.json')); print('coverage23 %.2f G pos/s' % (d['roofline']['positions_per_sec']/1e9))"
timeout -k 10 600 python bench.py --workload count23 --reads 2000000 --steps 3 --warmup 1 > $O/bench_count23.json 2> $O/bench_count23.err && python -c "
import json; d=json.load(open('$O/bench_count23.json')); print('count23 %.1f M reads/s' % (d['value']/1e6))"
