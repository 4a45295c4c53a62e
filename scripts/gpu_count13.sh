#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest 13-mer + all gpu" | tee $O/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log
if [ $rc -gt 1 ]; then exit $rc; fi
echo "== bench count13 partitioned" | tee -a $O/progress.txt
timeout -k 10 600 python bench.py --workload count13 --steps 5 --warmup 1 > $O/bench_count13_part.json 2> $O/bench_count13_part.err || { echo failed; tail -20 $O/bench_count13_part.err; exit 3; }
cat $O/bench_count13_part.json
echo "== bench count13 atomics" | tee -a $O/progress.txt
AIX_COUNT13_ATOMICS=1 timeout -k 10 600 python bench.py --workload count13 --steps 3 --warmup 1 > $O/bench_count13_atomics.json 2> $O/bench_count13_atomics.err || { echo failed; exit 4; }
cat $O/bench_count13_atomics.json
echo "== bench coverage23" | tee -a $O/progress.txt
timeout -k 10 600 python bench.py --workload coverage23 --steps 3 --warmup 1 > $O/bench_coverage23.json 2> $O/bench_coverage23.err || { echo failed; tail -20 $O/bench_coverage23.err; exit 5; }
cat $O/bench_coverage23.json
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_count13 -- python3 $R/bench.py --workload count13 --steps 3 --warmup 1 > $O/prof_count13.out 2> $O/prof_count13.err || { echo rocprof failed; tail -5 $O/prof_count13.err; exit 6; }
cat $O/prof_count13/*/*kernel_stats.csv | cut -c1-160 | head -12
