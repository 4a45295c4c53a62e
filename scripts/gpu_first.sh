#!/bin/bash
# First GPU pass: parity tests, smoke, bench, rocprofv3 kernel stats. Logs under gpurun_out/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rocm-smi --showproductname 2>/dev/null | head -8 > $O/device.txt
nproc >> $O/device.txt; lscpu | grep "Model name" >> $O/device.txt
echo "== pytest gpu" | tee $O/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $O/progress.txt; tail -15 $O/pytest_gpu.log
if [ $rc -gt 1 ]; then exit $rc; fi
echo "== smoke" | tee -a $O/progress.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo smoke failed; tail -5 $O/smoke.log; exit 3; }
tail -2 $O/smoke.log
echo "== bench lookup23" | tee -a $O/progress.txt
timeout -k 10 900 python bench.py --steps 10 --warmup 2 > $O/bench_lookup23.json 2> $O/bench_lookup23.err || { echo bench failed; tail -20 $O/bench_lookup23.err; exit 4; }
cat $O/bench_lookup23.json; tail -5 $O/bench_lookup23.err
echo "== bench lookup23 no-fastpath" | tee -a $O/progress.txt
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --no-fastpath --no-cpu-baseline > $O/bench_lookup23_nofast.json 2> $O/bench_lookup23_nofast.err || { echo bench2 failed; tail -20 $O/bench_lookup23_nofast.err; exit 5; }
cat $O/bench_lookup23_nofast.json
echo "== bench count13 / count23 / lookup13" | tee -a $O/progress.txt
timeout -k 10 600 python bench.py --workload count13 --steps 3 --warmup 1 > $O/bench_count13.json 2> $O/bench_count13.err || { echo count13 failed; tail -20 $O/bench_count13.err; exit 6; }
cat $O/bench_count13.json
timeout -k 10 600 python bench.py --workload count23 --reads 2000000 --steps 3 --warmup 1 > $O/bench_count23.json 2> $O/bench_count23.err || { echo count23 failed; tail -20 $O/bench_count23.err; exit 7; }
cat $O/bench_count23.json
timeout -k 10 600 python bench.py --workload lookup13 --steps 10 --warmup 2 > $O/bench_lookup13.json 2> $O/bench_lookup13.err || { echo lookup13 failed; tail -20 $O/bench_lookup13.err; exit 8; }
cat $O/bench_lookup13.json
echo "== rocprofv3 kernel stats (lookup23)" | tee -a $O/progress.txt
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lookup23 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_lookup23.out 2> $O/prof_lookup23.err || { echo rocprof failed; tail -20 $O/prof_lookup23.err; exit 9; }
find $O/prof_lookup23 -name "*stats*" | head; 
echo "== done" | tee -a $O/progress.txt
