#!/bin/bash
# last call of the round: the whole GPU suite + smoke on the final code, the lines / kernel stats later commits touched (distinct23, count13)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/stats; cd $R
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
step "last: pytest gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 3
step "last: smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || exit 4; tail -1 $O/smoke.log
Q="--no-cpu-baseline --no-secondary --no-gather-probe"
step "last: distinct23 / count13"
timeout -k 10 600 python bench.py --workload distinct23 --reads 5000000 --steps 5 --warmup 1 > $O/bench_distinct23reads5000000.json 2> $O/b4.err || { tail -5 $O/b4.err; exit 5; }
timeout -k 10 600 python bench.py --workload count13 --steps 5 --warmup 1 > $O/bench_count13.json 2> $O/b5.err || { tail -5 $O/b5.err; exit 5; }
timeout -k 10 600 python bench.py --workload count23 --reads 10000000 --steps 10 --warmup 2 $Q > $O/bench_count23reads10000000.json 2> $O/b6.err || { tail -5 $O/b6.err; exit 5; }
export TMPDIR=/tmp; cd /tmp
for t in "dist23|--workload distinct23 --reads 5000000" "c13|--workload count13" "c23|--workload count23 --reads 10000000"; do tag=${t%%|*}; args=${t#*|}
rm -rf $O/stats/$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/$tag -- python3 $R/bench.py $args --steps 5 --warmup 1 $Q > $O/stats/$tag.json 2> $O/stats/$tag.err || exit 8
done
step "last done"
