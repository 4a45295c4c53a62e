#!/bin/bash
# positions probe with two lanes per bucket line (default now): parity tests, then the positions23 line + kernel stats for profiles/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/stats; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "positions or count23 or sharded or bucket_table or queries_counts_positions" > $O/pytest_pos2.log 2>&1; rc=$?; tail -2 $O/pytest_pos2.log; [ $rc -eq 0 ] || { tail -30 $O/pytest_pos2.log; exit 3; }
timeout -k 10 600 python bench.py --workload positions23 --reads 5000000 --steps 5 --warmup 1 > $O/bench_positions23reads5000000.json 2> $O/bpos.err || { tail -5 $O/bpos.err; exit 5; }
python -c "
import json; d=json.load(open('$O/bench_positions23reads5000000.json')); print(d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'])"
export TMPDIR=/tmp; cd /tmp
rm -rf $O/stats/pos23
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/pos23 -- python3 $R/bench.py --workload positions23 --reads 5000000 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-gather-probe > $O/stats/pos23.json 2> $O/stats/pos23.err || exit 8
