#!/bin/bash
# last refresh of the round: the lines the finer grids move (default bench with its secondaries, coverage at config-5 size, count23 per 10 M reads)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O; cd $R
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 5; }
timeout -k 10 300 python bench.py --workload coverage23 --steps 3 --warmup 1 > $O/bench_coverage23.json 2> $O/bcov.err || { tail -5 $O/bcov.err; exit 5; }
timeout -k 10 200 python bench.py --workload count23 --reads 10000000 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-gather-probe > $O/bench_count23reads10000000.json 2> $O/bc23.err || { tail -5 $O/bc23.err; exit 5; }
python - <<PY
import json
for f in ("bench_default","bench_coverage23","bench_count23reads10000000"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"])
d=json.load(open("$O/bench_default.json")); s=d["secondary"]
print("Q_mix", s["lookup23_Q_mix"]["value"], "count23_strong", s["count23_strong"]["value"], s["count23_strong"]["ms_per_step"], "count13", s["count13_dense"]["ms_per_step"])
PY
