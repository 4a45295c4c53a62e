#!/bin/bash
# kernel durations of count23 (10 M reads): one-window-per-lane probe against the run-per-lane probe
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for run in 0 16; do
  AIX_COUNT23_RUN=$run timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_run$run -- python3 $R/bench.py --workload count23 --reads 10000000 --steps 5 --warmup 1 --no-cpu-baseline --no-gather-probe > $O/c23_run$run.json 2> $O/c23_run$run.err; echo "run=$run rc=$?"
done
cd $R
python - <<PY
import csv, glob
for run in (0, 16):
    for f in glob.glob("$O/prof_run%d/**/*kernel_stats.csv" % run, recursive=True):
        rows = list(csv.DictReader(open(f)))
        print("run", run)
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:7]:
            print("  %-64s calls %4s avg %8.3f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
