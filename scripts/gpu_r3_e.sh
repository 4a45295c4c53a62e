#!/bin/bash
# where do 11 s per 200 M reads go in K1? kernel trace of a 4-piece run
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o k1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload distinct23 --reads 60000000 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/d60.json 2> $GRAFT_REPO_ROOT/$O/d60.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python - <<PY
import json, glob, csv
d = json.load(open("$O/d60.json")); print("ms_per_step", d["ms_per_step"], d["config"].get("distinct_kmers"))
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f, "total kernel ms", tot / 1e6)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        print("  %-70s calls %5s total %9.2f ms avg %9.3f ms" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
