#!/bin/bash
# K1 scaling with the cached blocks; then the exact default bench command (with the e2e secondaries)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
for r in 5000000 60000000 200000000; do
  timeout -k 10 600 python bench.py --workload distinct23 --reads $r --steps 3 --warmup 1 --no-cpu-baseline > $O/distinct23_$r.json 2> $O/distinct23_$r.err; echo "d$r rc=$?" | tee -a $O/progress.txt
done
python - <<PY
import json
for r in (5000000, 60000000, 200000000):
    try:
        d = json.load(open("$O/distinct23_%d.json" % r))
        print(r, "ms_per_step %.1f" % d["ms_per_step"], "reads/s %.1f M" % (d["value"] / 1e6), d["config"].get("distinct_kmers"))
    except Exception as e:
        print(r, "failed", e)
PY
T0=$(date +%s.%N); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$? wall $(echo "$(date +%s.%N) - $T0" | bc) s" | tee -a $O/progress.txt
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
s = d["secondary"]
for k, v in s.items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("value", "unit", "seconds", "GBps", "error", "ms_per_step", "tool_phases")})
    if k == "e2e13":
        print("   in_process", {kk: vv for kk, vv in v.get("in_process", {}).items() if kk != "stats"}, v.get("in_process_result_to_memory"), v.get("fastq_in_process", {}).get("GBps"), v.get("cpu_baseline"))
PY
