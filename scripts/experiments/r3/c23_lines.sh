# count23 at 10 M reads: the call's own choice (back end 3 since the threshold went to 8 windows per key) and the probe path pinned
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/final; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "count23 or config4 or sharded or ingest" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python bench.py --workload count23 --reads 10000000 --steps 5 --warmup 1 > $O/bench_count23reads10000000.json 2>/dev/null || exit 2
timeout -k 10 300 python bench.py --workload count23 --reads 10000000 --probe-path --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_count23reads10000000probepath.json 2>/dev/null || exit 2
AIX_COUNT23_RUN=16 timeout -k 10 300 python bench.py --workload count23 --reads 10000000 --probe-path --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_count23_run16.json 2>/dev/null || exit 2
timeout -k 10 300 python bench.py --workload count23 --reads 10000000 --no-bucket-table --probe-path --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_count23reads10000000nobuckettable.json 2>/dev/null || exit 2
python - <<'PY'
import json
for f in ("bench_count23reads10000000", "bench_count23reads10000000probepath", "bench_count23_run16", "bench_count23reads10000000nobuckettable"):
    d = json.load(open("gpurun_out/final/%s.json" % f)); print(f, round(d["ms_per_step"], 2), "ms", d["config"].get("counting_backend"), round(d["roofline"]["frac"], 3))
PY
