#!/bin/bash
# K1 at 5 M / 60 M / 200 M reads after the per-piece blocks are kept; the K1 / merge / ingest tests first
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "distinct or kmer_counter or merge or tools_cli or positions13 or failing_host or histogram_backend or count23_fixed_file" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" | tee -a $O/progress.txt
tail -4 $O/pytest_sel.log
for r in 5000000 60000000 200000000; do
  timeout -k 10 600 python bench.py --workload distinct23 --reads $r --steps 3 --warmup 1 --no-cpu-baseline > $O/distinct23_$r.json 2> $O/distinct23_$r.err; echo "d$r rc=$?" | tee -a $O/progress.txt
done
python - <<PY
import json
for r in (5000000, 60000000, 200000000):
    try:
        d = json.load(open("$O/distinct23_%d.json" % r))
        print(r, "ms_per_step %.1f" % d["ms_per_step"], "kernel_ms %.1f" % d["roofline"]["kernel_ms"], "reads/s %.1f M" % (d["value"] / 1e6), d["config"].get("distinct_kmers"))
    except Exception as e:
        print(r, "failed", e)
PY
