#!/bin/bash
# round 3: whole GPU suite on the new code, then K1 at 5 M and at config-4 size (200 M reads = 13 pieces merged)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/progress.txt
tail -8 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --workload distinct23 --reads 5000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/distinct23_5M.json 2> $O/distinct23_5M.err; echo "d5 rc=$?" | tee -a $O/progress.txt
timeout -k 10 600 python bench.py --workload distinct23 --reads 200000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/distinct23_200M.json 2> $O/distinct23_200M.err; echo "d200 rc=$?" | tee -a $O/progress.txt
python - <<PY
import json
for f in ("distinct23_5M", "distinct23_200M"):
    try:
        d = json.load(open("$O/" + f + ".json"))
        print(f, "ms_per_step %.1f" % d["ms_per_step"], "reads/s %.1f M" % (d["value"] / 1e6), d["config"].get("distinct_kmers"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $O/distinct23_200M.err
