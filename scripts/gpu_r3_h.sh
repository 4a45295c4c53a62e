#!/bin/bash
# run-per-lane probe (k_run23_slots) against the one-window-per-lane probe: parity, then count23 at 10 M reads alternating on one box
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "histogram_backend or count23" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" | tee -a $O/progress.txt
tail -4 $O/pytest_sel.log
for rep in 1 2; do for run in 0 16 32; do
  AIX_COUNT23_RUN=$run timeout -k 10 300 python bench.py --workload count23 --reads 10000000 --steps 10 --warmup 2 --no-cpu-baseline --no-gather-probe > $O/c23_run${run}_$rep.json 2> $O/c23_run${run}_$rep.err; echo "run=$run rep=$rep rc=$?" >> $O/progress.txt
  python - <<PY
import json
d = json.load(open("$O/c23_run${run}_$rep.json"))
print("run=$run rep=$rep ms_per_step %.2f kernel_ms %.2f reads/s %.1f M digest %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"] / 1e6, d.get("tf_digest")))
PY
done; done 2>&1 | tee $O/summary.txt
