#!/bin/bash
# Template for a quick A/B pass on the GPU box: parity of what was touched first, then the benches it moves.
# Edit the -k expression and the workloads, run with `gpurun --timeout 900 -- 'bash scripts/gpu_ab.sh'`.
set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "count13 or lookup" > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -2 gpurun_out/ab/pytest.log
for w in "lookup23 --no-cpu-baseline --no-secondary --no-gather-probe" "count13 --no-cpu-baseline"; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err
  python -c "
import json; d=json.load(open('gpurun_out/ab/b.json')); print('$w', '%.4g' % d['value'], d['unit'], 'ms', '%.3f' % d['ms_per_step'])"
done
