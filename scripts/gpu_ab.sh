#!/bin/bash
set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "coverage or fuzz_queries or fuzz_13 or mirror or edge or sharded_entry" > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -2 gpurun_out/ab/pytest.log
for w in coverage13 coverage23; do
timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab/cov.json 2> gpurun_out/ab/cov.err
python -c "
import json; d=json.load(open('gpurun_out/ab/cov.json')); print('$w', '%.4g' % d['value'], d['unit'], 'ms', d['ms_per_step'], 'pos/s %.4g' % d['roofline']['positions_per_sec'])"
done
