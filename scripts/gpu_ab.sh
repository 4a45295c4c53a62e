#!/bin/bash
set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "count23 or corrupt or (fuzz_queries and (1 or 2))" > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -2 gpurun_out/ab/pytest.log
for extra in "" "--no-node-table"; do
timeout -k 10 300 python bench.py --workload count23 --steps 5 --warmup 2 $extra > gpurun_out/ab/c23.json 2> gpurun_out/ab/c23.err
python -c "
import json; d=json.load(open('gpurun_out/ab/c23.json')); print('count23 $extra', '%.4g' % d['value'], d['unit'], 'ms', d['ms_per_step'], d['roofline']['random_read']['frac'])"
done
