#!/bin/bash
set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "tf or lookup or quer or golden or mirror or coverage" > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -2 gpurun_out/ab/pytest.log
for extra in "" "--query-mix" "--no-early-exit"; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-gather-probe $extra > gpurun_out/ab/l23.json 2> gpurun_out/ab/l23.err
python -c "
import json; d=json.load(open('gpurun_out/ab/l23.json')); print('$extra', '%.4g' % d['value'], 'ms', d['ms_per_step'])"
done
timeout -k 10 300 python bench.py --workload coverage23 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab/cov.json 2> gpurun_out/ab/cov.err
python -c "
import json; d=json.load(open('gpurun_out/ab/cov.json')); print('coverage', '%.4g' % d['value'], d['unit'], 'ms', d['ms_per_step'])"
