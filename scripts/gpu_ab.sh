#!/bin/bash
set -e
mkdir -p gpurun_out/ab
for extra in "" "--no-early-exit" "--no-early-exit --no-fingerprint"; do
timeout -k 10 300 python bench.py --workload coverage23 --steps 5 --warmup 1 --no-cpu-baseline $extra > gpurun_out/ab/cov.json 2> gpurun_out/ab/cov.err
python -c "
import json; d=json.load(open('gpurun_out/ab/cov.json')); print('coverage23 $extra', '%.4g' % d['value'], d['unit'], 'ms', d['ms_per_step'], 'pos/s %.4g' % d['roofline']['positions_per_sec'])"
done
