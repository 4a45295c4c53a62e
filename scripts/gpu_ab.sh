#!/bin/bash
set -e
mkdir -p gpurun_out/ab
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -2 gpurun_out/ab/pytest.log
timeout -k 10 300 python bench.py --workload distinct23 --reads 5000000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab/d23.json 2> gpurun_out/ab/d23.err
python -c "
import json; d=json.load(open('gpurun_out/ab/d23.json')); print('distinct23', d['value'], d['ms_per_step'], d['config']['distinct_kmers'])"
timeout -k 10 300 python bench.py --workload positions23 --reads 5000000 --steps 3 --warmup 1 > gpurun_out/ab/p23.json 2> gpurun_out/ab/p23.err
python -c "
import json; d=json.load(open('gpurun_out/ab/p23.json')); print('positions23', d['value'], d['ms_per_step'], d['config']['host_buffer_call_ms'])"
