#!/bin/bash
set -e
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "positions" > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -2 gpurun_out/ab/pytest.log
python bench.py --workload positions23 --reads 5000000 --steps 3 --warmup 1 > gpurun_out/ab/pos.json 2> gpurun_out/ab/pos.err || { tail -20 gpurun_out/ab/pos.err; exit 1; }
cat gpurun_out/ab/pos.json
