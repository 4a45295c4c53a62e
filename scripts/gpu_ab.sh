#!/bin/bash
# quick A/B pass: parity of the touched kernels, then the benches they move
set -e
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "13 or lookup or tf_ or queries" > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -2 gpurun_out/ab/pytest.log
python bench.py --workload count13 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab/count13.json 2> gpurun_out/ab/count13.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/ab/lookup23.json 2> gpurun_out/ab/lookup23.err
python bench.py --workload lookup13 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/lookup13.json 2> gpurun_out/ab/lookup13.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab/prof_c13 -o c13 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload count13 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/ab/*.json
find gpurun_out/ab/prof_c13 -name "*kernel_stats.csv" | head -1 | xargs head -12
