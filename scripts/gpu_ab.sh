#!/bin/bash
# quick A/B pass for the 13-mer counting kernels: parity first, then the bench and its per-kernel times
set -e
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "13" > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
tail -2 gpurun_out/ab/pytest.log
python bench.py --workload count13 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab/count13.json 2> gpurun_out/ab/count13.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab/prof_c13 -o c13 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload count13 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cut -c1-330 gpurun_out/ab/count13.json
grep -h "k_c13" gpurun_out/ab/prof_c13/c13_kernel_stats.csv | awk -F'",' '{split($1,a,"("); print a[1], $2}' | cut -c1-70
