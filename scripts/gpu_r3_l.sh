#!/bin/bash
# positions fill after tf is read from the prefix sums; positions / 13-mer positions / fuzz tests first
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "positions or fuzz or python_mirrors or failing_host" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?" | tee -a $O/progress.txt
tail -3 $O/pytest_sel.log
for i in 1 2; do
timeout -k 10 300 python bench.py --workload positions23 --reads 5000000 --steps 5 --warmup 1 --no-cpu-baseline > $O/pos23_$i.json 2> $O/pos23_$i.err; python -c "
import json; d=json.load(open('$O/pos23_$i.json')); print('positions23 ms_per_step %.2f kernel_ms %.2f' % (d['ms_per_step'], d['roofline']['kernel_ms']), d['config']['host_buffer_call_ms'])"
done
