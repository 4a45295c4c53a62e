#!/bin/bash
# round 3, first GPU pass: the streaming-ingestion tests, the merge tests, the config-3 oracle test, then the end-to-end benches
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu > $O/ingest.log 2>&1; echo "ingest rc=$?" | tee -a $O/progress.txt
tail -5 $O/ingest.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config3_full_size or kmer_counter or tools_cli or count13_golden or positions_fill_equals" > $O/parity_sel.log 2>&1; echo "parity_sel rc=$?" | tee -a $O/progress.txt
tail -5 $O/parity_sel.log
timeout -k 10 600 python bench.py --workload e2e13 --e2e-gb 8 > $O/bench_e2e13.json 2> $O/bench_e2e13.err; echo "e2e13 rc=$?" | tee -a $O/progress.txt
tail -c 3000 $O/bench_e2e13.json
timeout -k 10 600 python bench.py --workload e2e23 --e2e-gb 8 > $O/bench_e2e23.json 2> $O/bench_e2e23.err; echo "e2e23 rc=$?" | tee -a $O/progress.txt
tail -c 2500 $O/bench_e2e23.json
