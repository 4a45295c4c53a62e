#!/bin/bash
# index shrink (side index instead of the key-record array): whole GPU suite, then the lookup / count / coverage / positions lines
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/progress.txt
tail -6 $O/pytest_gpu.log
