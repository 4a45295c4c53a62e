#!/bin/bash
# workspace-halving test + the counting tests around it
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ws; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests -m gpu -x -q -k "workspace or count13 or count23 or config4" > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log; exit $rc
