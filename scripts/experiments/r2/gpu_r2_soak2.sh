#!/bin/bash
# soak on the final code: seeds 600..659 of all four generators on the default paths, 660..699 with the histogram back end of count23 and
# small passes forced (several passes, second stream) and the MSD grouping of A2 at test sizes
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/soak2; mkdir -p $O; cd $R
AIX_FUZZ_SEEDS=600:660 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/soak_default.log 2>&1; rc=$?; tail -2 $O/soak_default.log
[ $rc -eq 0 ] || { tail -40 $O/soak_default.log; exit 3; }
AIX_COUNT23_HIST_MIN=0 AIX_COUNT23_PIECE=5000 AIX_COUNT13_PIECE=7000 AIX_A2_MSD=1 AIX_FUZZ_SEEDS=660:700 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/soak_pieces.log 2>&1; rc=$?; tail -2 $O/soak_pieces.log
[ $rc -eq 0 ] || { tail -40 $O/soak_pieces.log; exit 3; }
