#!/bin/bash
# soak: the differential fuzz generators on other seeds than the suite's, with the MSD grouping of A2 forced on (small inputs would take the
# sort path), once with the full LDS capacity and once with buckets set aside; K1's MSD path is the default at every size
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/soak; mkdir -p $O; cd $R
AIX_A2_MSD=1 AIX_FUZZ_SEEDS=300:380 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "queries_counts_positions or normalise_and_distinct" > $O/soak_msd.log 2>&1; rc=$?; tail -2 $O/soak_msd.log
[ $rc -eq 0 ] || { tail -40 $O/soak_msd.log; exit 3; }
AIX_A2_MSD=1 AIX_A2_TEST_CAP=6 AIX_POSITIONS_PIECE=4000 AIX_FUZZ_SEEDS=380:430 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "queries_counts_positions" > $O/soak_msd_cap.log 2>&1; rc=$?; tail -2 $O/soak_msd_cap.log
[ $rc -eq 0 ] || { tail -40 $O/soak_msd_cap.log; exit 3; }
AIX_C13_SHAPE=small AIX_FUZZ_SEEDS=300:340 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "13mer" > $O/soak_c13small.log 2>&1; rc=$?; tail -2 $O/soak_c13small.log
[ $rc -eq 0 ] || { tail -40 $O/soak_c13small.log; exit 3; }
