#!/bin/bash
# count23 with the partition + histogram kernels on their own CUs (AIX_COUNT23_HIST_CUS=n[,style]): 40 M reads = 5.1 pieces of 2^30 windows
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/cumask; mkdir -p $O; cd $R
run () { tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload count23 --reads 40000000 --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -5 $O/$tag.err; return 1; }
  python -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', 'ms_per_step', round(d['ms_per_step'],2))"; }
run base X=1 && run cus32 AIX_COUNT23_HIST_CUS=32 && run cus64 AIX_COUNT23_HIST_CUS=64 && run cus96 AIX_COUNT23_HIST_CUS=96 && run cus128 AIX_COUNT23_HIST_CUS=128 && \
run cus64s1 AIX_COUNT23_HIST_CUS=64,1 && run cus32s1 AIX_COUNT23_HIST_CUS=32,1 && run cus128s1 AIX_COUNT23_HIST_CUS=128,1 && run base2 X=1
