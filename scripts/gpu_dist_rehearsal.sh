#!/bin/bash
# Rehearse the N=2 bench path on ONE GPU: two ranks share cuda:0, collectives over gloo. Small sizes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export AIX_BENCH_ONE_DEVICE=1 AIX_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for wl in lookup23 count23 count13; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 3 --warmup 1 --workload $wl --queries 20000000 --genome 5000000 --reads 500000 > $O/dist2_$wl.json 2> $O/dist2_$wl.err || { echo "dist2 $wl failed"; tail -30 $O/dist2_$wl.err; exit 3; }
  cat $O/dist2_$wl.json
done
echo done
