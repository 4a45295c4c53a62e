#!/bin/bash
# Exercise every RCCL call of the N>1 bench path with a 1-rank nccl process group on one GPU (broadcast of the .pf,
# all_reduce of int32/int64/float64, barrier), launched exactly like the driver launches N>1.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export AIX_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for wl in lookup23 count13 count23; do
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 --steps 3 --warmup 1 --workload $wl --queries 20000000 --genome 5000000 --reads 1000000 --no-cpu-baseline > $O/nccl1_$wl.json 2> $O/nccl1_$wl.err || { echo "nccl1 $wl failed"; tail -30 $O/nccl1_$wl.err; exit 3; }
  python -c "
import json; d=json.loads([l for l in open('$O/nccl1_$wl.json') if l.startswith('{')][-1]); print('$wl', d['metric'], '%.4g' % d['value'], d.get('secondary'))"
done
echo done
