#!/bin/bash
# round-2 session 2: 13-mer split (register cursors, precomputed write-out deltas) and A2 final stage (padded runs): parity + timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab9; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "count13 or count23 or positions or fuzz" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 3
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
export TMPDIR=/tmp; cd /tmp
for w in "c13|--workload count13" "c23|--workload count23 --reads 10000000" "pos23|--workload positions23 --reads 5000000"; do tag=${w%%|*}; args=${w#*|}
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/bench.py $args $B > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; exit 8; }
f=$(ls -t $O/$tag/*/*kernel_stats.csv | head -1); echo "== $tag"; grep -E "k_c13|k_a2|k_k1_(split|count|scatter)|k_probe23" $f | cut -d, -f1-4 | sed 's/(.*",/",/'
python - <<PY
import json; d=json.load(open("$O/$tag.json")); print("$tag ms_per_step", d["ms_per_step"], d["value"], d["unit"])
PY
done
