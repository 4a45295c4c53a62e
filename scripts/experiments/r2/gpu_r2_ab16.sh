#!/bin/bash
# 13-mer / slot split kernel: 512 x 24 (two workgroups per CU, default) against 1024 x 32 (AIX_C13_SHAPE=big): parity of both, then same-box timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab16; mkdir -p $O; cd $R
for shape in small big; do
AIX_C13_SHAPE=$shape timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "count13 or count23 or fuzz_13 or fuzz_count or config4" > $O/pytest_$shape.log 2>&1; rc=$?; echo "$shape: $(tail -1 $O/pytest_$shape.log)"
[ $rc -eq 0 ] || { tail -30 $O/pytest_$shape.log; exit 3; }
done
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
export TMPDIR=/tmp; cd /tmp
for rep in 1 2; do for shape in small big; do
for w in "c13|--workload count13" "c23|--workload count23 --reads 10000000"; do tag=${w%%|*}; args=${w#*|}
AIX_C13_SHAPE=$shape timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_${shape}_$rep -- python3 $R/bench.py $args $B > $O/${tag}_${shape}_$rep.json 2> $O/${tag}_${shape}_$rep.err || { tail -5 $O/${tag}_${shape}_$rep.err; exit 8; }
f=$(ls -t $O/${tag}_${shape}_$rep/*/*kernel_stats.csv | head -1)
python - "$f" "$O/${tag}_${shape}_$rep.json" "$tag $shape $rep" <<'PY'
import csv,sys,json
d=json.load(open(sys.argv[2])); out=[sys.argv[3], "ms_per_step %.3f" % d["ms_per_step"]]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "k_c13_split" in n or "k_c13_hist" in n: out.append("%s %.1f us x%s" % (n.split("(")[0][-28:], float(r["AverageNs"])/1e3, r["Calls"]))
print(" | ".join(out))
PY
done; done; done
