#!/bin/bash
# Round-2 A/B pass 4: K1 MSD path vs radix path, N3, host-copy thread count.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2ab4; mkdir -p $O; cd $R
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
step "targeted tests"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "kmer_counter or positions13 or distinct or fuzz_normalise or edge_cases or concurrent or tools_cli or sharded" > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log
[ $rc -eq 0 ] || { grep -n "^E " $O/pytest_gpu.log | head -20; exit 3; }
B="--no-cpu-baseline --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], r["kernel_ms"], r["frac"]))
PY
}
step "K1"
run distinct23_msd --workload distinct23 --reads 5000000 $B || exit 5
AIX_K1_ROCPRIM=1 run distinct23_radix --workload distinct23 --reads 5000000 $B || exit 5
step "host path copy threads"
for t in 8; do AIX_HOST_COPY_THREADS=$t timeout -k 10 300 python scripts/gpu_hostpath.py > $O/hostpath_t$t.json 2> $O/hostpath_t$t.err || { tail -5 $O/hostpath_t$t.err; exit 6; }
  python - <<PY
import json; d=json.load(open("$O/hostpath_t$t.json")); print("threads $t:", " ".join("%s=%.3g" % (k.split("(")[0][-24:], v["lookups_per_s"]) for k, v in d.items()))
PY
done
export TMPDIR=/tmp; cd /tmp
step "rocprofv3 kernel trace: distinct23"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1 -- python3 $R/bench.py --workload distinct23 --reads 5000000 --steps 5 --warmup 1 --no-cpu-baseline --no-gather-probe > $O/prof_k1.out 2> $O/prof_k1.err || exit 8
f=$(find $O/prof_k1 -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
for i, r in enumerate(csv.DictReader(open("$f"))):
    if i < 14: print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
step "done"
