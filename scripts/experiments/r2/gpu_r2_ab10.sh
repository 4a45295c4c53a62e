#!/bin/bash
# round-2 session 2: A2 final stage (packed slot records, single atomic) and the 32-bit 13-mer answer table: parity + timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab10; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "13 or positions or coverage or fuzz or mirrors" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 3
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 5 --warmup 1"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]
print("%-28s %10.4g %s  ms_per_step %.3f kernel_ms %.3f  frac %.3f" % ("$n", d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"], r["frac"]))
PY
}
run lookup13 --workload lookup13 $B || exit 5
run coverage13 --workload coverage13 $B || exit 5
export TMPDIR=/tmp; cd /tmp
for w in "pos23|--workload positions23 --reads 5000000"; do tag=${w%%|*}; args=${w#*|}
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/bench.py $args $B > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; exit 8; }
f=$(ls -t $O/$tag/*/*kernel_stats.csv | head -1); echo "== $tag"; grep -E "k_c13|k_a2|k_k1_(split|count|scatter)|k_probe23" $f | cut -d, -f1-4 | sed 's/(.*",/",/'
python - <<PY
import json; d=json.load(open("$O/$tag.json")); print("$tag ms_per_step", d["ms_per_step"], d["value"], d["unit"])
PY
done
