#!/bin/bash
# count23: streaming probe on the minimizer-keyed table (opt-in) against the default, same box, with kernel stats: how much is k_fix23?
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab19; mkdir -p $O; cd $R
if [ "${TESTS:-0}" = "1" ]; then
AIX_MINIMIZER_TABLE=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "count23 or fuzz_queries" > $O/pytest.log 2>&1; rc=$?; tail -2 $O/pytest.log
[ $rc -eq 0 ] || { tail -30 $O/pytest.log; exit 3; }
fi
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
export TMPDIR=/tmp; cd /tmp
for cfg in "def|" "mk2|AIX_MINIMIZER_TABLE=1" "mk1|AIX_MINIMIZER_TABLE=1 AIX_MINIMIZER_LOAD=1" "mk4|AIX_MINIMIZER_TABLE=1 AIX_MINIMIZER_LOAD=4" "def2|"; do tag=${cfg%%|*}; envs=${cfg#*|}

( export $envs; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/bench.py --workload count23 --reads 10000000 $B > $O/$tag.json 2> $O/$tag.err ) || { tail -5 $O/$tag.err; exit 8; }
f=$(ls -t $O/$tag/*/*kernel_stats.csv | head -1)
python - "$f" "$O/$tag.json" "$tag" <<'PY'
import csv,sys,json
d=json.load(open(sys.argv[2])); out=[sys.argv[3], "ms_per_step %.3f" % d["ms_per_step"], "index MB %d" % (d["config"].get("index_hbm_bytes",0)/1e6)]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(k in n for k in ("k_stream23","k_fix23","k_probe23","k_c13_split","k_c13_hist")): out.append("%s %.1f us x%s" % (n.split("(")[0][-24:], float(r["AverageNs"])/1e3, r["Calls"]))
print(" | ".join(out))
PY
done
