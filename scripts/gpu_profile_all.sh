#!/bin/bash
# Full measurement pass behind profiles/rNN/ (run through gpurun, one PART per call: a call is limited to 20 minutes).
#   A1: GPU tests, smoke, the default bench, the 2-rank rehearsal      A2: every other bench line
#   A3: host path, gather probes, rocprofv3 --kernel-trace --stats per workload and of the exact default command
#   B1 / B2: the rocprofv3 --pmc passes, one counter group per run, kernel-trace only
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/pmc $O/stats; cd $R
PART=${PART:-A1}
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
Q="--no-cpu-baseline --no-secondary --no-gather-probe"
# tag | bench.py arguments of the PMC / stats passes
TAGS=(
 "l23rand|--workload lookup23"
 "l23mix|--workload lookup23 --query-mix"
 "c23|--workload count23 --reads 10000000 --probe-path"
 "cov23|--workload coverage23"
 "pos23|--workload positions23 --reads 5000000"
 "dist23|--workload distinct23 --reads 5000000"
 "c13|--workload count13"
 "gather|--workload gather --table-mib 4096 --elem 16 --unroll 1 --queries 400000000"
)
if [ "$PART" = "A1" ]; then
step "pytest gpu"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 3
step "smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || exit 4; tail -1 $O/smoke.log
step "bench default"
ts=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 5; }
echo "bench default wall $(( $(date +%s) - ts )) s" | tee -a $O/progress.txt
step "bench --gpus 2 (rehearsal: two ranks on one device, gloo)"
timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2_rehearsal.err || { tail -20 $O/bench_gpus2_rehearsal.err; exit 5; }
fi
if [ "$PART" = "A2" ]; then
step "workloads"
for w in "lookup23 --query-mix --cpu-sample 2000000 --no-secondary" "lookup23 --no-bucket-table $Q" "lookup23 --no-absence-filter $Q" "lookup23 --query-mix --no-bucket-table $Q" \
         "lookup23 --gpu-builder $Q" "lookup13" "count13" "count23 --reads 10000000" "count23 --reads 10000000 --probe-path --no-cpu-baseline" "count23 --reads 10000000 --no-bucket-table --probe-path --no-cpu-baseline" \
         "coverage23" "coverage23 --no-bucket-table --seqs 100000 --no-cpu-baseline" "coverage13" "distinct23 --reads 5000000" "positions23 --reads 5000000" "normalize --reads 5000000" \
         "e2e13" "e2e23"; do
  n=$(echo $w | sed "s/--no-cpu-baseline//; s/--probe-path/probepath/; s/--no-secondary//; s/--no-gather-probe//; s/--cpu-sample 2000000//" | tr -d ' -'); timeout -k 10 600 python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$n.json 2> $O/bench_$n.err || { echo "$w failed"; tail -10 $O/bench_$n.err; exit 6; }
done
timeout -k 10 600 python bench.py --workload distinct23 --reads 200000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_distinct23reads200000000.json 2> $O/bench_distinct23reads200000000.err || exit 6
AIX_COUNT23_ATOMICS=1 timeout -k 10 600 python bench.py --workload count23 --reads 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_count23_atomics.json 2> /dev/null || exit 6
AIX_COUNT23_RUN=16 timeout -k 10 600 python bench.py --workload count23 --reads 10000000 --probe-path --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_count23_run16.json 2> /dev/null || exit 6
AIX_COUNT13_ATOMICS=1 timeout -k 10 600 python bench.py --workload count13 --steps 3 --warmup 1 > $O/bench_count13_atomics.json 2> /dev/null || exit 6
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.load(open(f)); r=d.get("roofline",{}); print("%-46s %-40s %.4g %s  kernel_ms %.3f frac %.3f" % (f.split("/")[-1], d["metric"], d["value"], d["unit"], r.get("kernel_ms") or 0, r.get("frac") or 0))
PY
fi
if [ "$PART" = "A3" ]; then
step "host path"
timeout -k 10 900 python scripts/gpu_hostpath.py > $O/hostpath.json 2> $O/hostpath.err || { echo hostpath failed; tail -5 $O/hostpath.err; exit 6; }
step "gather probes"
rm -f $O/gather.jsonl
for cfg in "16384 16 1" "4096 16 1" "1600 16 1" "800 16 1" "100 16 1" "61 16 1" "31 16 1" "2 16 1" "4096 8 1" "4096 16 4" "1600 64 1" "1600 128 1"; do set -- $cfg
  timeout -k 10 300 python bench.py --workload gather --table-mib $1 --elem $2 --unroll $3 --queries 400000000 --steps 5 --warmup 1 >> $O/gather.jsonl 2>> $O/gather.err || exit 7
done
export TMPDIR=/tmp; cd /tmp
step "rocprofv3 --kernel-trace --stats"
for t in "${TAGS[@]}"; do tag=${t%%|*}; args=${t#*|}; [ $tag = gather ] && continue
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/$tag -- python3 $R/bench.py $args --steps 5 --warmup 1 $Q > $O/stats/$tag.json 2> $O/stats/$tag.err || { echo "stats $tag failed"; tail -5 $O/stats/$tag.err; exit 8; }
done
step "rocprofv3 --kernel-trace --stats of the exact default command"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/default -- python3 $R/bench.py > $O/stats/default.json 2> $O/stats/default.err || { echo "stats default failed"; tail -5 $O/stats/default.err; exit 8; }
fi
if [ "$PART" = "B1" ] || [ "$PART" = "B2" ]; then
export TMPDIR=/tmp; cd /tmp
run_pmc () { tag=$1; grp=$2; ctrs=$3; shift 3
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc/${tag}_$grp -- python3 $R/bench.py "$@" --steps 3 --warmup 1 $Q > $O/pmc/${tag}_$grp.json 2> $O/pmc/${tag}_$grp.err || { echo "pmc $tag $grp failed"; tail -5 $O/pmc/${tag}_$grp.err; return 1; } }
i=0
for t in "${TAGS[@]}"; do tag=${t%%|*}; args=${t#*|}; i=$((i + 1))
  if [ "$PART" = "B1" ] && [ $i -gt 4 ]; then continue; fi
  if [ "$PART" = "B2" ] && [ $i -le 4 ]; then continue; fi
  step "pmc $tag"
  run_pmc $tag fetch "FETCH_SIZE" $args || exit 9
  run_pmc $tag write "WRITE_SIZE" $args || exit 9
  [ $tag = gather ] || run_pmc $tag tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" $args || exit 9
  run_pmc $tag ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" $args || exit 9
  [ $tag = gather ] || run_pmc $tag sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" $args || exit 9
  if [ $tag = c13 ] || [ $tag = c23 ]; then run_pmc $tag lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES" $args || exit 9; fi
done
fi
cd $R
# (scripts/summarize_pmc.py gpurun_out/final runs on the build host after B1 and B2 have been merged)
step "done $PART"
