#!/bin/bash
# Full measurement pass: tests, smoke, default bench, secondary workloads, gather probes, rocprofv3 stats, PMC passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/pmc; cd $R
step () { echo "== $1" | tee -a $O/progress.txt; }
step "pytest gpu"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 3
step "smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || exit 4; tail -1 $O/smoke.log
step "bench default"
/usr/bin/env time -v true 2>/dev/null; ts=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 5; }
echo "bench default wall $(( $(date +%s) - ts )) s" | tee -a $O/progress.txt; cat $O/bench_default.json
step "secondary workloads"
for w in "lookup23 --no-early-exit --no-cpu-baseline --no-secondary --no-gather-probe" "lookup23 --no-early-exit --no-fingerprint --no-cpu-baseline --no-secondary --no-gather-probe" "lookup23 --no-fastpath --no-cpu-baseline --no-secondary --no-gather-probe" "lookup23 --query-mix --cpu-sample 2000000 --no-secondary --no-gather-probe" "lookup23 --query-mix --no-early-exit --no-cpu-baseline --no-secondary --no-gather-probe" "lookup23 --gpu-builder --no-cpu-baseline --no-secondary --no-gather-probe" "lookup13" "count13" "count23 --reads 2000000" "coverage23" "coverage13" "distinct23 --reads 5000000" "positions23 --reads 5000000" "normalize --reads 5000000"; do
  n=$(echo $w | sed 's/--no-cpu-baseline//; s/--no-secondary//; s/--no-gather-probe//; s/--cpu-sample 2000000//' | tr -d ' -'); timeout -k 10 600 python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$n.json 2> $O/bench_$n.err || { echo "$w failed"; tail -10 $O/bench_$n.err; exit 6; }
done
AIX_COUNT13_ATOMICS=1 timeout -k 10 600 python bench.py --workload count13 --steps 3 --warmup 1 > $O/bench_count13_atomics.json 2> /dev/null || exit 6
timeout -k 10 600 python scripts/gpu_hostpath.py > $O/hostpath.json 2> $O/hostpath.err || { echo hostpath failed; tail -5 $O/hostpath.err; exit 6; }
grep -h "index:" $O/bench_lookup23gpubuilder.err $O/bench_default.err | tee -a $O/progress.txt
step "gather probes"
rm -f $O/gather.jsonl
for cfg in "16384 16 1" "4096 16 1" "800 16 1" "61 16 1" "31 16 1" "2 16 1" "4096 8 1" "4096 16 4"; do set -- $cfg
  timeout -k 10 300 python bench.py --workload gather --table-mib $1 --elem $2 --unroll $3 --queries 400000000 --steps 5 --warmup 1 >> $O/gather.jsonl 2>> $O/gather.err || exit 7
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["metric"], "%.4g %s" % (d["value"], d["unit"]), "kernel_ms %.3f" % (d["roofline"]["kernel_ms"] or 0))
for l in open("$O/gather.jsonl"):
    d=json.loads(l); print(d["config"]["workload"], "%.1f G acc/s" % (d["value"]/1e9))
PY
export TMPDIR=/tmp; cd /tmp
step "rocprofv3 stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lookup23 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-gather-probe > $O/prof_lookup23.out 2> $O/prof_lookup23.err || exit 8
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_count13 -- python3 $R/bench.py --workload count13 --steps 3 --warmup 1 > $O/prof_count13.out 2> $O/prof_count13.err || exit 8
run_pmc () { name=$1; shift; ctrs=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc/$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-gather-probe "$@" > $O/pmc/$name.out 2> $O/pmc/$name.err || { echo "pmc $name failed"; tail -5 $O/pmc/$name.err; return 1; } }
step "pmc lookup23"
run_pmc l23_fetch "FETCH_SIZE" || exit 9
run_pmc l23_write "WRITE_SIZE" || exit 9
run_pmc l23_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" || exit 9
run_pmc l23_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" || exit 9
run_pmc l23_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" || exit 9
step "pmc calibration (gather 4 GiB x 16 B) and count13"
run_pmc gather_fetch "FETCH_SIZE" --workload gather --table-mib 4096 --elem 16 --unroll 1 --queries 400000000 || exit 10
run_pmc gather_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" --workload gather --table-mib 4096 --elem 16 --unroll 1 --queries 400000000 || exit 10
run_pmc c13_fetch "FETCH_SIZE" --workload count13 || exit 11
run_pmc c13_write "WRITE_SIZE" --workload count13 || exit 11
run_pmc c13_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" --workload count13 || exit 11
cd $R
python scripts/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
step "done"
