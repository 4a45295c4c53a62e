#!/bin/bash
# Refresh the parts of gpurun_out/final that later commits of the round changed (count23 lane default, K1 / A2 final stages): full GPU
# suite + smoke, default bench, --gpus 2 rehearsal, count23 / distinct23 benches, their kernel stats and PMC passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/pmc $O/stats; cd $R
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
Q="--no-cpu-baseline --no-secondary --no-gather-probe"
step "refresh: pytest gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 3
step "refresh: smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || exit 4; tail -1 $O/smoke.log
step "refresh: bench default"
ts=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 5; }
echo "bench default wall $(( $(date +%s) - ts )) s" | tee -a $O/progress.txt
step "refresh: bench --gpus 2 (rehearsal)"
timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2_rehearsal.err || { tail -20 $O/bench_gpus2_rehearsal.err; exit 5; }
step "refresh: workloads"
for w in "count23 --reads 10000000" "count23 --reads 10000000 --bucket-lanes 8 --no-cpu-baseline" "distinct23 --reads 5000000"; do
  n=$(echo $w | sed "s/--no-cpu-baseline//" | tr -d ' -'); timeout -k 10 600 python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$n.json 2> $O/bench_$n.err || { echo "$w failed"; tail -10 $O/bench_$n.err; exit 6; }
done
AIX_K1_ROCPRIM=1 timeout -k 10 600 python bench.py --workload distinct23 --reads 5000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_distinct23reads5000000_radixpath.json 2> /dev/null || exit 6
export TMPDIR=/tmp; cd /tmp
step "refresh: stats + pmc"
for t in "c23|--workload count23 --reads 10000000" "dist23|--workload distinct23 --reads 5000000"; do tag=${t%%|*}; args=${t#*|}
  rm -rf $O/stats/$tag
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/$tag -- python3 $R/bench.py $args --steps 5 --warmup 1 $Q > $O/stats/$tag.json 2> $O/stats/$tag.err || { echo "stats $tag failed"; exit 8; }
  for g in "fetch|FETCH_SIZE" "write|WRITE_SIZE" "tcc|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "ea|TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" "sq|SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "lds|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES"; do grp=${g%%|*}; ctrs=${g#*|}
    rm -rf $O/pmc/${tag}_$grp
    timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc/${tag}_$grp -- python3 $R/bench.py $args --steps 3 --warmup 1 $Q > $O/pmc/${tag}_$grp.json 2> $O/pmc/${tag}_$grp.err || { echo "pmc $tag $grp failed"; exit 9; }
  done
done
step "refresh done"
