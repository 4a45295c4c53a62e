#!/bin/bash
# After the coverage fix (positions [2^31, 2^32) mod 2^32 of a batch): the coverage lines, K1 lines (now self-checked), host path, and the cov23 trace / counters again
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/pmc $O/stats; cd $R
Q="--no-cpu-baseline --no-secondary --no-gather-probe"
PART=${PART:-A}
if [ "$PART" = "A" ]; then
for w in "coverage23" "coverage23 --no-bucket-table --seqs 100000 --no-cpu-baseline" "coverage13" "distinct23 --reads 5000000"; do
  n=$(echo $w | sed "s/--no-cpu-baseline//" | tr -d ' -'); timeout -k 10 600 python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$n.json 2> $O/bench_$n.err || { echo "$w failed"; tail -10 $O/bench_$n.err; exit 6; }
  python -c "import json; d=json.load(open('$O/bench_$n.json')); print('$n', d['value'], d['unit'], d['ms_per_step'], d['config'].get('nonzero_fraction'))"
done
timeout -k 10 600 python bench.py --workload distinct23 --reads 200000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_distinct23reads200000000.json 2> $O/bench_distinct23reads200000000.err || { tail -5 $O/bench_distinct23reads200000000.err; exit 6; }
python -c "import json; d=json.load(open('$O/bench_distinct23reads200000000.json')); print('distinct 200M', d['ms_per_step'], d['config'])"
timeout -k 10 900 python scripts/gpu_hostpath.py > $O/hostpath.json 2> $O/hostpath.err || { echo hostpath failed; tail -5 $O/hostpath.err; exit 6; }
export TMPDIR=/tmp; cd /tmp
rm -rf $O/stats/cov23
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/cov23 -- python3 $R/bench.py --workload coverage23 --steps 5 --warmup 1 $Q > $O/stats/cov23.json 2> $O/stats/cov23.err || { echo "stats failed"; exit 8; }
fi
if [ "$PART" = "B" ]; then
export TMPDIR=/tmp; cd /tmp
run_pmc () { tag=$1; grp=$2; ctrs=$3; shift 3; rm -rf $O/pmc/${tag}_$grp
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc/${tag}_$grp -- python3 $R/bench.py "$@" --steps 3 --warmup 1 $Q > $O/pmc/${tag}_$grp.json 2> $O/pmc/${tag}_$grp.err || { echo "pmc $tag $grp failed"; tail -5 $O/pmc/${tag}_$grp.err; return 1; } }
args="--workload coverage23"
run_pmc cov23 fetch "FETCH_SIZE" $args && run_pmc cov23 write "WRITE_SIZE" $args && run_pmc cov23 tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" $args && \
run_pmc cov23 ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" $args && \
run_pmc cov23 sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" $args || exit 9
fi
echo "done $PART"
