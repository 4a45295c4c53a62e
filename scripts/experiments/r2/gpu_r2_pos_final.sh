#!/bin/bash
# positions23 with tf = occurrences in the reads (the reference pipeline) and with tf of the genome; MSD path and sort path; stats + PMC of the default
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/pmc $O/stats; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sharded_entry_points" > $O/pytest_dist.log 2>&1; rc=$?; tail -3 $O/pytest_dist.log
[ $rc -eq 0 ] || exit 3
Q="--no-cpu-baseline --no-secondary --no-gather-probe"
timeout -k 10 600 python bench.py --workload positions23 --reads 5000000 --steps 5 --warmup 1 > $O/bench_positions23reads5000000.json 2> $O/bench_positions23reads5000000.err || { tail -5 $O/bench_positions23reads5000000.err; exit 5; }
timeout -k 10 600 python bench.py --workload positions23 --reads 5000000 --positions-tf genome --steps 5 --warmup 1 $Q > $O/bench_positions23reads5000000_tfgenome.json 2> $O/b2.err || { tail -5 $O/b2.err; exit 5; }
AIX_A2_MSD=0 timeout -k 10 600 python bench.py --workload positions23 --reads 5000000 --steps 5 --warmup 1 $Q > $O/bench_positions23reads5000000_sortpath.json 2> $O/b3.err || { tail -5 $O/b3.err; exit 5; }
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_positions23*.json")):
    d=json.load(open(f)); r=d["roofline"]; print("%-60s %.4g %s ms %.3f frac %.3f %s" % (f.split("/")[-1], d["value"], d["unit"], d["ms_per_step"], r["frac"], d.get("cpu_baseline",{}).get("value")))
PY
export TMPDIR=/tmp; cd /tmp
rm -rf $O/stats/pos23 $O/pmc/pos23_*
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/pos23 -- python3 $R/bench.py --workload positions23 --reads 5000000 --steps 5 --warmup 1 $Q > $O/stats/pos23.json 2> $O/stats/pos23.err || { tail -5 $O/stats/pos23.err; exit 8; }
run_pmc () { tag=$1; grp=$2; ctrs=$3; shift 3
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc/${tag}_$grp -- python3 $R/bench.py "$@" --steps 3 --warmup 1 $Q > $O/pmc/${tag}_$grp.json 2> $O/pmc/${tag}_$grp.err || { echo "pmc $tag $grp failed"; tail -5 $O/pmc/${tag}_$grp.err; return 1; } }
A="--workload positions23 --reads 5000000"
run_pmc pos23 fetch "FETCH_SIZE" $A || exit 9
run_pmc pos23 write "WRITE_SIZE" $A || exit 9
run_pmc pos23 tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" $A || exit 9
run_pmc pos23 ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" $A || exit 9
run_pmc pos23 sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" $A || exit 9
run_pmc pos23 lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES" $A || exit 9
cd $R; python scripts/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1; grep -A3 "k_a2_final" $O/pmc_summary.txt | head; grep "=> fabric" $O/pmc_summary.txt
