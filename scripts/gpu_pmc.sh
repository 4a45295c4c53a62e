#!/bin/bash
# PMC passes (each its own rocprofv3 run, --pmc only with --kernel-trace) + random-gather roofline probes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/pmc
cd $R
echo "== gather probes" | tee $O/progress.txt
for cfg in "4096 16 1" "4096 16 4" "4096 8 1" "4096 8 4" "800 16 1" "800 16 4" "31 16 1" "31 16 4" "16384 16 4"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --workload gather --table-mib $1 --elem $2 --unroll $3 --queries 400000000 --steps 5 --warmup 1 >> $O/gather.jsonl 2>> $O/gather.err || { echo gather failed $cfg; tail -5 $O/gather.err; exit 3; }
done
python - <<'PY'
import json
for l in open("gpurun_out/gather.jsonl"):
    d=json.loads(l); print(d["config"]["workload"], "%.1f G acc/s" % (d["value"]/1e9), "%.0f GB/s@64B" % d["roofline"]["achieved"])
PY
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $O/pmc/counters_list.txt 2>&1 || true
run_pmc () { # name counters... -- bench args
  name=$1; shift; ctrs=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc/$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/pmc/$name.out 2> $O/pmc/$name.err || { echo "pmc $name failed"; tail -5 $O/pmc/$name.err; return 1; }
}
echo "== pmc lookup23" | tee -a $O/progress.txt
run_pmc l23_fetch "FETCH_SIZE" || exit 4
run_pmc l23_write "WRITE_SIZE" || exit 4
run_pmc l23_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" || exit 4
run_pmc l23_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" || exit 4
run_pmc l23_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" || exit 4
echo "== pmc lookup13 (calibration: 1e8 random 8-byte reads of a 512 MiB table)" | tee -a $O/progress.txt
run_pmc l13_fetch "FETCH_SIZE" --workload lookup13 || exit 5
run_pmc l13_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" --workload lookup13 || exit 5
run_pmc gather_fetch "FETCH_SIZE" --workload gather --table-mib 4096 --elem 16 --unroll 4 --queries 400000000 || exit 6
run_pmc gather_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" --workload gather --table-mib 4096 --elem 16 --unroll 4 --queries 400000000 || exit 6
run_pmc c13_fetch "FETCH_SIZE" --workload count13 || exit 7
echo "== done" | tee -a $O/progress.txt
find $O/pmc -name "*counter_collection.csv" | head -20
