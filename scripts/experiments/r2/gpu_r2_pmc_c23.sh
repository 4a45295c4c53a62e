#!/bin/bash
# PMC passes of count23 and count13 on the final code (2^30-window passes, 256-workgroup split), then the default bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/pmc $O/stats; cd $R
step () { echo "== $1 $(date +%T)" | tee -a $O/progress.txt; }
Q="--no-cpu-baseline --no-secondary --no-gather-probe"
export TMPDIR=/tmp; cd /tmp
step "pmc refresh: c23 / c13"
for t in "c23|--workload count23 --reads 10000000" "c13|--workload count13"; do tag=${t%%|*}; args=${t#*|}
  for g in "fetch|FETCH_SIZE" "write|WRITE_SIZE" "tcc|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "ea|TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" "sq|SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "lds|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES"; do grp=${g%%|*}; ctrs=${g#*|}
    rm -rf $O/pmc/${tag}_$grp
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc/${tag}_$grp -- python3 $R/bench.py $args --steps 3 --warmup 1 $Q > $O/pmc/${tag}_$grp.json 2> $O/pmc/${tag}_$grp.err || { echo "pmc $tag $grp failed"; tail -5 $O/pmc/${tag}_$grp.err; exit 9; }
  done
done
step "pmc refresh done"
