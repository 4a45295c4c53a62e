#!/bin/bash
# PMC passes for the positions fill (kernel-trace + pmc only, one counter group per run)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmcpos; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
Q="--no-cpu-baseline --no-secondary --no-gather-probe --steps 3 --warmup 1"
run_pmc () { grp=$1; ctrs=$2
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/$grp -- python3 $R/bench.py --workload positions23 --reads 5000000 $Q > $O/$grp.json 2> $O/$grp.err || { echo "pmc $grp failed"; tail -5 $O/$grp.err; return 1; } }
run_pmc sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" || exit 9
run_pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU" || exit 9
run_pmc lds2 "SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" || exit 9
cd $R
python - <<'PY'
import csv,glob,collections,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmcpos"
for grp in ("sq","lds","lds2"):
    fs=glob.glob(f"{O}/{grp}/*/*counter_collection.csv")
    if not fs: print(grp,"no file"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items():
        if any(x in k for x in ("k_a2_final","k_k1_split","k_k1_scatter","k_k1_count","k_a2_probe")):
            print(grp,k,{a:f"{b:.3g}" for a,b in v.items()})
PY
