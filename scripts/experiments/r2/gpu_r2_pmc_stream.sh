#!/bin/bash
# PMC of the streaming counter (minimizer-keyed table, one key per bucket) beside the default probe
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmcstream; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
Q="--workload count23 --reads 10000000 --no-cpu-baseline --no-secondary --no-gather-probe --steps 3 --warmup 1"
for cfg in "mk|1" "def|0"; do tag=${cfg%%|*}; on=${cfg#*|}
for g in "sq|SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "fetch|FETCH_SIZE" "lds|SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do grp=${g%%|*}; ctrs=${g#*|}
( export AIX_MINIMIZER_TABLE=$on AIX_MINIMIZER_LOAD=1; timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/${tag}_$grp -- python3 $R/bench.py $Q > $O/${tag}_$grp.json 2> $O/${tag}_$grp.err ) || { tail -5 $O/${tag}_$grp.err; exit 9; }
done; done
cd $R
python - <<'PY'
import csv,glob,collections,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmcstream"
for tag in ("mk","def"):
  for grp in ("sq","fetch","lds"):
    fs=glob.glob(f"{O}/{tag}_{grp}/*/*counter_collection.csv")
    if not fs: continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0][-30:]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k,v in acc.items():
        if any(x in k for x in ("k_stream23","k_probe23")):
            print(tag,grp,k,{a:f"{b/n[(k,a)]:.4g}" for a,b in v.items()})
PY
