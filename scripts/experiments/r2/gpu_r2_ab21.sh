#!/bin/bash
# same-box A/B of K1's final stage: buckets of ~700 entries, 2 048-slot tables, five workgroups per CU (variant_A) against buckets of ~350
# entries, 1 024-slot tables, eight workgroups per CU (variant_B); parity of B first
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab21; mkdir -p $O; cd $R
cp aindex_amd/lib/variant_B.so aindex_amd/lib/libaindex_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "kmer_counter or distinct or tools or normalise or sharded_entry" > $O/pytest_B.log 2>&1; rc=$?; tail -2 $O/pytest_B.log
[ $rc -eq 0 ] || { tail -30 $O/pytest_B.log; exit 3; }
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
export TMPDIR=/tmp
for rep in 1 2; do for v in A B; do
cp $R/aindex_amd/lib/variant_$v.so $R/aindex_amd/lib/libaindex_hip.so
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${v}_$rep -- python3 $R/bench.py --workload distinct23 --reads 5000000 $B > $O/${v}_$rep.json 2> $O/${v}_$rep.err || { tail -5 $O/${v}_$rep.err; exit 8; }
f=$(ls -t $O/${v}_$rep/*/*kernel_stats.csv | head -1)
python - "$f" "$O/${v}_$rep.json" "$v $rep" <<'PY'
import csv,sys,json
d=json.load(open(sys.argv[2])); out=[sys.argv[3], "ms_per_step %.3f" % d["ms_per_step"]]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "k_k1_" in n and int(r["Calls"])>2: out.append("%s %.0f" % (n.split("(")[0].split("::")[-1][:22], float(r["AverageNs"])/1e3))
print(" | ".join(out))
PY
cd $R
done; done
