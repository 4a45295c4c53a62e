#!/bin/bash
# absence filter size (bits per key) against Q_rand / Q_mix, same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab29; mkdir -p $O; cd $R
B="--no-cpu-baseline --no-secondary --no-gather-probe --steps 10 --warmup 2"
run () { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -8 $O/$n.err; return 1; }
  python - <<PY
import json; d=json.load(open("$O/$n.json")); r=d["roofline"]; c=d["config"]
print("%-20s %10.4g %s  kernel_ms %.3f  filter words %d  bucket lines/query %.4f" % ("$n", d["value"], d["unit"], r["kernel_ms"], c.get("absence_filter_words",0), c.get("bucket_lines_per_query",0)))
PY
}
for bits in 16 6 8 12 24 32 16; do
AIX_BLOOM_BITS=$bits run qrand_b$bits --workload lookup23 $B || exit 5
AIX_BLOOM_BITS=$bits run qmix_b$bits --workload lookup23 --query-mix $B || exit 5
done
