#!/bin/bash
# Copy what the judge should see from gpurun_out/final (scratch) into profiles/rNN (tracked): bench lines, per-workload kernel
# stats, PMC summaries, logs. Usage: bash scripts/collect_profiles.sh r02
set -e
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/gpurun_out/final; D=$R/profiles/${1:-r02}; mkdir -p $D
cp $S/bench_*.json $S/gather.jsonl $S/hostpath.json $S/progress.txt $S/pytest_gpu.log $S/smoke.log $D/ 2>/dev/null || true
for t in default l23rand l23mix c23 cov23 pos23 dist23 c13; do
  f=$(ls -t $S/stats/$t/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/${t}_kernel_stats.csv
done
for f in pmc_summary.txt pmc_summary.json pmc_traffic.json; do [ -f $S/$f ] && cp $S/$f $D/; done
[ -f $S/pmc_traffic.json ] && cp $S/pmc_traffic.json $R/profiles/pmc_traffic.json
ls $D | wc -l
