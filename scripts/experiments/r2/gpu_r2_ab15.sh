#!/bin/bash
# level-2 count pass with eight loads in flight per lane: parity (K1 / A2 / fuzz), then the positions23 / distinct23 lines and kernel stats of gpurun_out/final again
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/stats; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "positions or kmer_counter or distinct or tools or fuzz" > $O/pytest_k1a2.log 2>&1; rc=$?; tail -3 $O/pytest_k1a2.log
[ $rc -eq 0 ] || exit 3
Q="--no-cpu-baseline --no-secondary --no-gather-probe"
timeout -k 10 600 python bench.py --workload positions23 --reads 5000000 --steps 5 --warmup 1 > $O/bench_positions23reads5000000.json 2> $O/b1.err || { tail -5 $O/b1.err; exit 5; }
timeout -k 10 600 python bench.py --workload positions23 --reads 5000000 --positions-tf genome --steps 5 --warmup 1 $Q > $O/bench_positions23reads5000000_tfgenome.json 2> $O/b2.err || { tail -5 $O/b2.err; exit 5; }
AIX_A2_MSD=0 timeout -k 10 600 python bench.py --workload positions23 --reads 5000000 --steps 5 --warmup 1 $Q > $O/bench_positions23reads5000000_sortpath.json 2> $O/b3.err || { tail -5 $O/b3.err; exit 5; }
timeout -k 10 600 python bench.py --workload distinct23 --reads 5000000 --steps 5 --warmup 1 > $O/bench_distinct23reads5000000.json 2> $O/b4.err || { tail -5 $O/b4.err; exit 5; }
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_positions23*.json"))+["$O/bench_distinct23reads5000000.json"]:
    d=json.load(open(f)); r=d["roofline"]; print("%-60s %.4g %s ms %.3f frac %.3f traffic %s" % (f.split("/")[-1], d["value"], d["unit"], d["ms_per_step"], r["frac"], r.get("traffic")))
PY
export TMPDIR=/tmp; cd /tmp
for t in "pos23|--workload positions23 --reads 5000000" "dist23|--workload distinct23 --reads 5000000"; do tag=${t%%|*}; args=${t#*|}
  rm -rf $O/stats/$tag
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/$tag -- python3 $R/bench.py $args --steps 5 --warmup 1 $Q > $O/stats/$tag.json 2> $O/stats/$tag.err || { echo "stats $tag failed"; exit 8; }
  f=$(ls -t $O/stats/$tag/*/*kernel_stats.csv | head -1); grep -E "k_a2|k_k1_" $f | cut -d, -f1-4 | sed 's/(.*",/",/'
done
