set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 python -m pytest tests -q -x -m gpu -k "coverage or mirrors or fuzz" 2>&1 | tail -3 || exit 1
for w in "coverage23" "coverage13" "coverage23 --seqs 200000" ; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$w', round(d['ms_per_step'],1), 'ms', d['config']['nonzero_fraction'])" || exit 1
done
