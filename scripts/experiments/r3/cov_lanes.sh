cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for l in 8 4 2; do
  timeout -k 10 300 python bench.py --workload coverage23 --bucket-lanes $l --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('lanes $l', round(d['ms_per_step'],1), 'ms')" || exit 1
done; done
