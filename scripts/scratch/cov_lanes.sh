cd ${GRAFT_REPO_ROOT:-/root/repo}
for l in 8 4 2 1; do
  timeout -k 10 300 python bench.py --workload coverage23 --bucket-lanes $l --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('lanes $l', round(d['ms_per_step'],1), 'ms', d['config']['nonzero_fraction'])" || exit 1
done
for l in 8 2; do
  timeout -k 10 300 python bench.py --workload lookup23 --query-mix --bucket-lanes $l --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-gather-probe 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('qmix lanes $l', d['roofline']['kernel_ms'])" || exit 1
done
