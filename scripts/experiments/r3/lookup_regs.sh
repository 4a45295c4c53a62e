# after the rare path of query23 re-reads its bytes (k_lookup23_ascii 69 -> 59 VGPRs: eight waves per SIMD): headline, Q_mix, coverage
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 1000 python -m pytest tests -q -x -m gpu -k "q23 or golden or oracle or fuzz or mirrors or config3 or coverage or canonical or bucket" 2>&1 | tail -2 || exit 1
for i in 1 2; do
timeout -k 10 300 python3 bench.py --workload lookup23 --steps 20 --warmup 10 --no-cpu-baseline --no-gather-probe --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('Q_rand', round(d['roofline']['kernel_ms'],4), 'Q_mix', round(d['secondary']['lookup23_Q_mix']['ms_per_step'],4))"
timeout -k 10 300 python3 bench.py --workload coverage23 --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('coverage23', round(d['ms_per_step'],1), 'ms')"
done
