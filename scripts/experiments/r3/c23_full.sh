# config 4 at full size (200 M reads) with the probe / histogram overlap on and off
cd ${GRAFT_REPO_ROOT:-/root/repo}
for ov in 1 0; do
  AIX_COUNT23_OVERLAP=$ov timeout -k 10 400 python3 bench.py --workload count23 --reads 200000000 --steps 2 --warmup 1 --no-cpu-baseline --no-gather-probe 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('overlap $ov: 200 M reads', round(d['ms_per_step'],1), 'ms', round(d['value']/1e6,1), 'M reads/s')" || exit 1
done
