# split kernel of count23 with the tile of slots held in registers between its two passes (AIX_C23_SLOTS_REGS=1) against re-reading it
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
for rep in 1 2; do for r in 0 1; do
  rm -rf /tmp/sr_$r
  AIX_C23_SLOTS_REGS=$r AIX_COUNT23_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sr_$r -- python3 bench.py --workload count23 --reads 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe > /tmp/sr_$r.json 2>/dev/null || exit 1
  f=$(ls /tmp/sr_$r/*/*kernel_stats.csv | head -1)
  python3 -c "
import csv, json
for r in csv.DictReader(open('$f')):
    if 'split_chunked' in r['Name']: print('regs $r', r['Name'][:48], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
"
  AIX_C23_SLOTS_REGS=$r timeout -k 10 300 python3 bench.py --workload count23 --reads 40000000 --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('regs $r count23 40 M reads', round(d['ms_per_step'],1), 'ms')"
done; done
