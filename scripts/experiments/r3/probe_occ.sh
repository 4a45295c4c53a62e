cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
for pad in 0 20000 40000 52000 65000; do
  rm -rf /tmp/po_$pad
  AIX_PROBE_LDS_PAD=$pad AIX_COUNT23_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po_$pad -- python3 bench.py --workload count23 --reads 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe > /dev/null 2>&1 || exit 1
  f=$(ls /tmp/po_$pad/*/*kernel_stats.csv | head -1)
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'probe23_slots' in r['Name'] or 'split_chunked' in r['Name']: print('pad $pad', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
"
done
