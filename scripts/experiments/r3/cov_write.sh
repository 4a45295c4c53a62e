# coverage23: time and fabric-side write traffic (WRITE_SIZE) of the build in the tree — spilled registers show up as writes beyond the 40 GB profile
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -q -x -m gpu -k "coverage or mirrors" 2>&1 | tail -2 || exit 1
for i in 1 2; do timeout -k 10 300 python3 bench.py --workload coverage23 --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('coverage23', round(d['ms_per_step'],1), 'ms')"; done
timeout -k 10 300 python3 bench.py --workload coverage13 --steps 3 --warmup 1 --no-cpu-baseline --no-gather-probe 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('coverage13', round(d['ms_per_step'],1), 'ms')"
cd /tmp; rm -rf /tmp/cw
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/cw -- python3 $GRAFT_REPO_ROOT/bench.py --workload coverage23 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-gather-probe > /dev/null 2>&1 || exit 2
python3 - <<'PY'
import csv, glob
f=glob.glob('/tmp/cw/*/*counter_collection.csv')[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'k_coverage' in r['Kernel_Name'] and r['Counter_Name']=='WRITE_SIZE']
print('k_coverage launches', len(v), 'WRITE_SIZE mean', sum(v)/len(v), '-> GB written per launch', sum(v)/len(v)*1024/1e9, '(x2 per the gfx950 note:', sum(v)/len(v)*2048/1e9, ')')
PY
