set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/final
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "python_mirrors or wrapper" 2>&1 | tail -3 && \
timeout -k 10 900 python scripts/gpu_hostpath.py > gpurun_out/final/hostpath.json 2> gpurun_out/final/hostpath.err && python -c "
import json; d=json.load(open('gpurun_out/final/hostpath.json'))
for k,v in d.items(): print(k, round(v['lookups_per_s']/1e6,1), 'M/s', v.get('steps_s',''))"
