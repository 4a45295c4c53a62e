#!/bin/bash
# round 3: end-to-end ingestion against the part size; one box
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu > $O/ingest.log 2>&1; echo "ingest rc=$?" | tee -a $O/progress.txt
tail -3 $O/ingest.log
for mb in 64 32 128 256; do
  AIX_INGEST_PART_MB=$mb timeout -k 10 300 python bench.py --workload e2e13 --e2e-gb 8 --no-cpu-baseline > $O/e2e13_$mb.json 2> $O/e2e13_$mb.err; echo "e2e13 $mb rc=$?" | tee -a $O/progress.txt
  python - <<PY
import json
d=json.load(open("$O/e2e13_$mb.json"))["config"]
ip=d["in_process"]; st=ip["stats"]
print("part $mb MB: tool %.3f s | in-process file %.3f s (%.1f GB/s) read %.3f h2d %.3f wait %.3f compute %.3f out %.3f | to memory %.3f s (%.1f GB/s) | fastq %.1f GB/s" % (d["seconds"], ip["seconds"], ip["GBps"], st["seconds_read"], st["seconds_h2d"], st["seconds_wait"], st["seconds_compute"], st["seconds_output"], d["in_process_result_to_memory"]["seconds"], d["in_process_result_to_memory"]["GBps"], d["fastq_in_process"]["GBps"]))
print(d.get("tool_phases"))
PY
done 2>&1 | tee -a $O/summary.txt
for mb in 64 256; do
  AIX_INGEST_PART_MB=$mb timeout -k 10 300 python bench.py --workload e2e23 --e2e-gb 8 --no-cpu-baseline > $O/e2e23_$mb.json 2> $O/e2e23_$mb.err; echo "e2e23 $mb rc=$?" | tee -a $O/progress.txt
  python - <<PY
import json
d=json.load(open("$O/e2e23_$mb.json"))["config"]; st=d["stats"]
print("e2e23 part $mb MB: %.3f s (%.1f GB/s, %.0f M reads/s) read %.3f h2d %.3f wait %.3f compute %.3f out %.3f" % (d["seconds"], d["GBps"], d["value"]/1e6, st["seconds_read"], st["seconds_h2d"], st["seconds_wait"], st["seconds_compute"], st["seconds_output"]))
PY
done 2>&1 | tee -a $O/summary.txt
