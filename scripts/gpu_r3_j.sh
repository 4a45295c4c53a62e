#!/bin/bash
# random-read ceiling against the bytes a lane consumes per access (16 / 32 / 64 / 128) over the 1.6 GiB and 4 GiB tables
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3j; mkdir -p $O
for mib in 1600 4096; do for elem in 16 32 64 128; do
  timeout -k 10 200 python bench.py --workload gather --table-mib $mib --elem $elem --queries 400000000 --steps 5 --warmup 1 > $O/gather_${mib}_$elem.json 2> $O/gather_${mib}_$elem.err
  python - <<PY
import json
d = json.load(open("$O/gather_${mib}_$elem.json"))
print("table $mib MiB elem $elem B: %.1f G accesses/s, %.2f TB/s of 128-byte lines, %.2f TB/s consumed" % (d["value"] / 1e9, d["value"] * 128 / 1e12, d["value"] * $elem / 1e12))
PY
done; done 2>&1 | tee $O/summary.txt
