#!/bin/bash
# rocprofv3 --kernel-trace --stats of the EXACT default command (python3 bench.py): the summary the bench line's roofline.kernel_ms must agree with
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O/stats; cd /tmp; export TMPDIR=/tmp
rm -rf $O/stats/default
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/default -- python3 $R/bench.py > $O/stats/default.json 2> $O/stats/default.err || { tail -5 $O/stats/default.err; exit 8; }
python3 - <<PY
import json,glob,csv
d=json.load(open("$O/stats/default.json")); print("bench line kernel_ms", d["roofline"]["kernel_ms"], "value", d["value"])
f=sorted(glob.glob("$O/stats/default/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "k_lookup23_ascii<0" in r["Name"] or "k_lookup23_ascii<(aix" in r["Name"]: print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6)
PY
