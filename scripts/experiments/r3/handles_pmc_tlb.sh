#!/bin/bash
# Q_mix levels: translation-side counters per handle (eight identical handles in one process, AIX_OPEN_TUNE=0; dispatch order = handle order)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3tlb; mkdir -p $O
export TMPDIR=/tmp AIX_OPEN_TUNE=0
cd /tmp
run() { tag=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$tag -- python3 $R/scripts/gpu_r3_handles.py 8 4 > $O/pmc_$tag.json 2> $O/pmc_$tag.err || { echo "pass $tag failed"; tail -3 $O/pmc_$tag.err; } }
run d TCP_CLIENT_UTCL1_INFLIGHT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
run e GRBM_UTCL2_BUSY TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum
run f TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_LFIFO_FULL_sum
cd $R
python - <<PY
import csv, glob, json, collections
for tag in "def":
    fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print("pass", tag, "no csv"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "k_lookup23_ascii" in r["Kernel_Name"]]
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    meas = ids[8:]
    try:
        times = json.load(open("$O/pmc_%s.json" % tag))
    except Exception:
        times = []
    for h in range(8):
        mine = meas[4 * h: 4 * h + 4]
        agg = collections.Counter()
        for d in mine:
            for k, v in disp[d].items():
                agg[k] += v / max(1, len(mine))
        t = times[h]["kernel_ms"] if h < len(times) else None
        print("pass", tag, "handle", h, "kernel_ms(under pmc)", round(t, 3) if t else None, {k.replace("TCP_UTCL1_", "").replace("_sum", ""): round(v / 1e6, 3) for k, v in agg.items()})
PY
