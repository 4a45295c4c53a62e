#!/bin/bash
# Q_mix placement spread: per-handle counters of k_lookup23_ascii, five identical handles in one process (dispatch order = handle order)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -E "Counter_Name" | sed -e 's/\s\+/ /g' | grep -i -E "UTCL|TLB|TAG_STALL|TCP_PENDING|TCP_TCC_READ_REQ|TCP_TCP_LATENCY|TA_BUSY|TCP_TCC_READ_REQ_LATENCY|TD_BUSY|MALL|EA_RDREQ|_GMI|HBM" > $O/counters2.txt
cd /tmp
run() { tag=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$tag -- python3 $R/scripts/gpu_r3_handles.py 5 4 > $O/pmc_$tag.json 2> $O/pmc_$tag.err || { echo "pass $tag failed"; tail -3 $O/pmc_$tag.err; } }
run a TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
run b TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_BUSY_sum
run c TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum
cd $R
python - <<PY
import csv, glob, json, collections
for tag in "abc":
    fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print("pass", tag, "no csv"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "k_lookup23_ascii" in r["Kernel_Name"]]
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    # 5 warm-up launches (one per handle), then 4 launches per handle
    meas = ids[5:]
    try:
        times = json.load(open("$O/pmc_%s.json" % tag))
    except Exception:
        times = []
    for h in range(5):
        mine = meas[4 * h: 4 * h + 4]
        agg = collections.Counter()
        for d in mine:
            for k, v in disp[d].items():
                agg[k] += v / max(1, len(mine))
        t = times[h]["kernel_ms"] if h < len(times) else None
        print("pass", tag, "handle", h, "kernel_ms(under pmc)", t, {k: round(v / 1e6, 3) for k, v in agg.items()}, "(millions per launch)")
PY
