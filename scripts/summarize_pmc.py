#!/usr/bin/env python3
"""Condense rocprofv3 counter_collection / kernel_stats CSVs of scripts/gpu_profile_all.sh into text + JSON."""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]


def counters(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, "pmc", name, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean_for(name, kern):
    out = {}
    for k, v in counters(name).items():
        if kern in k:
            for c, xs in v.items():
                out[c] = sum(xs) / len(xs)
    return out


summary = {}
for tag, kern, passes in (("lookup23", "k_lookup23_ascii", ["l23_fetch", "l23_write", "l23_tcc", "l23_ea", "l23_sq"]),
                          ("gather_4GiB_16B", "k_gather", ["gather_fetch", "gather_ea"]),
                          ("count13_split", "k_c13_split_chunked", ["c13_fetch", "c13_write", "c13_lds"]),
                          ("count13_hist", "k_c13_hist_chunked", ["c13_fetch", "c13_write", "c13_lds"])):
    d = {}
    for p in passes:
        d.update(mean_for(p, kern))
    summary[tag] = d
    print(f"[{tag}] per launch (mean over launches):")
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {v:,.0f}")
# traffic per launch, MI355X_MICROARCH.md rules: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte
# requests at 64 B -> x2 (calibrated below on the gather kernel: 1.00 request per random access)
traffic = {}
for tag in summary:
    d = summary[tag]
    if "FETCH_SIZE" in d:
        rd = d["FETCH_SIZE"] * 1024 * 2
        wr = d.get("WRITE_SIZE", 0.0) * 1024
        traffic[tag] = {"read_bytes_x2_corrected": rd, "write_bytes": wr, "bytes_per_launch": rd + wr,
                        "fetch_size_raw_kib": d["FETCH_SIZE"], "write_size_raw_kib": d.get("WRITE_SIZE")}
        print(f"[{tag}] HBM-side traffic per launch: read {rd/1e9:.2f} GB (FETCH_SIZE x 1024 x 2), write {wr/1e9:.2f} GB")
g = summary.get("gather_4GiB_16B", {})
if "TCC_EA0_RDREQ_sum" in g:
    print(f"calibration: gather issues 4.0e8 random 16-byte reads; TCC_EA0_RDREQ = {g['TCC_EA0_RDREQ_sum']:,.0f} "
          f"({g['TCC_EA0_RDREQ_sum']/4e8:.3f} per access), 32B requests = {g.get('TCC_EA0_RDREQ_32B_sum', 0):,.0f}")
for f in glob.glob(os.path.join(O, "prof_*", "*", "*kernel_stats.csv")):
    print("==", f.replace(O + "/", ""))
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 8:
            print(f"    {r['Name'][:70]:70s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {r['Percentage']}%")
json.dump({"summary": summary, "traffic": traffic}, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
