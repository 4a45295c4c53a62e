#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of scripts/gpu_profile_all.sh into pmc_summary.txt / pmc_summary.json and the per-workload
traffic table bench.py reads (profiles/pmc_traffic.json).

Rules of /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 sections): counters are collected in their own runs
(--kernel-trace --pmc only, one counter group per run); FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 B, so reads are FETCH_SIZE x 1024 x 2 (calibrated here on k_gather: ~1.00 TCC_EA0_RDREQ per random
16-byte access, no 32-byte requests); both count fabric-side requests, Infinity-Cache hits included.
"""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# tag -> (bench workload key for pmc_traffic.json, kernels of the timed step: name substring -> launches per step)
TAGS = {
    "l23rand": ("lookup23:Q_rand", {"k_lookup23_ascii<0": 1}),
    "l23mix": ("lookup23:Q_mix", {"k_lookup23_ascii<0": 1}),
    "c23": ("count23", {"k_probe23_slots": None, "k_c13_split_chunked<aix::SrcSlots": None, "k_c13_hist_chunked": None}),
    "cov23": ("coverage23", {"k_coverage": 1}),
    "pos23": ("positions23", {"k_a2_probe": None, "k_k1_split<aix::A2Keys": None, "k_k1_count": None, "k_k1_scatter<unsigned long": None, "k_a2_final": None, "radix_sort": None}),
    "dist23": ("distinct23", {"k_window_codes": None, "k_k1_split": None, "k_k1_count": None, "k_k1_scatter": None, "k_k1_final": None, "k_k1_gather": None}),
    "c13": ("count13", {"k_c13_split_chunked<aix::Src13": None, "k_c13_hist_chunked": None}),
    "gather": (None, {"k_gather": 1}),
}


def counters(tag, grp):
    """kernel name -> counter -> list of per-dispatch values"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, "pmc", f"{tag}_{grp}", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def bench_json(tag, grp):
    try:
        return json.load(open(os.path.join(O, "pmc", f"{tag}_{grp}.json")))
    except Exception:
        return None


summary, traffic = {}, {}
for tag, (key, kernels) in TAGS.items():
    per_kernel = collections.defaultdict(dict)           # kernel substring -> counter -> SUM over the dispatches of one timed step
    steps = None
    for grp in ("fetch", "write", "tcc", "ea", "sq", "lds"):
        agg = counters(tag, grp)
        if not agg:
            continue
        bj = bench_json(tag, grp)
        # how often the measured entry point ran in the profiled process: warm-up + timed steps + whatever else the workload calls it for
        total_steps = (bj.get("roofline", {}).get("calls_in_process") or (bj["steps"] + bj["warmup"])) if bj else 4
        for sub in kernels:
            for kname, cs in agg.items():
                if sub not in kname:
                    continue
                for c, xs in cs.items():
                    # all dispatches of this kernel inside the timed + warm-up steps, divided by the number of steps
                    # (kernels of the set-up phase have other names or are launched once and are excluded by `sub`)
                    per_kernel[sub][c] = per_kernel[sub].get(c, 0.0) + sum(xs) / total_steps
                    per_kernel[sub]["_dispatches_per_step"] = len(xs) / total_steps
    if not per_kernel:
        continue
    summary[tag] = {k: dict(v) for k, v in per_kernel.items()}
    print(f"[{tag}] per timed step (sum over the step's dispatches of each kernel):")
    rd = wr = 0.0
    for sub, d in per_kernel.items():
        print(f"  {sub}  ({d.get('_dispatches_per_step', 0):.2f} dispatches per step)")
        for c, v in sorted(d.items()):
            if not c.startswith("_"):
                print(f"      {c:28s} {v:,.0f}")
        rd += d.get("FETCH_SIZE", 0.0) * 1024 * 2
        wr += d.get("WRITE_SIZE", 0.0) * 1024
        if d.get("SQ_LDS_IDX_ACTIVE"):
            print(f"      LDS bank-conflict share     {d.get('SQ_LDS_BANK_CONFLICT', 0) / d['SQ_LDS_IDX_ACTIVE']:.3f}")
        if d.get("SQ_WAVE_CYCLES"):
            print(f"      wait share of wave cycles   {d.get('SQ_WAIT_ANY', 0) / d['SQ_WAVE_CYCLES']:.3f} parked, {d.get('SQ_WAIT_INST_ANY', 0) / d['SQ_WAVE_CYCLES']:.3f} issue-stalled")
        if d.get("TCC_HIT_sum") is not None and (d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)) > 0:
            print(f"      L2 hit rate                 {d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum']):.3f}")
    print(f"  => fabric-side traffic per step: read {rd / 1e9:.2f} GB (FETCH_SIZE x 1024 x 2), write {wr / 1e9:.2f} GB")
    if key:
        bj = bench_json(tag, "fetch") or {}
        cfg, roof = bj.get("config", {}), bj.get("roofline", {})
        e = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
             "source": f"profiles/r03/pmc_summary.txt [{tag}]: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate runs), KiB x 1024, "
                       "FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 B; calibrated on k_gather). Fabric-side bytes: Infinity-Cache hits are counted."}
        for k in ("queries_per_launch", "reads_per_launch", "positions_per_launch", "windows_per_launch"):
            if k in roof:
                e[k] = roof[k]
        if "bucket_table" in cfg:
            e["bucket_table"] = int(cfg["bucket_table"])
            e["absence_filter"] = int(cfg.get("absence_filter_words", 0) > 0)
        traffic[key] = e

g = summary.get("gather", {}).get("k_gather", {})
if "TCC_EA0_RDREQ_sum" in g:
    print(f"calibration: k_gather issues 4.0e8 random 16-byte reads per launch; TCC_EA0_RDREQ = {g['TCC_EA0_RDREQ_sum']:,.0f} "
          f"({g['TCC_EA0_RDREQ_sum'] / 4e8:.3f} per access), 32-byte requests = {g.get('TCC_EA0_RDREQ_32B_sum', 0):,.0f}, "
          f"FETCH_SIZE x 1024 x 2 = {g.get('FETCH_SIZE', 0) * 2048 / 1e9:.2f} GB (51.2 GB of 128-byte lines)")
for f in sorted(glob.glob(os.path.join(O, "stats", "*", "*", "*kernel_stats.csv"))):
    print("==", f.replace(O + "/", ""))
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 8:
            print(f"    {r['Name'][:78]:78s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e6:9.3f} ms  {r['Percentage']}%")
json.dump({"summary": summary, "traffic": traffic}, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
if traffic:
    json.dump(traffic, open(os.path.join(O, "pmc_traffic.json"), "w"), indent=1)
