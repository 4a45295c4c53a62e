#!/bin/bash
# Build host, after the parts of scripts/gpu_profile_all.sh have been merged into gpurun_out/final: PMC summaries (text + json), the files the
# judge reads copied into profiles/rNN, a trimmed kernel trace of the exact default command and the check of its lookup launches against the line.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); RN=${1:-r03}; S=$R/gpurun_out/final
python3 $R/scripts/summarize_pmc.py $S > $S/pmc_summary.txt
bash $R/scripts/collect_profiles.sh $RN > /dev/null
python3 - "$R" "$RN" <<'PY'
import csv, glob, os, json, sys
R, RN = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(R + '/gpurun_out/final/stats/default/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
with open(f'{R}/profiles/{RN}/default_kernel_trace.csv', 'w') as o:
    o.write('dispatch,kernel,start_ns,duration_us,grid_x,workgroup_x,vgpr,lds\n')
    for r in rows:
        o.write('%s,"%s",%s,%.3f,%s,%s,%s,%s\n' % (r['Dispatch_Id'], r['Kernel_Name'][:90].replace('"', "'"), r['Start_Timestamp'],
                                                 (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X'], r['Workgroup_Size_X'], r['VGPR_Count'], r['LDS_Block_Size']))
lk = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows if 'k_lookup23_ascii<0' in r['Kernel_Name'] and int(r['Grid_Size_X']) >= 65536 * 256 // 2]
d = json.load(open(R + '/gpurun_out/final/stats/default.json'))
W, K = d['warmup'], d['steps']
st = [r for r in csv.DictReader(open(f'{R}/profiles/{RN}/default_kernel_stats.csv')) if 'k_lookup23_ascii<0' in r['Name']]
txt = ["k_lookup23_ascii<0, true, 8> launches of the exact default command (`python bench.py`) under rocprofv3 --kernel-trace --stats"]
if st: txt.append("(profiles/%s/default_kernel_stats.csv averages all of them: %s calls, %.3f ms)" % (RN, st[0]['Calls'], float(st[0]['AverageNs']) / 1e6))
txt.append("durations of the 100 M-query launches in launch order, ms: " + " ".join("%.3f" % x for x in lk))
rand, mix = lk[:W + K], lk[W + K:]
txt.append("the %d Q_rand launches (%d warm-up + %d timed): mean %.3f ms, the %d timed ones %.3f ms; the line printed by that very run says kernel_ms %.3f"
           % (len(rand), W, K, sum(rand) / len(rand), K, sum(rand[W:]) / len(rand[W:]), d['roofline']['kernel_ms']))
if mix: txt.append("the %d Q_mix launches of the secondary measurement: mean %.3f ms (the line: %.3f)" % (len(mix), sum(mix) / len(mix), d['secondary']['lookup23_Q_mix']['ms_per_step']))
d2 = json.load(open(R + '/gpurun_out/final/stats/l23rand.json'))
st2 = [r for r in csv.DictReader(open(f'{R}/profiles/{RN}/l23rand_kernel_stats.csv')) if 'k_lookup23_ascii<0' in r['Name']]
if st2: txt.append("headline workload alone (l23rand_kernel_stats.csv): %s calls, average %.3f ms; the bench line of that run says kernel_ms %.3f" % (st2[0]['Calls'], float(st2[0]['AverageNs']) / 1e6, d2['roofline']['kernel_ms']))
open(f'{R}/profiles/{RN}/default_lookup_launches.txt', 'w').write("\n".join(txt) + "\n")
print("\n".join(txt))
PY
