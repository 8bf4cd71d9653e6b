#!/bin/bash
# GPU box: the launches of the LAST EEMFlow+ forward of tools/bench_plus.py in time order -> gpurun_out/<tag>/timeline.txt
# tools/plus_timeline.sh <tag>   (environment switches are inherited)
tag=${1:-plus_tl}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/p -- python3 tools/bench_plus.py > /dev/null 2>&1
find $out/p -name "*kernel_trace.csv" -exec cp {} $out/trace.csv \;
python3 - "$out" <<'P'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(out + '/trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'enc1_kernel' in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]['Start_Timestamp'])
with open(out + '/timeline.txt', 'w') as f:
    f.write("forward: %d launches, %.1f us\n" % (e - s, (int(rows[e]['Start_Timestamp']) - t0) / 1e3))
    for r in rows[s:e]:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:64]
        f.write("%8.1f +%6.1f %s grid=%s\n" % ((st - t0) / 1e3, (en - st) / 1e3, n, r.get('Grid_Size_X', '')))
P
rm -rf $out/p $out/trace.csv
