#!/bin/bash
# GPU box: the launches of the LAST forward of a bench tool in time order - start offset, duration, idle gap before it (no kernel running),
# "|" = overlaps an earlier launch.  tools/fwd_timeline.sh <tag> <first-kernel-substring> <tool.py> [args]  -> gpurun_out/<tag>/timeline.txt
tag=$1; first=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BENCH_N=3 BENCH_WARM=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 "$@" > /dev/null 2>&1
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$first" > $out/timeline.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
fw = ev[starts[-1]:]
t0 = fw[0][0]; busy_end = t0; idle = 0; agg = {}
print("forward: %d launches, %.3f ms from first start to last end" % (len(fw), (max(e[1] for e in fw) - t0) / 1e6))
for s, e, n in fw:
    gap = max(0, s - busy_end); idle += gap
    name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    print("%9.1f us  +%7.1f us  gap %6.1f  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, "| " if s < busy_end else "", name))
    busy_end = max(busy_end, e)
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
print("idle (no kernel running) %.1f us; kernel time summed %.1f us" % (idle / 1e3, sum(a[1] for a in agg.values())))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("%8.1f us  x%3d  %s" % (a[1], a[0], k))
P
rm -rf $out/t
tail -30 $out/timeline.txt
