#!/bin/bash
# GPU box: the launches of ONE E-RAFT forward at batch $1 in time order - start offset, duration, idle gap before it (no kernel of the
# forward running) - from rocprofv3's kernel trace -> gpurun_out/<tag>/eraft_b<batch>_timeline.txt
b=${1:-1}; tag=${2:-eraft}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BENCH_N=3 BENCH_WARM=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 tools/bench_eraft.py $b > /dev/null 2>&1
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/eraft_b${b}_timeline.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# forwards start with the pad kernel
starts = [i for i, e in enumerate(ev) if "pad2_kernel" in e[2] or "pad4_kernel" in e[2]]
lo = starts[-1]; hi = len(ev)
fw = ev[lo:hi]
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
tail -40 $out/eraft_b${b}_timeline.txt
