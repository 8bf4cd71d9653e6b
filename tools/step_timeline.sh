#!/bin/bash
# GPU box: the launches of the LAST training step of a bench tool in time order WITH their queue (stream): start offset, duration, queue,
# "|" = overlaps an earlier launch; then per queue: busy time, and the step's wall time.
# tools/step_timeline.sh <tag> <first-kernel-substring> <tool.py> [args]  -> gpurun_out/<tag>/timeline.txt
tag=$1; first=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 "$@" > /dev/null 2>&1
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$first" > $out/timeline.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
starts = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
fw = ev[starts[-2]:starts[-1]] if len(starts) > 1 else ev[starts[-1]:]
t0 = fw[0][0]; busy_end = t0; idle = 0; agg = {}; perq = {}
qs = sorted({e[3] for e in fw}); qn = {q: i for i, q in enumerate(qs)}
print("step: %d launches, %.3f ms from first start to the next step's first start; queues %s" % (len(fw), ((ev[starts[-1]][0] if len(starts) > 1 else max(e[1] for e in fw)) - t0) / 1e6, qs))
for s, e, n, q in fw:
    gap = max(0, s - busy_end); idle += gap
    name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]
    print("%9.1f us  +%7.1f us  gap %6.1f  q%d %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, qn[q], "| " if s < busy_end else "  ", name))
    busy_end = max(busy_end, e)
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    perq[q] = perq.get(q, 0.0) + (e - s) / 1e3
print("idle (no kernel running) %.1f us; kernel time summed %.1f us; per queue: %s" % (idle / 1e3, sum(a[1] for a in agg.values()), {qn[q]: round(v, 1) for q, v in perq.items()}))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("%8.1f us  x%3d  %s" % (a[1], a[0], k))
P
rm -rf $out/t
