#!/bin/bash
# GPU box: per-launch kernel trace of a command, grouped by (kernel, grid): tools/trace_groups.sh <tag> -- <python args...>
tag=$1; shift; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BENCH_N=2 BENCH_WARM=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/p -- python3 "$@" > /dev/null 2>&1
f=$(find $out/p -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    key = (name.split("(")[0][:40], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
rows = sorted(g.items(), key=lambda kv: -sum(kv[1]))
print("total %.2f ms over %d launches" % (tot / 1e3, sum(len(v) for v in g.values())))
for k, v in rows[:28]:
    print("%5.1f%% %8.1f us avg x %4d  %-40s grid %s x %s x %s / %s" % (100 * sum(v) / tot, sum(v) / len(v), len(v), k[0], k[1], k[2], k[3], k[4]))
P
rm -rf $out/p
