#!/bin/bash
# GPU box: rocprofv3 kernel stats of the EEMFlow+ forward at 1280x720 batch 1 -> gpurun_out/<tag>/eemflow_plus_kernel_stats.csv
tag=${1:-plus}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 tools/bench_plus.py 2>/dev/null | tail -1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -- python3 tools/bench_plus.py > /dev/null 2>&1
find $out/p -name "*kernel_stats.csv" -exec cp {} $out/eemflow_plus_kernel_stats.csv \;
find $out/p -name "*kernel_trace.csv" -exec cp {} $out/trace.csv \;
python3 - <<P
import csv
rows = list(csv.DictReader(open("$out/eemflow_plus_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("kernel time total %.2f ms over %d launches" % (tot / 1e6, calls))
for r in rows[:24]:
    print("%6.1f%% %7.1f us x %5d  %s" % (float(r["Percentage"]), float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:100]))
P
rm -rf $out/p
