#!/bin/bash
# GPU box: rocprofv3 kernel stats of the E-RAFT training step (640x480 x 12 iterations x batch 4) -> gpurun_out/<tag>/eraft_train_step_kernel_stats.csv
tag=${1:-ertrain}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 tools/bench_eraft_train.py 2>/dev/null | tail -1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -- python3 tools/bench_eraft_train.py > /dev/null 2>&1
find $out/p -name "*kernel_stats.csv" -exec cp {} $out/eraft_train_step_kernel_stats.csv \;
rm -rf $out/p
python3 - <<P
import csv
rows = list(csv.DictReader(open("$out/eraft_train_step_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.1f ms over %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:30]:
    print("%6.1f%% %8.1f us x %5d  %s" % (float(r["Percentage"]), float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:110]))
P
