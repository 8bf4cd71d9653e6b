#!/bin/bash
# GPU box: rocprofv3 kernel stats of the E-RAFT forward at batch $1 (default 1) -> gpurun_out/<tag>/eraft_b<batch>_kernel_stats.csv
b=${1:-1}; tag=${2:-eraft}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 tools/bench_eraft.py $b 2>/dev/null | tail -1
BENCH_N=4 BENCH_WARM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -- python3 tools/bench_eraft.py $b > /dev/null 2>&1
find $out/p -name "*kernel_stats.csv" -exec cp {} $out/eraft_b${b}_kernel_stats.csv \;
rm -rf $out/p
python3 - <<P
import csv
rows = list(csv.DictReader(open("$out/eraft_b${b}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per forward: %.2f ms" % (tot / 5 / 1e6))
for r in rows[:22]:
    print("%6.1f%% %7.1f us x %4d/fwd  %s" % (float(r["Percentage"]), float(r["AverageNs"]) / 1e3, int(r["Calls"]) / 5, r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]))
P
