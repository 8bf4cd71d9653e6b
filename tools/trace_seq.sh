#!/bin/bash
# GPU box: the launch sequence of a command's last ~N kernels: tools/trace_seq.sh <tag> <n> -- <python args...>
tag=$1; n=$2; shift; shift; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BENCH_N=1 BENCH_WARM=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/p -- python3 "$@" > /dev/null 2>&1
f=$(find $out/p -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$n" <<'P'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-int(sys.argv[2]):]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print("%8.1f us  gap %6.1f  %-44s grid %s x %s x %s" % ((e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, name, r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")))
    prev = e
P
rm -rf $out/p
