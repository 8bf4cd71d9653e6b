#!/bin/bash
# GPU box: per-kernel times of the gconvb launches in an E-RAFT batch-4 forward for each diagnostic build (tools/gb_abl.sh)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in "" "$@"; do
  lib=eemflow_amd/libeemflow_hip${n:+_abl$n}.so
  out=gpurun_out/gbabl_$n; mkdir -p $out
  EEM_LIB_PATH=$PWD/$lib BENCH_N=3 BENCH_WARM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -- python3 tools/bench_eraft.py 4 > /dev/null 2>&1
  f=$(find $out/p -name "*kernel_stats.csv" | head -1)
  echo "== abl '$n'"
  python3 - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gconvb_kernel" in r["Name"]:
        print("   %-24s calls %4d avg %8.1f us" % (r["Name"].split("gconvb_kernel")[1][:8], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
P
  rm -rf $out
done
