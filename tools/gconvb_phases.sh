#!/bin/bash
# GPU box, diagnostic build: gconvb.hip's launches in the E-RAFT forward (batch $1, default 4) with pieces switched off (EEM_GB_DBG bits:
# 1 no MFMAs, 2 no weight loads in the k-loop, 4 no A-fragment reads in the k-loop, 8 staging waves idle, 16 plain epilogue)
b=${1:-4}
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in ${GB_SET:-0 1 2 4 8 16 6 14 7 15 31}; do
  rm -rf gpurun_out/gbdbg
  EEM_GB_DBG=$d BENCH_N=3 BENCH_WARM=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gbdbg -- python3 tools/bench_eraft.py $b > /dev/null 2>&1
  python3 - $d <<'P'
import csv, glob, sys
tot = {}
for f in glob.glob("gpurun_out/gbdbg/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gconvb_kernel" in r["Name"]:
            k = r["Name"].split("gconvb_kernel")[1].split(">")[0] + ">"
            tot[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
print("EEM_GB_DBG=%-2s " % sys.argv[1] + "  ".join("%s x%d %.1f" % (k, c, a) for k, (c, a) in sorted(tot.items())) + "   sum/fwd %.0f us" % (sum(c * a for c, a in tot.values()) / 5))
P
done
rm -rf gpurun_out/gbdbg
