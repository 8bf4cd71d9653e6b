#!/bin/bash
# GPU box, diagnostic build: the weight-gradient kernels with and without their MFMAs (EEM_WG_DBG=1) - per-kernel averages of the training step
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 1 2 3; do
  rm -rf gpurun_out/wgdbg$d
  EEM_WG_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wgdbg$d -- python3 tools/bench_train.py > gpurun_out/wgdbg$d.log 2>&1
  echo "EEM_WG_DBG=$d: $(grep 'train step' gpurun_out/wgdbg$d.log)"
  python3 - <<P
import csv,glob
for f in glob.glob("gpurun_out/wgdbg$d/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f))):
        if "wgrad_enc" in r["Name"]: print("   %-70s %4s x %8.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:70], r["Calls"], float(r["AverageNs"])/1e3))
P
done
