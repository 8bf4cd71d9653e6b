#!/bin/bash
# GPU box: per-kernel durations of the voxelizer at n events (default 200000), both normalisation forms
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
n=${1:-200000}
for tp in 0 8; do
export EEM_VOX_TWOPASS=$tp
rm -rf gpurun_out/voxtrace_$tp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/voxtrace_$tp -- python3 tools/voxel_bench.py $n > gpurun_out/voxtrace_$tp.log 2>&1
grep "n=" gpurun_out/voxtrace_$tp.log
python3 - <<P
import csv,glob
print("EEM_VOX_TWOPASS=$tp n=$n")
for f in glob.glob("gpurun_out/voxtrace_$tp/**/*kernel_trace.csv", recursive=True):
    rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
    rows=[r for r in rows if "vox" in r["Kernel_Name"]]
    t0=int(rows[-12]["Start_Timestamp"])
    for r in rows[-12:]: print("%-28s start %8.1f  dur %8.1f us  grid %s" % (r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][-28:], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size_X", r.get("Grid_Size",""))))
P
done
