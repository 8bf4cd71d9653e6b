#!/bin/bash
# GPU box: voxelizer parity tests, timing, per-kernel times (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -m pytest tests -q -m gpu -x -k "voxel or hrem or HREM or data_rows" 2>&1 | tail -5
python3 tools/voxel_bench.py 200000 2000000 2>&1 | grep -v amdgpu.ids
echo direct; EEM_VOX_DIRECT=1 python3 tools/voxel_bench.py 2000000 2>&1 | grep -v amdgpu.ids
for n in 2000000 200000; do
rm -rf gpurun_out/voxprof_$n
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/voxprof_$n -- python3 tools/voxel_bench.py $n > gpurun_out/voxprof_$n.log 2>&1
python3 - <<P
import csv,glob
print("n=$n")
for f in glob.glob("gpurun_out/voxprof_$n/**/*kernel_trace.csv", recursive=True):
    rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
    for r in rows[-11:]: print("%-40s %8.1f us" % (r["Kernel_Name"].replace("(anonymous namespace)::","")[:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
P
done
