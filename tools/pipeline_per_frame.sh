#!/bin/bash
# GPU box: tools/pipeline_trace.py (events -> voxelize_many -> forward_many -> flow_error_many, ten samples per call, two chains) - frames/s, the
# kernel time each launch type costs per frame (rocprofv3 --kernel-trace --stats; both chains' launches overlap, so the sum exceeds the wall time),
# and the voxelizer's band size.  tools/pipeline_per_frame.sh <tag>  -> gpurun_out/<tag>_pipeline_per_frame.txt
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_pipeline_per_frame.txt
{
  for m in all noerr; do PIPE_MODE=$m python3 tools/pipeline_trace.py 60 2>/dev/null | tail -1; done
  for b in 4608 2304 18432; do echo "EEM_VOX_BAND_FLOATS=$b: $(EEM_VOX_BAND_FLOATS=$b python3 tools/pipeline_trace.py 60 2>/dev/null | tail -1)"; done
} > $out
mkdir -p gpurun_out/${tag}_pipe
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_pipe/p -- python3 tools/pipeline_trace.py 40 > /dev/null 2>&1
find gpurun_out/${tag}_pipe/p -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_pipe/kernel_stats.csv \;
rm -rf gpurun_out/${tag}_pipe/p
python3 - gpurun_out/${tag}_pipe/kernel_stats.csv >> $out <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frames = 46 * 10                                     # 6 warm-up + 40 timed calls of ten samples
print("kernel time per frame (us), both chains, under rocprofv3:")
for r in rows[:22]:
    print("%7.2f  %6s calls  %s" % (float(r['TotalDurationNs']) / 1e3 / frames, r['Calls'], r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:80]))
P
cat $out
