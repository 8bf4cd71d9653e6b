#!/bin/bash
# GPU box: kernel timeline of the timed loop with four frames in flight (rocprofv3 --kernel-trace) -> gpurun_out/trace4/timeline.txt
out=gpurun_out/trace4; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 bench.py --steps 60 --warmup 20 --cpu-seconds 0 --no-other-rows --no-side-rows --preheat 20 "$@" > $out/bench.json 2> $out/err.txt
f=$(find $out/raw -name "*kernel_trace.csv" | head -1)
python3 tools/trace4.py "$f" > $out/timeline.txt
rm -rf $out/raw
tail -60 $out/timeline.txt
