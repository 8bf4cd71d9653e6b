#!/bin/bash
# Kernel timeline of the bench loop (GPU box): tools/timeline.sh <tag> [streams...] -> gpurun_out/<tag>/trace_s<N>.csv
tag=${1:-tl}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in "${@:-3}"; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/tr$s -- python3 bench.py --steps 60 --warmup 10 --cpu-seconds 0 --no-other-rows --streams $s --kernel-reps 1 > $out/bench_s$s.json 2> $out/err_s$s.txt
  f=$(find $out/tr$s -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && cp $f $out/trace_s$s.csv
  rm -rf $out/tr$s
done
ls -la $out
