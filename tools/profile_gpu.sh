#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats of the same
# command, PMC passes (separate runs, no trace domains).  Everything lands in gpurun_out/<tag>/.
set -u
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BENCH="bench.py --steps 300 --warmup 30"
timeout 600 python3 $BENCH > $out/bench.json 2> $out/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $BENCH --cpu-seconds 0 --no-other-rows --no-side-rows --streams 1 --frames-in-flight 4 > $out/bench_under_rocprof.json 2> $out/stats.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  t=$(echo $pass | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $out/pmc/$t -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-graph --streams 1 --frames-in-flight 4 --kernel-reps 3 --no-other-rows --no-side-rows > /dev/null 2> $out/pmc_$t.err
done
EEM_COMMIT=${EEM_COMMIT:-unknown} python3 tools/pmc_traffic.py $out/pmc $out/pmc_traffic.json > /dev/null
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
ls $out
