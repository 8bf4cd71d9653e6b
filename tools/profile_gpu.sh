#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats of the timed loop's launch configuration,
# PMC passes (separate runs, no trace domains).  Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -u
tag=${1:-prof}
co=${2:-10}                                    # coalescing width of the timed loop (bench.py --coalesce)
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
# one chain at a time (--streams 1): every kernel's average then belongs to the batch-$co launch alone, as the bench line's HIP events do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 300 --warmup 30 --long-steps 0 --cpu-seconds 0 --no-other-rows --no-side-rows --streams 1 --coalesce $co > $out/bench_under_rocprof.json 2> $out/stats.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  t=$(echo $pass | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $out/pmc/$t -- python3 bench.py --steps 20 --warmup 10 --preheat 10 --long-steps 0 --cpu-seconds 0 --no-graph --streams 1 --coalesce $co --kernel-reps 2 --no-other-rows --no-side-rows > /dev/null 2> $out/pmc_$t.err
done
EEM_COMMIT=${EEM_COMMIT:-unknown} python3 tools/pmc_traffic.py $out/pmc $out/pmc_traffic.json 720 1280 1 $co > /dev/null
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
ls $out
