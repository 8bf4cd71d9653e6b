#!/bin/bash
# GPU box: training parity tests, step timing (new / old encoder wgrad), per-kernel trace of the training step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -4
python3 tools/bench_train.py 2>&1 | grep -v amdgpu.ids
python3 tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids
EEM_NO_WGRAD_ENC=1 python3 tools/bench_train.py 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/trprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trprof -- python3 tools/bench_train.py > gpurun_out/trprof.log 2>&1
python3 - <<P
import csv,glob
for f in glob.glob("gpurun_out/trprof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:22]: print("%-90s %6s %9.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
P
