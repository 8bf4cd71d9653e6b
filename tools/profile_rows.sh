#!/bin/bash
# GPU box: rocprofv3 kernel-stats summaries of the side rows (training step, voxelizer) -> gpurun_out/<tag>/
tag=${1:-rows}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 tools/bench_train.py > $out/train_346x260_b32.txt 2>/dev/null
python3 tools/bench_train.py 8 720 1280 > $out/train_1280x720_b8.txt 2>/dev/null
python3 tools/voxel_bench.py 200000 2000000 2>/dev/null > $out/voxel_bench.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/train -- python3 tools/bench_train.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/voxel -- python3 tools/voxel_bench.py 2000000 > /dev/null 2>&1
find $out/train -name "*kernel_stats.csv" -exec cp {} $out/train_kernel_stats.csv \;
find $out/voxel -name "*kernel_stats.csv" -exec cp {} $out/voxel_kernel_stats.csv \;
cat $out/*.txt
python3 tools/bench_eraft.py 1 > $out/eraft_b1.txt 2>/dev/null
python3 tools/bench_eraft.py 4 > $out/eraft_b4.txt 2>/dev/null
python3 tools/bench_plus.py 2>/dev/null | tail -1 > $out/eemflow_plus.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/eraft -- python3 tools/bench_eraft.py 4 > /dev/null 2>&1
find $out/eraft -name "*kernel_stats.csv" -exec cp {} $out/eraft_b4_kernel_stats.csv \;
cat $out/eraft_b1.txt $out/eraft_b4.txt $out/eemflow_plus.txt
