#!/bin/bash
# GPU box: the round-6 evidence beyond tools/profile_gpu.sh (bench line, kernel stats, PMC passes): weight-gradient kernels alone and with
# parts removed, training-step timelines, E-RAFT / EEMFlow+ kernel stats, side rows.  tools/profile_round6.sh <tag>  (diagnostic build needed)
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# 1. weight-gradient kernels alone: ring / tile-bx3 / tile-fp32 (release build)
python3 tools/wgrad_bench.py 20 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_wgrad_bench.txt
# 2. the same with the compute phase / the atomics / both removed (diagnostic build): what overlaps what
( export EEM_LIB_PATH=$PWD/eemflow_amd/libeemflow_hip_diag.so
  for d in 0 1 2 3; do
    echo "=== EEM_WG_DBG=$d (bit 0: no compute phase (operand reads + MFMAs), bit 1: no atomics); columns: ring / tile-bx3 / tile-fp32, us alone"
    EEM_WG_DBG=$d python3 tools/wgrad_bench.py 20 C 2>&1 | grep -v amdgpu.ids | awk '{printf "%-28s %9s %11s %12s\n", $1" "$2" "$3" "$4, $(NF-5), $(NF-3), $(NF-1)}'
  done ) > gpurun_out/${tag}_wgrad_nomfma.txt 2>&1
# ... and inside the training step (beside the data-gradient stream): per-kernel averages
tools/wgrad_nomfma.sh >> gpurun_out/${tag}_wgrad_nomfma.txt 2>&1
# 3. training-step timelines with queues: the round's sources, and the weight-gradient forms it replaced
bash tools/step_timeline.sh ${tag}_train_tl pad4_kernel tools/bench_train.py > /dev/null 2>&1
cp gpurun_out/${tag}_train_tl/timeline.txt gpurun_out/${tag}_train_timeline.txt
EEM_NO_WGRAD_BX3=1 EEM_NO_WGRAD_TAIL=1 EEM_TRAIN_LATE_STATS=1 EEM_UPBWD_THREADS=1 bash tools/step_timeline.sh ${tag}_train_tl5 pad4_kernel tools/bench_train.py > /dev/null 2>&1
cp gpurun_out/${tag}_train_tl5/timeline.txt gpurun_out/${tag}_train_timeline_round5_forms.txt
bash tools/step_timeline.sh ${tag}_train_tl_c4 pad4_kernel tools/bench_train.py 8 720 1280 > /dev/null 2>&1
cp gpurun_out/${tag}_train_tl_c4/timeline.txt gpurun_out/${tag}_train_timeline_1280x720_b8.txt
# 4. side rows + kernel stats
tools/profile_rows.sh ${tag}_rows > gpurun_out/${tag}_rows.txt 2>&1
tools/prof_eraft_train.sh ${tag}_ertrain > gpurun_out/${tag}_ertrain.txt 2>&1
tools/prof_plus.sh ${tag}_plus > gpurun_out/${tag}_plus.txt 2>&1
for r in default none all; do
  if [ $r = default ]; then unset EEM_WGRAD_RING; else export EEM_WGRAD_RING=$r; fi
  echo "EEM_WGRAD_RING=$r: $(python3 tools/bench_eraft_train.py 2>/dev/null | tail -1)"
done >> gpurun_out/${tag}_ertrain.txt
unset EEM_WGRAD_RING
echo "EEM_NO_WGRAD_BX3=1: $(EEM_NO_WGRAD_BX3=1 python3 tools/bench_eraft_train.py 2>/dev/null | tail -1)" >> gpurun_out/${tag}_ertrain.txt
echo "EEM_NO_DGRAD_S2W=1: $(EEM_NO_DGRAD_S2W=1 python3 tools/bench_eraft_train.py 2>/dev/null | tail -1)" >> gpurun_out/${tag}_ertrain.txt
echo "EEM_NO_WGRAD_FEW=1: $(EEM_NO_WGRAD_FEW=1 python3 tools/bench_eraft_train.py 2>/dev/null | tail -1)" >> gpurun_out/${tag}_ertrain.txt
ls gpurun_out | grep ${tag}_
# 5. second half of the round: EEMFlow+'s Winograd kernel (timelines with and without it, the side-stream form, the frame rates over 40
# forwards), the training step's prologue / first-layer switches, E-RAFT's encoder convs on the Winograd kernel
tools/plus_timeline.sh ${tag}_plus_tl > /dev/null 2>&1
cp gpurun_out/${tag}_plus_tl/timeline.txt gpurun_out/${tag}_eemflow_plus_timeline.txt
EEM_NO_WNC=1 tools/plus_timeline.sh ${tag}_plus_tl_nownc > /dev/null 2>&1
cp gpurun_out/${tag}_plus_tl_nownc/timeline.txt gpurun_out/${tag}_eemflow_plus_timeline_no_wnc.txt
{
  for i in 1 2; do
    echo "default:                   $(python3 tools/bench_plus.py 2>/dev/null | tail -1)"
    echo "EEM_NO_WNC=1:              $(EEM_NO_WNC=1 python3 tools/bench_plus.py 2>/dev/null | tail -1)"
    echo "EEM_PLUS_SIDE=1:           $(EEM_PLUS_SIDE=1 python3 tools/bench_plus.py 2>/dev/null | tail -1)"
    echo "EEM_PLUS_WNC_MINPX_JOBS=30000 (level 3 off the Winograd kernel): $(EEM_PLUS_WNC_MINPX_JOBS=30000 python3 tools/bench_plus.py 2>/dev/null | tail -1)"
    echo "EEM_FEWOUT_SMALL_BLOCKS=0: $(EEM_FEWOUT_SMALL_BLOCKS=0 python3 tools/bench_plus.py 2>/dev/null | tail -1)"
  done
  echo "batch 4: $(python3 tools/bench_plus.py 4 2>/dev/null | tail -1)"
} > gpurun_out/${tag}_eemflow_plus_switches.txt 2>&1
{
  export EEM_BT_N=300
  for i in 1 2; do
    echo "default:                                    $(python3 tools/bench_train.py 2>/dev/null)"
    echo "EEM_WGRAD_LAST_SIDE=1:                      $(EEM_WGRAD_LAST_SIDE=1 python3 tools/bench_train.py 2>/dev/null)"
    echo "EEM_TRAIN_SIDE_PREP=0:                      $(EEM_TRAIN_SIDE_PREP=0 python3 tools/bench_train.py 2>/dev/null)"
    echo "EEM_WGRAD_LAST_SIDE=1 EEM_TRAIN_SIDE_PREP=0: $(EEM_WGRAD_LAST_SIDE=1 EEM_TRAIN_SIDE_PREP=0 python3 tools/bench_train.py 2>/dev/null)"
  done
  echo "1280x720 batch 8 default:                   $(python3 tools/bench_train.py 8 720 1280 2>/dev/null)"
  echo "1280x720 batch 8 both off:                  $(EEM_WGRAD_LAST_SIDE=1 EEM_TRAIN_SIDE_PREP=0 python3 tools/bench_train.py 8 720 1280 2>/dev/null)"
  python3 tools/train_host_time.py 2>/dev/null | tail -1
  unset EEM_BT_N
} > gpurun_out/${tag}_train_switches.txt 2>&1
{
  for b in 1 4; do for i in 1 2; do
    echo "default      b$b: $(BENCH_N=20 python3 tools/bench_eraft.py $b 2>/dev/null | tail -1)"
    echo "NO_WNC=1     b$b: $(EEM_ERAFT_NO_WNC=1 BENCH_N=20 python3 tools/bench_eraft.py $b 2>/dev/null | tail -1)"
  done; done
} > gpurun_out/${tag}_eraft_wnc.txt 2>&1
ls gpurun_out | grep ${tag}_
# 6. the event-input pipeline alone (bench.py's coalesced chain): frames/s, per-frame kernel time by launch, voxelizer band sizes
tools/pipeline_per_frame.sh ${tag} > /dev/null 2>&1
