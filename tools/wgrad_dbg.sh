#!/bin/bash
# GPU box (diagnostic build): tools/wgrad_bench.py with parts of the weight-gradient kernels removed: tools/wgrad_dbg.sh <shape filter> "<EEM_WG_DBG values>"
mkdir -p gpurun_out
export EEM_LIB_PATH=$PWD/eemflow_amd/libeemflow_hip_diag.so
for d in $2; do
  echo "=== EEM_WG_DBG=$d (bit 0: no compute phase, 1: no atomics, 2: no DMA, 3: no operand reads, 4: no MFMAs)"
  EEM_WG_DBG=$d timeout 300 python tools/wgrad_bench.py 20 "$1" 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6}'
done
