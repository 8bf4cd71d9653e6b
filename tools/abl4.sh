#!/bin/bash
# GPU box: ablation timings of wino4_kernel (results are wrong in the ablated builds - timing only): tools/abl4.sh <C> <abl...>
C=$1; shift
for abl in "$@"; do
  touch eemflow_amd/csrc/conv_wino4.hip
  EEM_EXTRA_FLAGS="-DEEM_STAMPS=$C -DEEM_W4_ABL=$abl" python3 -c "from eemflow_amd.build import build_library; build_library(verbose=False)"
  echo "=== ablation $abl (1: no DMA in the loop, 2: no MFMA, 3: no row transform, 4: no weight reads, 0: none)"
  python3 tools/stamps4.py 2>&1 | grep -E "k-step  [2-5]|output|prologue" | head -14
done
