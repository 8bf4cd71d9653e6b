#!/bin/bash
# build container: diagnostic builds of the library with parts of gconvb_kernel switched off (GB_ABL bits: 1 weight loads, 2 A reads,
# 4 conversion, 8 staging loads) -> eemflow_amd/libeemflow_hip_abl<N>.so (git-ignored), selected by EEM_LIB_PATH
cd "$(dirname "$0")/.."
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fvisibility=hidden -fvisibility-inlines-hidden -DGB_ABL=$n -c eemflow_amd/csrc/gconvb.hip -o /tmp/gconvb_abl$n.o 2>/dev/null
  objs=$(ls eemflow_amd/csrc/build/*.o | grep -v "/gconvb.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o eemflow_amd/libeemflow_hip_abl$n.so $objs /tmp/gconvb_abl$n.o && echo built abl$n
done
