#!/bin/bash
# GPU box: what each launch of the frame costs BESIDE the others, four frames in flight - frames/s with that launch skipped
# (EEM_SKIP_KERNELS; the flow is garbage in those runs).  tools/marginal.sh [bench args]
# the switches used here exist in the diagnostic build only: EEM_BUILD_TAG=diag EEM_EXTRA_FLAGS=-DEEM_DIAG python -m eemflow_amd.build (before gpurun)
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
run() {
  EEM_SKIP_KERNELS="$1" python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows --no-side-rows --long-steps 0 "${@:2}" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
}
base=$(run "" "$@"); echo "nothing skipped: $base"
b=$(echo $base | cut -d" " -f2)
for k in "enc.pconv1_1" "enc.pconv1_2" "enc.pconv2_1" "enc.pconv2_2" "enc.pconv2_3" "enc.pconv3_1" "enc.pconv3_2" "enc.pconv3_3" "tail head" "dec." "tail up"; do
  r=$(run "$k" "$@"); t=$(echo $r | cut -d" " -f2)
  [ -n "$t" ] || { echo "skip $k: the run printed no line"; continue; }      # (every launch skipped at once leaves bench.py nothing to time)
  python3 -c "print('skip %-16s %s frames/s  -> marginal cost %.1f us of %.1f' % ('$k', '$r'.split()[0], ($b - $t) * 1e3, $b * 1e3))"
done
