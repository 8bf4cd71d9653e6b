#!/bin/bash
# Compact per-kernel table from one bench.py run (GPU box): tools/kernels.sh [tag] [extra bench args]
tag=${1:-k}; shift
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-other-rows "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python3 - <<P
import json
d=json.loads(open("gpurun_out/bench_$tag.json").read().strip().splitlines()[-1])
print("fps", d["value"], "ms/step", d["ms_per_step"], "sum_us", d["schedule_sum_us"], "enc_tflops", d["encoder_tflops_in_kernel"])
print("roof", {k: d["roofline"][k] for k in ("kernel", "kernel_us", "achieved", "frac")})
for k in d["kernels"][:9]: print("%-32s %7.2f us %7.2f TF %7.1f GB/s" % (k["name"], k["us"], k["tflops"], k["gbs"]))
print("tail_us", round(sum(k["us"] for k in d["kernels"][8:]), 1), "err", d.get("flow_max_abs_err_vs_oracle"))
P
