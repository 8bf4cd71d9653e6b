#!/bin/bash
# GPU box: headline / latency for each F(4x4) layer mask
for m in "$@"; do
  EEM_WINO4_LAYERS=$m python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows > gpurun_out/mask$m.json 2> gpurun_out/mask$m.err || tail -3 gpurun_out/mask$m.err
  python3 - <<P
import json
d = json.load(open("gpurun_out/mask$m.json"))
print("mask $m:", d["value"], "fps; latency", d["latency_ms_b1"], "pipeline", d["latency_and_pipeline"].get("pipeline_frames_per_s"), "b4x2", d["latency_and_pipeline"].get("batch4_two_in_flight_frames_per_s"), "err", d.get("flow_max_abs_err_vs_oracle"))
print("   in-flight grids:", " ".join(k["name"].split()[0].replace("enc.","")+"="+str(k["us"]) for k in d["kernels"][:8]))
print("   single-frame:   ", {k.replace("enc.",""): v for k, v in list(d["roofline_single_frame_launch"]["kernels_us"].items())[:8]})
P
done
