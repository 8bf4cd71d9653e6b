#!/bin/bash
# GPU box: F(4x4) kernel check - Winograd-vs-direct parity, then the per-kernel table (single-frame launch configuration)
tag=${1:-w4}
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd or golden or full_size" 2>&1 | tail -3
python3 bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-other-rows > gpurun_out/$tag.json 2> gpurun_out/$tag.err || tail -5 gpurun_out/$tag.err
python3 - <<P
import json
d = json.load(open("gpurun_out/$tag.json"))
print(d["value"], "fps; latency", d["latency_ms_b1"], "|", " ".join(k["name"].split()[0].replace("enc.","")+"="+str(k["us"]) for k in d["kernels"][:8]))
print("single-frame grids:", {k.replace("enc.",""): v for k, v in list(d["roofline_single_frame_launch"]["kernels_us"].items())[:8]})
P
