#!/bin/bash
# GPU box: parity smoke + bench line (kernel times) - tools/quick.sh <tag>
tag=${1:-q}
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or oracle or full_size or graph" 2>&1 | tail -2
python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows > gpurun_out/$tag.json 2> gpurun_out/$tag.err
python3 - <<P
import json
d = json.load(open("gpurun_out/$tag.json"))
print("$tag", d["value"], "fps; latency", d["latency_ms_b1"], "pipeline", d["latency_and_pipeline"].get("pipeline_frames_per_s"), "|", " ".join(f'{k["name"].split()[0].replace("enc.","")}={k["us"]}' for k in d["kernels"]))
P
