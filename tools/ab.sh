#!/bin/bash
# A/B of an environment switch on the bench (GPU box): tools/ab.sh <tag> VAR  -> bench lines with VAR unset / VAR=1
tag=${1:-ab}; var=${2:-EEM_NO_STAGGER}
out=gpurun_out/$tag; mkdir -p $out
for v in "" 1; do
  if [ -z "$v" ]; then unset $var; n=default; else export $var=$v; n=${var}_$v; fi
  timeout 300 python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows > $out/bench_$n.json 2> $out/err_$n.txt
  timeout 300 python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows --streams 1 > $out/bench_s1_$n.json 2>> $out/err_$n.txt
  python3 - <<P
import json
for f in ("$out/bench_$n.json", "$out/bench_s1_$n.json"):
    try:
        d = json.load(open(f))
        print("$n", d["config"]["streams_per_gpu"], "streams:", d["value"], "fps;", " ".join(f'{k["name"].split()[0]}={k["us"]}' for k in d["kernels"]))
    except Exception as e:
        print("$n", f, "failed", e)
P
done
