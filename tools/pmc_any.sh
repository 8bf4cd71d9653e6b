#!/bin/bash
# SQ counters per kernel for an arbitrary python tool (GPU box): tools/pmc_any.sh <tag> <script.py> [args]; prints averages per (kernel, grid)
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  t=$(echo $pass | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $out/$t -- python3 "$@" > /dev/null 2> $out/$t.err
done
python3 - <<P
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        name = (k[:34], r["Grid_Size"])
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0])))
for name, cs in rows[:12]:
    w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"]) if cs.get("SQ_WAVES") else 1
    print(name, "launches", len(cs.get("SQ_WAVES", [])))
    print("   " + "  ".join(f"{c.replace('SQ_','')}/wave={sum(v)/len(v)/max(w,1):.4g}" for c, v in sorted(cs.items())))
P
