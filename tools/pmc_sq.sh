#!/bin/bash
# SQ counters per kernel (GPU box): tools/pmc_sq.sh <tag> ; prints per-kernel averages of each counter
tag=${1:-sq}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PASSES=${PMC_PASSES:-}
if [ -n "$PASSES" ]; then IFS=";" read -ra PL <<< "$PASSES"; else PL=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_WAVE32_LDS SQ_WAVES SQ_INSTS_SMEM"); fi
for pass in "${PL[@]}"; do
  t=$(echo $pass | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $out/$t -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-graph --streams 1 --kernel-reps 2 --no-other-rows > /dev/null 2> $out/$t.err
done
python3 - <<P
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv" not in k and "wino" not in k: continue
        m = re.search(r"(\w+_kernel)<([^>]*)>", k)
        name = (m.group(1) + "<" + m.group(2)[:28] + ">") if m else k[:40]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    print(name)
    print("   " + "  ".join(f"{c.replace('SQ_','')}={sum(v)/len(v):.3g}" for c, v in sorted(cs.items())))
P
