#!/bin/bash
# GPU box: HBM-side MB per FRAME and per encoder kernel (FETCH_SIZE x 2 + WRITE_SIZE) for launches of <co> frames: tools/pmc_batch.sh <tag> <co> [env...]
tag=$1; co=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 bench.py --steps 20 --warmup 10 --long-steps 0 --preheat 10 --cpu-seconds 0 --no-graph --streams 1 --coalesce $co --kernel-reps 2 --no-other-rows --no-side-rows > /dev/null 2> $out/$c.err
done
python3 - <<P
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$out/%s/**/*_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:44]][c].append(float(r["Counter_Value"]))
tot = 0
print("$* : MB per frame at $co frames per launch")
for k, cs in sorted(agg.items()):
    if not any(s in k for s in ("enc12", "enc1_kernel", "wino", "bx3_s2", "tail", "s2_kernel")): continue
    fe = sum(cs["FETCH_SIZE"]) / max(len(cs["FETCH_SIZE"]), 1) * 2048 / $co
    wr = sum(cs["WRITE_SIZE"]) / max(len(cs["WRITE_SIZE"]), 1) * 1024 / $co
    tot += fe + wr
    print("   %-46s fetch %6.1f  write %6.1f  sum %6.1f   (%d launches, fetch min %.1f max %.1f)" % (k, fe / 1e6, wr / 1e6, (fe + wr) / 1e6, len(cs["FETCH_SIZE"]), min(cs["FETCH_SIZE"]) * 2048 / $co / 1e6, max(cs["FETCH_SIZE"]) * 2048 / $co / 1e6))
print("   total %.1f MB per frame" % (tot / 1e6))
P
