#!/bin/bash
# GPU box: HBM-side bytes (FETCH_SIZE x 2, WRITE_SIZE; MI355X_MICROARCH.md's gfx950 corrections as tools/pmc_traffic.py applies them) per
# launch of the first two encoder layers - the fused launch (conv_enc12.hip) against the two it replaces - at 1280x720, one frame pair
# per launch.  usage: tools/pmc_e12.sh <tag>
tag=${1:-pmc_e12}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for fuse in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    EEM_FUSE12=$fuse timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/f$fuse/$c -- python3 bench.py --steps 5 --warmup 2 --long-steps 0 --preheat 4 --cpu-seconds 0 --no-graph --streams 1 --coalesce 1 --frames-in-flight 4 --kernel-reps 3 --no-other-rows --no-side-rows > /dev/null 2> $out/f${fuse}_$c.err
  done
done
python3 - <<P
import csv, glob, collections
for fuse in (1, 0):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$out/f%d/%s/**/*_counter_collection.csv" % (fuse, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40]][c].append(float(r["Counter_Value"]))
    print("EEM_FUSE12=%d" % fuse)
    tot = 0
    for k, cs in sorted(agg.items()):
        if not any(s in k for s in ("enc12", "enc1_kernel", "wino4_kernel<16", "bx3", "wino")):
            continue
        fe = sum(cs["FETCH_SIZE"]) / max(len(cs["FETCH_SIZE"]), 1) * 1024 * 2
        wr = sum(cs["WRITE_SIZE"]) / max(len(cs["WRITE_SIZE"]), 1) * 1024
        tot += fe + wr
        print("   %-42s fetch %7.1f MB  write %7.1f MB  sum %7.1f MB  (%d launches)" % (k, fe / 1e6, wr / 1e6, (fe + wr) / 1e6, len(cs["FETCH_SIZE"])))
    print("   encoder total %.1f MB" % (tot / 1e6))
P
