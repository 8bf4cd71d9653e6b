#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json (read by bench.py).

HBM bytes per launch = FETCH_SIZE(KB) * 1024 * 2 + WRITE_SIZE(KB) * 1024: on gfx950 FETCH_SIZE reports half
of the bytes of a wide (16 B/lane) coalesced read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact
for 16-B-per-lane stores... our stores are dword-per-lane 128-B segments, counted as reported.
Usage: python tools/pmc_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/ subdirs> <out.json>
"""
import collections
import csv
import glob
import json
import re
import sys

ENC_NAMES = {("5", "16"): ["enc.pconv1_1 5->16 s2 +pad"], ("16", "16"): ["enc.pconv1_2 16->16"],
             ("16", "32"): ["enc.pconv2_1 16->32 s2"], ("32", "32"): ["enc.pconv2_2 32->32", "enc.pconv2_3 32->32"],
             ("32", "64"): ["enc.pconv3_1 32->64 s2"], ("64", "64"): ["enc.pconv3_2 64->64", "enc.pconv3_3 64->64"]}


def load(d, counter):
    f = glob.glob(f"{d}/{counter}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        m = re.search(r"(?:enc_conv2?|s2)_kernel<(\d+), (\d+)[,>]", k)
        if m:
            agg[(m.group(1), m.group(2))].append(float(r["Counter_Value"]))
            continue
        m = re.search(r"wino(?:32|4)?_kernel<(\d+),", k)        # Winograd C -> C layers (either form)
        if m:
            agg[(m.group(1), m.group(1))].append(float(r["Counter_Value"]))
        elif "enc1_kernel" in k:                                 # pconv1_1 with 16-byte DMA
            agg[("5", "16")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def csrc_sha():
    """sha256 over the kernel sources: bench.py quotes these numbers only for the sources they were measured on."""
    import hashlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eemflow_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    d, out = sys.argv[1], sys.argv[2]
    fetch, write = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    res = {}
    for key, names in ENC_NAMES.items():
        if key in fetch and key in write:
            hbm = fetch[key] * 1024 * 2 + write[key] * 1024
            for n in names:
                res[n] = {"hbm_bytes": round(hbm), "fetch_size_kb_raw": round(fetch[key], 1),
                          "write_size_kb_raw": round(write[key], 1), "fetch_correction": 2.0}
    # the workload the passes were collected on (tools/profile_gpu.sh runs bench.py's default shape); bench.py only
    # quotes these numbers when it runs that shape
    fpl = int(sys.argv[6]) if len(sys.argv) > 6 else 1           # frames per launch: the coalescing width of the timed loop's chains
    res["_workload"] = {"height": int(sys.argv[3]) if len(sys.argv) > 3 else 720, "width": int(sys.argv[4]) if len(sys.argv) > 4 else 1280,
                        "batch": int(sys.argv[5]) if len(sys.argv) > 5 else 1, "frames_per_launch": fpl}
    res["_encoder_mb_per_frame"] = round(sum(v["hbm_bytes"] for k, v in res.items() if k.startswith("enc.")) / fpl / 1e6, 1)
    import datetime
    import os
    res["_meta"] = {"csrc_sha": csrc_sha(), "commit": os.environ.get("EEM_COMMIT", "unknown"),
                    "date": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ"),
                    "command": "bench.py --steps 20 --warmup 10 --preheat 10 --long-steps 0 --no-graph --streams 1 --coalesce <frames_per_launch> --kernel-reps 2 "
                               "under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
