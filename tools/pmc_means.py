"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files: python3 tools/pmc_means.py <pmc dir> <out prefix>
(<pmc dir>/<PASS>/**/counter_collection.csv -> <out prefix><PASS>.csv with Kernel_Name, Counter_Name, Dispatches, Mean_Value)."""
import csv
import glob
import os
import sys
from collections import defaultdict

src, prefix = sys.argv[1], sys.argv[2]
for d in sorted(os.listdir(src)):
    files = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"], r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    with open(prefix + d + ".csv", "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean_Value"])
        for (kn, cn), (n, s) in sorted(acc.items()):
            w.writerow([kn, cn, n, round(s / n, 3)])
    print(prefix + d + ".csv", len(acc), "rows")
