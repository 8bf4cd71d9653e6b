"""Timeline of the last frames of a rocprofv3 kernel trace of bench.py with frames in flight: per queue, each frame's encoder
and tail spans (first kernel start -> last kernel end), the kernels' own durations and the gaps between them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short, int(r.get("Workgroup_Size_X", 0) or 0), int(r.get("Grid_Size_X", 0) or 0)))
ks.sort()
t_end = ks[-1][1]
ks = [k for k in ks if k[0] > t_end - 3_000_000]            # last 3 ms
t0 = ks[0][0]
byq = collections.defaultdict(list)
for k in ks: byq[k[2]].append(k)
print("queues:", {q: len(v) for q, v in byq.items()})
def kind(n): return "tail" if n.startswith("tail") else "enc"
for q, v in byq.items():
    print(f"== queue {q}")
    frames, cur = [], []
    for k in v:
        if cur and kind(k[3]) == "enc" and kind(cur[-1][3]) == "tail": frames.append(cur); cur = []
        cur.append(k)
    for fr in frames[1:4]:
        enc = [k for k in fr if kind(k[3]) == "enc"]; tail = [k for k in fr if kind(k[3]) == "tail"]
        if not enc or not tail: continue
        print(f" frame at {(enc[0][0]-t0)/1e3:8.1f} us: encoder span {(enc[-1][1]-enc[0][0])/1e3:6.1f} us (kernels {sum(k[1]-k[0] for k in enc)/1e3:6.1f}), "
              f"tail span {(tail[-1][1]-tail[0][0])/1e3:6.1f} us (kernels {sum(k[1]-k[0] for k in tail)/1e3:6.1f}), enc->tail gap {(tail[0][0]-enc[-1][1])/1e3:5.1f}")
        prev = None
        for k in fr:
            gap = (k[0] - prev) / 1e3 if prev else 0.0
            print(f"     +{gap:5.1f}  {(k[1]-k[0])/1e3:6.1f} us  {k[3][:60]}")
            prev = k[1]
