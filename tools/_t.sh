cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/erprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/erprof -- python3 tools/bench_eraft.py > gpurun_out/erprof.log 2>&1
tail -3 gpurun_out/erprof.log
python3 - <<P
import csv,glob
f=glob.glob("gpurun_out/erprof/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:24]: print("%-90s %6s %9.1f us  %5s%%" % (r["Name"][:90].replace("(anonymous namespace)::",""), r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
P
