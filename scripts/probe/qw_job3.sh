cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_sk2m
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sk2m -o sk -- python3 $GRAFT_REPO_ROOT/scripts/bench_sinks.py --N 2000000 --steps 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_sk2m.log 2>&1
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_sk2m/*kernel_trace.csv
python3 - <<'PY'
import csv,os
p=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_sk2m/sk_kernel_stats.csv"
rows=list(csv.DictReader(open(p)))
for r in rows[:22]:
    print("%-60s %6s %10.3f ms tot %9.1f us avg" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
