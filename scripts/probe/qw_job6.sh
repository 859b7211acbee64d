cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_sk2m
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sk2m -o sk -- python3 $GRAFT_REPO_ROOT/scripts/bench_sinks.py --N 2000000 --steps 4 > $GRAFT_REPO_ROOT/gpurun_out/prof_sk2m.log 2>&1
python3 - <<'PY'
import csv,os
p=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_sk2m/sk_kernel_trace.csv"
d=[]
for r in csv.DictReader(open(p)):
    if r["Kernel_Name"].startswith("k_qw_gather"):
        d.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
d.sort(reverse=True)
tot=sum(d)
print("k_qw_gather launches", len(d), "total ms", round(tot/1e3,2))
for n in (5,10,20,50,100,200): print("top", n, "=", round(sum(d[:n])/1e3,2), "ms")
print("longest", [round(x) for x in d[:15]])
import statistics
print("median us", statistics.median(d))
PY
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_sk2m/*kernel_trace.csv
