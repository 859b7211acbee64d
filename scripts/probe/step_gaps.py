"""idle gaps of the GPU inside one bench step, from a rocprofv3 kernel trace csv (debug aid)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_rootbox_partial")]
i0, i1 = idx[-3], idx[-2]
t0 = int(rows[i0]["Start_Timestamp"])
busy_end = t0
tot_gap = 0
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - busy_end
    if gap > 3000:
        tot_gap += gap
        print("%9.1f gap %6.1f us before %s" % ((s - t0)/1e3, gap/1e3, r["Kernel_Name"][:60]))
    busy_end = max(busy_end, e)
print("step %.1f us, idle %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0)/1e3, tot_gap/1e3))
# per-kernel totals within the step
from collections import defaultdict
tot = defaultdict(float); cnt = defaultdict(int)
for r in rows[i0:i1]:
    k = r["Kernel_Name"].split("(")[0][:50]
    tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3; cnt[k] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:40]:
    print("%8.1f us  x%-3d %s" % (v, cnt[k], k))
