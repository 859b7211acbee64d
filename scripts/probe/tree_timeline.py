"""print the kernel timeline of one tree build from a rocprofv3 kernel trace csv (debug aid)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last k_rootbox_partial and print until k_stock_top after it
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_rootbox_partial")]
i0 = idx[-3]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %7.1f  q%s  %s" % (s/1e3, e/1e3, (e - s)/1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
    if r["Kernel_Name"].startswith("k_stock_top") or s > 3e6:
        break
