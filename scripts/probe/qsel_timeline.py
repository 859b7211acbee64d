"""per-launch durations of k_qselect_level in a kernel trace (debug aid)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_qselect_level")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-17:]
tot = 0
for i, r in enumerate(last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3
    tot += d
    print(i, "%.1f us" % d, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))
print("total", tot)
