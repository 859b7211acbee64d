#!/usr/bin/env python3
"""Print the kernel timeline of the last tree build in a rocprofv3 kernel_trace.csv (start offset, duration, gap)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_rootbox_partial")]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    name = r["Kernel_Name"].split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0)/1e3, (e - s)/1e3, (s - prev_end)/1e3, name))
    prev_end = max(prev_end, e)
    if name.startswith("k_pack_posm"):
        break
