#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean of every counter per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    if not any(k in name for k in ("k_density", "k_grav", "k_hydro")):
        continue
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        vs = sorted(v)
        print("   %-28s n=%d mean=%.6g median=%.6g min=%.6g max=%.6g" %
              (c, len(v), sum(v)/len(v), vs[len(vs)//2], vs[0], vs[-1]))
