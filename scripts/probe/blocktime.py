#!/usr/bin/env python3
"""probe (needs the -DGH_DEBUG_BLOCKTIME build): per-group duration / tiles / passes of one density pass"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from bench import WORKLOADS
from gandalf_amd.host import Simulation

wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "plummer1m"]
sim = Simulation(os.path.join(ROOT, "tests", "params", wl["params"]), **wl["overrides"])
sim.generate_ic(); sim.post_ic_setup(); sim.main_loop(2)
dev = sim.device()
dev.build_tree()
os.environ["GH_DEBUG_BLOCKTIME_FILE"] = "/tmp/blocktime.bin"
dev.update_density()
a = np.fromfile("/tmp/blocktime.bin").reshape(-1, 8)
t = a[:, 0]/100.0          # wall_clock64 ticks at 100 MHz -> microseconds
print("groups", len(t), "sum(us)", t.sum(), "mean", t.mean(), "median", np.median(t), "p99", np.percentile(t, 99), "max", t.max())
print("tiles mean", a[:, 1].mean(), "max", a[:, 1].max(), " passes mean", a[:, 2].mean(), "max", a[:, 2].max())
order = np.argsort(-t)[:15]
for g in order:
    print("group %6d  %8.1f us  tiles %6d  passes %2d  N %2d  ext %.3g %.3g %.3g  hmax %.3g" % (g, t[g], a[g, 1], a[g, 2], a[g, 3], a[g, 4], a[g, 5], a[g, 6], a[g, 7]))
h, e = np.histogram(t, bins=[0, 20, 40, 60, 80, 100, 150, 200, 300, 500, 1000, 1e9])
print("histogram (us):", list(zip(e[:-1].astype(int), h)))
