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

if int(sim.get_param("self_gravity")):
    dev.zero_accelerations()
    dev.update_forces()
    w = np.fromfile("/tmp/blocktime.bin.walk").reshape(-1, 8)
    tw = w[:, 0]/100.0
    print("WALK groups", len(tw), "mean", tw.mean(), "median", np.median(tw), "p99", np.percentile(tw, 99), "max", tw.max(), " steps mean", w[:, 1].mean(), "max", w[:, 1].max())
    for g in np.argsort(-tw)[:10]:
        print("  group %6d %8.1f us steps %5d glen %5d len_c0 %5d len_h0 %5d len_d0 %4d Rg %.3g hmax %.3g" % (g, tw[g], w[g, 1], w[g, 2], w[g, 3], w[g, 4], w[g, 5], w[g, 6], w[g, 7]))
    print("  histogram (us):", np.histogram(tw, bins=[0, 50, 100, 150, 200, 300, 500, 1000, 1e9])[0])
    e = np.fromfile("/tmp/blocktime.bin.eval")/100.0
    print("EVAL leaves", len(e), "mean", e.mean(), "median", np.median(e), "p99", np.percentile(e, 99), "max", e.max())
    print("  histogram (us):", np.histogram(e, bins=[0, 20, 40, 60, 80, 100, 150, 200, 300, 500, 1e9])[0])
