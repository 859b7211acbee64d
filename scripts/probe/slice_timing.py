#!/usr/bin/env python3
"""probe: wall time of update_density / update_forces for slices of the group range (tail-effect check)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS
from gandalf_amd.host import Simulation

wl = WORKLOADS["plummer1m"]
sim = Simulation(os.path.join(ROOT, "tests", "params", wl["params"]), **wl["overrides"])
sim.generate_ic(); sim.post_ic_setup(); sim.main_loop(2)
dev = sim.device()
dev.build_tree()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0)/reps*1e3
for W in (1, 2, 4, 8, 16):
    for r in range(W):
        dev.set_shard(r, W)
        td = t(dev.update_density)
        dev.zero_accelerations()
        tf = t(dev.update_forces)
        print("W=%2d r=%2d density %.3f ms  forces(walk+eval) %.3f ms" % (W, r, td, tf), flush=True)
dev.set_shard(0, 1)
