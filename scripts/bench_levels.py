#!/usr/bin/env python3
"""Block-timestep throughput on one MI355X: the metric's own definition - active particles summed
over the steps / wall time of the step loop - for the 1M Plummer sphere with Nlevels = 5.  Not the headline bench
(bench.py measures BASELINE's global-timestep configs); prints one JSON line.

    python scripts/bench_levels.py [--N 1048576] [--levels 5] [--cycles 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1048576)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--cycles", type=int, default=2, help="timed resynchronisation cycles (2^(levels-1) steps each)")
    args = ap.parse_args()
    from gandalf_amd.host import Simulation
    sim = Simulation(os.path.join(ROOT, "tests", "params", "plummer_4k.dat"), Nhydro=args.N, Nlevels=args.levels, run_id="PLUMLV")
    sim.generate_ic()
    t0 = time.perf_counter()
    sim.post_ic_setup()
    setup_s = time.perf_counter() - t0
    dev = sim.device()
    levels = np.bincount(dev.download("level").astype(int), minlength=args.levels)
    nres = dev.get_block_clock()[0][1]
    sim.main_loop(nres)                       # one whole cycle as warm-up
    dev.active_count(reset=True)
    dev.reset_timers()
    nsteps = 0
    t0 = time.perf_counter()
    for _ in range(args.cycles):
        n = dev.get_block_clock()[0][1]
        sim.main_loop(n)
        nsteps += n
    elapsed = time.perf_counter() - t0
    nact = dev.active_count()
    timers, _, _ = dev.timers()
    print(json.dumps({"metric": "particle-steps/s (active particles, block timesteps)", "value": nact/elapsed, "unit": "particle-steps/s",
                      "N": args.N, "Nlevels": args.levels, "base_steps": nsteps, "active_particle_steps": nact,
                      "mean_active_fraction": nact/(float(args.N)*nsteps), "ms_per_base_step": 1e3*elapsed/nsteps,
                      "equivalent_global_steps_per_s": None, "levels_after_setup": [int(x) for x in levels],
                      "phase_ms_per_base_step": {k: v/nsteps for k, v in timers.items()}, "setup_s": setup_s}))


if __name__ == "__main__":
    main()
