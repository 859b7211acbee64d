#!/usr/bin/env python3
"""Config-5 shape on one MI355X: Boss-Bodenheimer cloud, sink creation + smooth accretion, block timesteps.
Not the headline bench; prints one JSON line.

    python scripts/bench_sinks.py [--N 262144] [--levels 5] [--steps 64] [--params bb_sinks_8k]

--params bb_units_1600: the settings of the reference's own tests/astro_tests/bossbodenheimer.dat (physical units, sink
density 5e-13 g cm^-3: the sinks form late in the collapse, the timed steps are the early ones).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=262144)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--params", default="bb_sinks_8k")
    args = ap.parse_args()
    from gandalf_amd.host import Simulation
    sim = Simulation(os.path.join(ROOT, "tests", "params", args.params + ".dat"), Nhydro=args.N, Nlevels=args.levels, run_id="BBSCALE")
    t0 = time.perf_counter()
    sim.setup()
    setup_s = time.perf_counter() - t0
    dev = sim.device()
    n0 = dev.N
    sim.main_loop(4)
    dev.reset_timers()
    t0 = time.perf_counter()
    sim.main_loop(args.steps)
    elapsed = time.perf_counter() - t0
    timers, _, _ = dev.timers()
    sk = dev.sinks()
    print(json.dumps({"workload": "Boss-Bodenheimer cloud, sinks + smooth accretion, Nlevels = %d (%s.dat)" % (args.levels, args.params), "N_start": n0, "N_end": dev.N,
                      "sinks": int(len(sk["radius"])), "sink_Ngas": [int(x) for x in sk["Ngas"]], "steps": args.steps,
                      "ms_per_step": 1e3*elapsed/args.steps, "phase_ms_per_step": {k: v/args.steps for k, v in timers.items()},
                      "setup_s": setup_s, "t": sim.t}))


if __name__ == "__main__":
    main()
