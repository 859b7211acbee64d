#!/usr/bin/env python3
"""Profiling driver: set a workload up, then launch each hot kernel a few times so that a rocprofv3
pass (kernel-trace or --pmc) sees clean, repeated dispatches.

    rocprofv3 --pmc FETCH_SIZE -d out -- python3 scripts/prof_pass.py --workload box256k --reps 3
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402
from gandalf_amd.host import Simulation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="box256k", choices=sorted(WORKLOADS))
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    sim = Simulation(os.path.join(ROOT, "tests", "params", wl["params"]), **wl["overrides"])
    sim.generate_ic()
    sim.post_ic_setup()
    sim.main_loop(2)
    dev = sim.device()
    for _ in range(args.reps):
        dev.build_tree()
        dev.update_density()
        dev.zero_accelerations()
        dev.update_forces()
    st_d = None
    dev.build_tree()
    st_d = dev.update_density(stats=True)
    dev.zero_accelerations()
    st_f = dev.update_forces(stats=True)
    print("density", st_d)
    print("forces", st_f)


if __name__ == "__main__":
    main()
