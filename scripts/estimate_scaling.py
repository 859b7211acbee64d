#!/usr/bin/env python3
"""ESTIMATE (one GPU): per-rank compute time of the work-sharded step for W = 1, 2, 4, 8 ranks.

After a normal single-GPU setup + 2 steps, the phases of one step are timed at fixed positions with the
shard set to slice r of W (r = 0 and r = W//2): replicated phases (tree build, restock, KDK) + this slice's
density / gravity walk / force evaluation + the pack / unpack kernels of both exchanges.  The collectives
themselves are NOT included (no second GPU here); DESIGN.md section 7 adds a bandwidth model for them.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from gandalf_amd.host import Simulation  # noqa: E402
from gandalf_amd.multigpu import Exchanger  # noqa: E402


def main():
    wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "plummer1m"]
    sim = Simulation(os.path.join(ROOT, "tests", "params", wl["params"]), **wl["overrides"])
    sim.generate_ic()
    sim.post_ic_setup()
    sim.main_loop(2)
    dev = sim.device()
    out = {}
    for W in (1, 2, 4, 8):
        for r in sorted({0, W//2}):
            dev.set_shard(r, W)
            x = Exchanger(dev, r, W, torch.device("cuda", 0))
            x.simulate = True
            reps = 3
            for it in range(reps + 1):
                if it == 1:
                    dev.reset_timers()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                dev.build_tree()
                dev.update_density()
                x.exchange(dev.X_DENSITY)
                dev.update_hmax()
                dev.zero_accelerations()
                dev.update_forces()
                x.exchange(dev.X_FORCES)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0)/reps*1e3
            timers, _, _ = dev.timers()
            out["W%d_r%d" % (W, r)] = {"wall_ms_per_pass": wall, "phase_ms": {k: v/reps for k, v in timers.items()}}
            print("W=%d rank %d: %.2f ms per pass (no KDK, no collectives)  %s" %
                  (W, r, wall, {k: round(v/reps, 2) for k, v in timers.items()}), flush=True)
    dev.set_shard(0, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
