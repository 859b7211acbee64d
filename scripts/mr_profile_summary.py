#!/usr/bin/env python3
"""Per-kernel summary of the timed steps of a multi-rank rehearsal (scripts/probe/threaded_ranks.py under
`rocprofv3 --kernel-trace`): only the dispatches between the two sentinel "flip" kernels that bracket the W-rank
main loop count.  With THR_SERIAL=1 the ranks take turns on the one GPU, so every duration is what the kernel costs
on a GPU of its own and (sum of durations)/(ranks x steps) is the device work of one rank per step.

    python scripts/mr_profile_summary.py gpurun_out/prof/p_results.db WORLD NSTEPS > profiles/r03_multirank_8x1m_kernels.txt
"""
import sqlite3
import sys


def main():
    db, world, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    c = sqlite3.connect(db).cursor()
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    flips = [i for i, r in enumerate(rows) if "flip" in r[0]]
    assert len(flips) >= 2, "sentinel kernels not found"
    win = rows[flips[-2] + 1:flips[-1]]
    agg = {}
    for name, s, e in win:
        short = name.split("(")[0].replace("void ", "")
        if "rocprim" in short:
            short = "rocprim::" + short.split("::")[-1][:40]
        a = agg.setdefault(short, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (e - s)/1e3; a[2] = max(a[2], (e - s)/1e3)
    tot = sum(a[1] for a in agg.values())
    print("# %d ranks, %d timed steps: %d kernel dispatches, sum of durations %.3f ms = %.3f ms per rank per step" % (world, nsteps, len(win), tot/1e3, tot/1e3/world/nsteps))
    print("# %-58s %8s %12s %12s %14s" % ("kernel", "calls", "avg us", "max us", "us/rank/step"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-60s %8d %12.2f %12.2f %14.2f" % (k[:60], a[0], a[1]/a[0], a[2], a[1]/world/nsteps))


if __name__ == "__main__":
    main()
