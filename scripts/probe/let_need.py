#!/usr/bin/env python3
"""How much of another rank's subtree do a rank's gravity walks really touch?  (sizing of the halo exchange)

Builds the tree of a Plummer sphere on one GPU, then evaluates, for rank 0 of a `world`-rank decomposition and every
other rank s, the reference's own opening tests (Tree.cpp:659-700) of every leaf of rank 0 against the cells of rank
s's subtree, level by level: a cell counts as visited when some leaf of rank 0 opens its parent.  Prints the visited
leaves per rank - the lower bound any halo selection has to import."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gandalf_amd.host import Simulation  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sim = Simulation(os.path.join(ROOT, "tests", "params", "plummer_4k.dat"), Nhydro=N)
sim.generate_ic()
sim.post_ic_setup()
dev = sim.device()
dev.build_tree()
dev.update_density()
t = dev.export_tree()
lev, rc, rmax, hmax, cd, cn = t["level"], t["rcell"], t["rmax"], t["hmax"], t["cdistsqd"], t["N"]
ltot = int(lev.max())
L = int(np.log2(world))
kr = 2.0


def subtree(c):
    return np.arange(c, c + (1 << (ltot - lev[c] + 1)) - 1)


def top(r):             # pre-order id of level-L cell r
    c = 0
    for b in range(L):
        bit = (r >> (L - 1 - b)) & 1
        c = c + 1 if bit == 0 else c + (1 << (ltot - lev[c]))
    return c


parent = np.full(len(lev), -1)
inner = np.nonzero(lev < ltot)[0]
parent[inner + 1] = inner
parent[inner + (1 << (ltot - lev[inner]))] = inner
SA = subtree(top(0))
la = SA[(lev[SA] == ltot) & (cn[SA] > 0)]
print("N=%d world=%d ltot=%d: rank 0 has %d leaves" % (N, world, ltot, len(la)), flush=True)
for s in range(1, world):
    SB = subtree(top(s))
    opened = np.zeros(len(lev), bool)
    for lv in range(L + 1, ltot):
        cells = SB[(lev[SB] == lv) & (cn[SB] > 0)]
        if lv > L + 1:
            cells = cells[opened[parent[cells]]]
        for c0 in range(0, len(cells), 64):
            cc = cells[c0:c0 + 64]
            d2 = ((rc[cc][:, None, :] - rc[la][None, :, :])**2).sum(-1)
            o = (d2 <= (rmax[cc][:, None] + rmax[la][None, :] + kr*np.maximum(hmax[la][None, :], hmax[cc][:, None]))**2) | (d2 < cd[cc][:, None])
            opened[cc] = o.any(1)
    leaves = SB[lev[SB] == ltot]
    vis = opened[parent[leaves]].sum()
    print("  from rank %d: %d of %d leaves visited (%.1f %%)" % (s, vis, len(leaves), 100.0*vis/len(leaves)), flush=True)
