#!/usr/bin/env python3
"""debug: density pass of an Nleafmax case, split and fused paths, against the fixture"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gandalf_amd
from gandalf_amd.params import read_params_file
case = sys.argv[1] if len(sys.argv) > 1 else "plummer_4k_nl1"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", case + "_passes.npz")))
for fused in ("", "1"):
    if fused:
        os.environ["GH_DENSITY_FUSED"] = "1"
    sim = gandalf_amd.GandalfHip(read_params_file(os.path.join(ROOT, "tests", "params", case + ".dat")))
    sim.upload(g["in_r"], g["in_m"], g["in_h"], v=g["in_v"], u=g["in_u"])
    sim.build_tree()
    t = sim.export_tree()
    leaf = t["level"] == t["ltot"]
    print("  leaves:", int(leaf.sum()), "N per leaf min/max:", int(t["N"][leaf].min()), int(t["N"][leaf].max()), " leaf hmax == 0:", int((t["hmax"][leaf] == 0).sum()),
          " cells with hmax == 0:", int((t["hmax"] == 0).sum()), " order is a permutation:", bool(np.array_equal(np.sort(t["order"]), np.arange(len(t["order"])))))
    try:
        st = sim.update_density(stats=True)
        print("fused=%r ok" % fused, st)
    except Exception as e:      # noqa: BLE001
        print("fused=%r error: %s" % (fused, e))
    h, rho = sim.download("h"), sim.download("rho")
    bad = ~np.isfinite(h) | ~np.isfinite(rho)
    print("  non-finite:", int(bad.sum()), " h rel err max:", float(np.nanmax(np.abs(h - g["dens_h"])/g["dens_h"])), " n(h err > 1e-10):", int((np.abs(h - g["dens_h"])/g["dens_h"] > 1e-10).sum()))
    w = np.argsort(-np.abs(h - g["dens_h"])/g["dens_h"])[:5]
    pos = np.empty(len(t["order"]), dtype=np.int64); pos[t["order"]] = np.arange(len(t["order"]))
    badpos = np.sort(pos[bad])
    print("  tree-order positions of the non-finite particles (first 40):", badpos[:40], " distinct groups of 16:", len(set((badpos//16).tolist())), "of", int(bad.sum()))
    for i in w:
        print("   i=%d h_in=%.6g h=%.6g ref=%.6g rho=%.6g ref=%.6g" % (i, g["in_h"][i], h[i], g["dens_h"][i], rho[i], g["dens_rho"][i]))
    sim.close()
