#!/usr/bin/env python3
"""debug: force pass of an Nleafmax case (lists path / fused path), stage by stage"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gandalf_amd
from gandalf_amd.params import read_params_file
case = sys.argv[1]
if len(sys.argv) > 2 and sys.argv[2] == "fused":
    os.environ["GH_GRAV_FUSED"] = "1"
g = dict(np.load(os.path.join(ROOT, "tests", "golden", case + "_passes.npz")))
sim = gandalf_amd.GandalfHip(read_params_file(os.path.join(ROOT, "tests", "params", case + ".dat")))
sim.upload(g["in_r"], g["in_m"], g["in_h"], v=g["in_v"], u=g["in_u"])
sim.build_tree(); print("tree ok", flush=True)
sim.update_density(); print("density ok", flush=True)
sim.zero_accelerations()
sim.update_forces(); print("forces ok", flush=True)
a, aref = sim.download("a"), g["force_a"]
na = np.linalg.norm(aref, axis=1)
print("a err", np.max(np.linalg.norm(a - aref, axis=1)/np.maximum(na, na.mean())), "gpot err", np.max(np.abs(sim.download("gpot") - g["force_gpot"])/np.abs(g["force_gpot"])), flush=True)
