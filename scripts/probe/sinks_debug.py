"""step-by-step comparison of the GPU sink run with the CPU restatement (debug aid; run on the GPU box)"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gandalf_amd
from gandalf_amd.capi import NbodyHip
from gandalf_amd.params import read_params_file
from oracle.pyoracle import Oracle, NbodyOracle
from test_oracle import bb_initial_h

case = "bb_sinks_8k"
g = np.load(os.path.join(ROOT, "tests", "golden", case + "_steps.npz"))
p = read_params_file(os.path.join(ROOT, "tests", "params", case + ".dat"))
s = lambda k: g["setup_" + k]
h0 = bb_initial_h(p, s("m"))
sim = gandalf_amd.GandalfHip(p)
sim.upload(s("r"), s("m"), h0, v=s("v"), u=s("u"))
nb = NbodyHip(ndim=3, softening=int(p["nbody_softening"]), nbody_mult=float(p["nbody_mult"]))
nb.hybrid_setup(sim, initial_h_provided=True)
o = Oracle(p, nthreads=8)
o.set_particles(s("r"), s("m"), h0, v=s("v"), u=s("u"))
e = np.zeros(0)
no = NbodyOracle(e.reshape(0, 3), e.reshape(0, 3), e, e, int(p["nbody_softening"]), float(p["nbody_mult"]))
no.hybrid_setup(o, h_provided=True)
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    nb.hybrid_step(sim, 1)
    no.hybrid_step(o, 1)
    No = o.num_particles()
    print("step", step, "N", sim.N, No, "stars", nb.num_stars(), no.get("m").shape[0], "dt", sim.timestep if hasattr(sim, "timestep") else "", o.timestep)
    sk, so = sim.sinks(), o.sinks()
    print("  Ngas", sk["Ngas"], so["Ngas"], "menc", sk["menc"], so["menc"], "mmax", sk["mmax"], so["mmax"])
    print("  star m", nb.download("m"), no.get("m"))
    if sim.N == No:
        fl = sim.download("flags").astype(int); fo = o.get_int("flags")
        print("  dead", ((fl & 4) != 0).sum(), ((fo & 1) != 0).sum(), "sinkid diff", (sim.download("sinkid").astype(int) != o.get_int("sinkid")).sum(),
              "rho err", np.max(np.abs(sim.download("rho") - o.get("rho"))/o.get("rho")))
        dense = o.get("rho") >= float(p["rho_sink"])
        print("  potmin(dense) gpu", ((fl[dense] & 8) != 0).sum(), "ref", ((fo[dense] & 8) != 0).sum(), "ndense", dense.sum())
        print("  taccrete", sk["taccrete"], so["taccrete"], "trot", sk["trot"], so["trot"], "mmean", sk["mmean"], so["mmean"])
