#!/usr/bin/env python3
"""step-by-step comparison HIP vs oracle for a block-timestep fixture (first divergence beyond rounding)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, PARAMS
from gandalf_amd.params import read_params_file
from gandalf_amd import GandalfHip
from oracle.pyoracle import Oracle
from test_oracle import upload_block_state
case = sys.argv[1]
g = load_golden(case + "_steps")
p = read_params_file("%s/%s.dat" % (PARAMS, case))
o = Oracle(p, nthreads=4)
s = lambda k: g["setup_" + k]
o.set_particles(s("r"), s("m"), s("h"), v=s("v"), u=s("u"))
upload_block_state(o, g, "setup_")
sim = GandalfHip(p)
sim.upload(s("r"), s("m"), s("h"), v=s("v"), u=s("u"))
for k in ["a", "r0", "v0", "a0", "u0", "dudt", "dudt0", "rho", "dt", "tlast", "dt_next", "div_v", "pressure", "sound", "hfactor", "invomega", "zeta", "hrangesqd", "alpha", "dalphadt", "gpot", "atree", "level", "levelneib", "nstep", "nlast"]:
    sim.upload_field(k, np.asarray(s(k), dtype=np.float64))
sim.upload_field("flags", np.zeros(len(s("m"))))
n, _, nresync = [int(x) for x in s("n_Nsteps_nresync")]
lmax, lstep = [int(x) for x in s("levelmax_levelstep_Nlevels_diffmax")[:2]]
sim.set_block_clock(n, nresync, lmax, lstep, float(s("dt_max")[0]))
t0, dt0 = s("t_timestep")
sim.set_time(float(t0), float(dt0))
for step in range(1, int(g["nsteps"][0]) + 1):
    o.step(1); sim.step(1)
    out = []
    for k in ("h", "rho", "a", "u", "dudt", "div_v"):
        a, b = sim.download(k), o.get(k)
        a = a.reshape(len(a), -1); b = b.reshape(len(b), -1)
        e = np.abs(a - b).max(axis=1)/np.maximum(np.abs(b).max(axis=1), np.abs(b).mean() + 1e-300)
        out.append("%s %.1e@%d" % (k, e.max(), int(e.argmax())))
    lv = (sim.download("level").astype(int) != o.get_int("level")).sum()
    ln = (sim.download("levelneib").astype(int) != o.get_int("levelneib")).sum()
    print("step %2d  %s  level!= %d levelneib!= %d" % (step, "  ".join(out), lv, ln), flush=True)
    if lv or ln or step in (6, 7):
        for k in ("level", "levelneib", "nstep", "nlast"):
            a, b = sim.download(k).astype(int), o.get_int(k)
            bad = np.nonzero(a != b)[0]
            print("     %s differs at %s gpu %s ref %s" % (k, bad[:8], a[bad[:8]], b[bad[:8]]))
        i0 = 760
        print("     gpu level", sim.download("level").astype(int)[i0:], "levelneib", sim.download("levelneib").astype(int)[i0:], "nlast", sim.download("nlast").astype(int)[i0:], "nstep", sim.download("nstep").astype(int)[i0:])
        print("     ref level", o.get_int("level")[i0:], "levelneib", o.get_int("levelneib")[i0:], "nlast", o.get_int("nlast")[i0:], "nstep", o.get_int("nstep")[i0:])
        print("     clock gpu", sim.get_block_clock(), "ref", o.get_block())
        if step >= 8: break
