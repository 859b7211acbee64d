import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import PARAMS, load_golden
import gandalf_amd
from gandalf_amd.params import read_params_file
case = sys.argv[1] if len(sys.argv) > 1 else "adsod_1d_levels"
g = load_golden(case + "_steps")
sim = gandalf_amd.GandalfHip(read_params_file("%s/%s.dat" % (PARAMS, case)))
s = lambda k: g["setup_" + k]
sim.upload(s("r"), s("m"), s("h"), v=s("v"), u=s("u"))
for k in ["a", "r0", "v0", "a0", "u0", "dudt", "dudt0", "rho", "dt", "tlast", "dt_next", "div_v", "pressure", "sound", "hfactor",
          "invomega", "zeta", "hrangesqd", "alpha", "dalphadt", "gpot", "level", "levelneib", "nstep", "nlast"]:
    sim.upload_field(k, np.asarray(s(k), dtype=np.float64))
sim.upload_field("flags", np.zeros(len(s("m"))))
n, _, nresync = [int(x) for x in s("n_Nsteps_nresync")]
lmax, lstep = [int(x) for x in s("levelmax_levelstep_Nlevels_diffmax")[:2]]
sim.set_block_clock(n, nresync, lmax, lstep, float(s("dt_max")[0]))
sim.set_time(*[float(x) for x in s("t_timestep")])
print("clock", sim.get_block_clock(), "levels", np.bincount(s("level")))
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    try:
        print("step", it, sim.step(1), sim.get_block_clock())
    except Exception as e:
        print("ERR", e)
        for k in ["flags", "level", "nstep", "nlast", "h", "rho", "u", "tlast"]:
            a = sim.download(k)
            print(k, a[:6], a[505:515], "nan", np.isnan(a).sum(), "min", a.min(), "max", a.max())
        break
    import os
    rf = "scripts/probe/lvref_%d.npz" % (it + 1)
    if os.path.exists(rf):
        ref = np.load(rf)
        for k in ref.files:
            a = sim.download(k).reshape(len(ref[k]), -1); b = ref[k].reshape(len(ref[k]), -1).astype(float)
            bad = np.nonzero((np.abs(a - b) > 1e-9*np.maximum(np.abs(b), np.abs(b).mean() + 1e-300)).any(axis=1))[0]
            if len(bad): print("   DIFF", k, len(bad), bad[:10], a[bad[:4]].ravel(), b[bad[:4]].ravel())
