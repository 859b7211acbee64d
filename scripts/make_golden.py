#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the compiled reference (oracle/_ref/ref_dump).

Run in the build container only (needs /root/reference to have been compiled by
`make -f oracle/ref.mk`).  The fixtures are DATA: inputs and the reference's outputs for them.

    python scripts/make_golden.py
"""
import os
import subprocess
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gdmp import read_gdmp  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
GOLD = os.path.join(ROOT, "tests", "golden")

PARTICLE_IN = ["r", "v", "m", "h", "u", "iorig", "ptype"]
DENS_OUT = ["h", "rho", "invomega", "zeta", "hfactor", "hrangesqd", "sound", "pressure", "u", "div_v"]
FORCE_OUT = ["a", "atree", "gpot", "gpot_hydro", "dudt", "div_v", "dalphadt", "levelneib"]
STEP_OUT = ["r", "v", "a", "atree", "alpha", "dalphadt", "h", "rho", "u", "dudt", "gpot", "dt", "iorig", "r0", "v0", "a0", "u0", "dudt0",
            "t_timestep", "n_Nsteps_nresync"]
TREE = ["tree_Ncell_ltot_gtot_Ntot_Nleafmax", "cell_cnext", "cell_copen", "cell_level", "cell_ifirst",
        "cell_ilast", "cell_N", "cell_Nactive", "cell_cdistsqd", "cell_m", "cell_rmax", "cell_hmax",
        "cell_bbmin", "cell_bbmax", "cell_hboxmin", "cell_hboxmax", "cell_rcell", "cell_r", "cell_v",
        "inext"]


def run(args, cwd, threads=None, env=None):
    env = dict(env or os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")
    if threads:
        env["OMP_NUM_THREADS"] = str(threads)
    subprocess.run([REF] + args, cwd=cwd, check=True, stdout=subprocess.DEVNULL, env=env)


NSTEPS = {"plummer_4k_ts3": 10, "plummer_4k_tb4": 10, "box3d_4k_tb4": 10, "adsod_1d": 20, "adsod_1d_wadsley2008": 20, "adsod_1d_price2008": 20, "adsod_1d_mm97": 20, "adsod_mirror": 20}


def passes(name, nsteps=None):
    nsteps = nsteps or NSTEPS.get(name, 3)
    par = os.path.join(ROOT, "tests", "params", name + ".dat")
    with tempfile.TemporaryDirectory() as tmp:
        run(["passes", par, os.path.join(tmp, "p")], tmp)
        setup = read_gdmp(os.path.join(tmp, "p_setup.gdmp"))
        tree = read_gdmp(os.path.join(tmp, "p_tree.gdmp"))
        dens = read_gdmp(os.path.join(tmp, "p_density.gdmp"))
        forc = read_gdmp(os.path.join(tmp, "p_forces.gdmp"))
        out = {"ndim": setup["ndim"], "Nhydro": setup["Nhydro"]}
        for k in PARTICLE_IN:
            out["in_" + k] = setup[k]
        # state the reference held when setup finished (initial step quantities)
        for k in ["a", "atree", "gpot", "dudt", "div_v", "rho", "invomega", "zeta", "dt", "t_timestep"]:
            out["setup_" + k] = setup[k]
        for k in TREE:
            out["tree_" + k] = tree[k]        # tree after BuildTree on the setup state
        for k in DENS_OUT:
            out["dens_" + k] = dens[k]
        out["dens_cell_hmax"] = dens["cell_hmax"]
        out["dens_cell_hboxmin"] = dens["cell_hboxmin"]
        out["dens_cell_hboxmax"] = dens["cell_hboxmax"]
        out["dens_gather_offsets"] = dens["gather_offsets"]
        out["dens_gather_ids"] = dens["gather_ids"]
        for k in FORCE_OUT:
            out["force_" + k] = forc[k]
        if "Nstar" in setup:                 # hybrid gas + stars: star state, stars <- gas tree forces, + star-star direct sum
            sg = read_gdmp(os.path.join(tmp, "p_stargas.gdmp"))
            sa = read_gdmp(os.path.join(tmp, "p_starall.gdmp"))
            for k in ["r", "v", "m", "h"]:
                out["star_" + k] = setup["star_" + k]
            for k in ["a", "gpot"]:
                out["stargas_" + k] = sg["star_" + k]
                out["starall_" + k] = sa["star_" + k]
            out["starall_adot"] = sa["star_adot"]
        np.savez_compressed(os.path.join(GOLD, name + "_passes.npz"), **out)
        print(name, "passes ->", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")
    with tempfile.TemporaryDirectory() as tmp:
        run(["steps", par, os.path.join(tmp, "s"), str(nsteps)], tmp)
        setup = read_gdmp(os.path.join(tmp, "s_setup.gdmp"))
        final = read_gdmp(os.path.join(tmp, "s_final.gdmp"))
        out = {"ndim": setup["ndim"], "Nhydro": setup["Nhydro"], "nsteps": np.array([nsteps], dtype=np.int32)}
        for k in STEP_OUT:
            out["setup_" + k] = setup[k]
            out["final_" + k] = final[k]
        out["setup_m"] = setup["m"]
        if "Nstar" in setup:
            for k in ["r", "v", "a", "adot", "r0", "v0", "a0", "adot0", "m", "h", "gpot", "dt", "tlast"]:
                out["setup_star_" + k] = setup["star_" + k]
                out["final_star_" + k] = final["star_" + k]
        np.savez_compressed(os.path.join(GOLD, name + "_steps.npz"), **out)
        print(name, "steps ->", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


LEVEL_OUT = ["level", "levelneib", "nstep", "nlast", "flags", "tlast", "dt_next", "levelmax_levelstep_Nlevels_diffmax", "dt_max", "div_v",
             "pressure", "sound", "hfactor", "invomega", "zeta", "hrangesqd"]


def levels(name, nsteps=40):
    """block timesteps (Nlevels > 1): setup state (incl. the level structure) and the state after nsteps MainLoop calls"""
    par = os.path.join(ROOT, "tests", "params", name + ".dat")
    with tempfile.TemporaryDirectory() as tmp:
        run(["steps", par, os.path.join(tmp, "s"), str(nsteps)], tmp)
        setup = read_gdmp(os.path.join(tmp, "s_setup.gdmp"))
        final = read_gdmp(os.path.join(tmp, "s_final.gdmp"))
        out = {"ndim": setup["ndim"], "Nhydro": setup["Nhydro"], "nsteps": np.array([nsteps], dtype=np.int32)}
        for k in STEP_OUT + LEVEL_OUT:
            out["setup_" + k] = setup[k]
            out["final_" + k] = final[k]
        out["setup_m"] = setup["m"]
        np.savez_compressed(os.path.join(GOLD, name + "_steps.npz"), **out)
        print(name, "levels ->", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")



SINK_PART = ["flags", "sinkid", "m", "hrangesqd", "hfactor", "invomega", "zeta", "pressure", "sound", "tlast"]
SINK_STAR = ["r", "v", "a", "adot", "r0", "v0", "a0", "adot0", "m", "h", "gpot", "dt", "tlast", "invh", "radius", "dt_internal", "level", "nstep", "nlast"]
SINK_REC = ["radius", "mmax", "menc", "dmdt", "ketot", "gpetot", "rotketot", "utot", "taccrete", "trad", "trot", "tvisc", "angmom", "Ngas", "istar"]


def sinks(name, nsteps=12):
    """sink particles (Sinks.cpp) on the Boss-Bodenheimer cloud: the IC the reference generated, its post-setup state and the
    state after nsteps MainLoop calls (gas incl. dead / potmin flags and sinkid, the stars the sinks are, the SinkParticle records).
    One OpenMP thread: AccreteMassToSinks assigns sinkid in a parallel loop over sinks (Sinks.cpp:427-468)."""
    par = os.path.join(ROOT, "tests", "params", name + ".dat")
    with tempfile.TemporaryDirectory() as tmp:
        run(["steps", par, os.path.join(tmp, "s"), str(nsteps)], tmp, threads=1)
        setup = read_gdmp(os.path.join(tmp, "s_setup.gdmp"))
        final = read_gdmp(os.path.join(tmp, "s_final.gdmp"))
        out = {"ndim": setup["ndim"], "Nhydro": setup["Nhydro"], "nsteps": np.array([nsteps], dtype=np.int32)}
        for k in STEP_OUT + SINK_PART + ["level", "levelneib", "nstep", "nlast", "levelmax_levelstep_Nlevels_diffmax", "dt_max"]:
            out["setup_" + k] = setup[k]
            out["final_" + k] = final[k]
        out["final_Nhydro"] = final["Nhydro"]
        out["final_mmean_hminsink"] = final["mmean_hminsink"]
        out["final_Nsink"] = final["Nsink"]
        for k in SINK_STAR:
            out["final_star_" + k] = final["star_" + k]
        for k in SINK_REC:
            out["final_sink_" + k] = final["sink_" + k]
        np.savez_compressed(os.path.join(GOLD, name + "_steps.npz"), **out)
        print(name, "sinks ->", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def hybrid_levels(name, nsteps=24):
    """gas + stars on the block-timestep ladder (no sinks): setup state and the state after nsteps MainLoop calls, gas incl. the
    level structure, stars incl. level / nstep / nlast"""
    par = os.path.join(ROOT, "tests", "params", name + ".dat")
    with tempfile.TemporaryDirectory() as tmp:
        run(["steps", par, os.path.join(tmp, "s"), str(nsteps)], tmp)
        setup = read_gdmp(os.path.join(tmp, "s_setup.gdmp"))
        final = read_gdmp(os.path.join(tmp, "s_final.gdmp"))
        out = {"ndim": setup["ndim"], "Nhydro": setup["Nhydro"], "nsteps": np.array([nsteps], dtype=np.int32)}
        for k in STEP_OUT + LEVEL_OUT:
            out["setup_" + k] = setup[k]
            out["final_" + k] = final[k]
        out["setup_m"] = setup["m"]
        for k in SINK_STAR:
            if "star_" + k in setup:
                out["setup_star_" + k] = setup["star_" + k]
                out["final_star_" + k] = final["star_" + k]
        np.savez_compressed(os.path.join(GOLD, name + "_steps.npz"), **out)
        print(name, "hybrid levels ->", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def treeerror(n=32768):
    """the reference's tree-accuracy measurement (tests/paper_tests/treeerror.py:22-35): RMS relative force error of the
    KD-tree run against neib_search = bruteforce on the same Plummer sphere.  Stored: the brute-force accelerations and
    potentials (float32 is ample for a 5e-3 effect) and the reference's own tree error."""
    src = open(os.path.join(ROOT, "tests", "params", "plummer_4k.dat")).read().replace("Nhydro = 4096", "Nhydro = %d" % n)
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, text in (("kd", src), ("bf", src.replace("neib_search = kdtree", "neib_search = bruteforce"))):
            par = os.path.join(tmp, tag + ".dat")
            open(par, "w").write(text)
            run(["steps", par, os.path.join(tmp, tag), "0"], tmp)
            res[tag] = read_gdmp(os.path.join(tmp, tag + "_setup.gdmp"))
    kd, bf = res["kd"], res["bf"]
    assert np.array_equal(kd["iorig"], bf["iorig"])
    ferr = np.sqrt(np.mean(np.sum((kd["a"] - bf["a"])**2, axis=1)/np.sum(bf["a"]**2, axis=1)))
    perr = np.sqrt(np.mean((kd["gpot"] - bf["gpot"])**2))
    np.savez_compressed(os.path.join(GOLD, "plummer_32k_treeerror.npz"), Nhydro=np.array([n]), iorig=bf["iorig"].astype(np.int32),
                        a_bruteforce=bf["a"].astype(np.float32), gpot_bruteforce=bf["gpot"].astype(np.float32),
                        ref_force_error=np.array([ferr]), ref_gpot_error=np.array([perr]))
    print("treeerror: reference force error %.4e gpot error %.4e" % (ferr, perr))


def fromfile(name, nsteps=20):
    """a run that starts from a snapshot file (ic = file): setup + nsteps, in_file resolved against the repo root"""
    par = os.path.join(ROOT, "tests", "params", name + ".dat")
    with tempfile.TemporaryDirectory() as tmp:
        txt = open(par).read().replace("in_file = ", "in_file = " + ROOT + "/")
        tpar = os.path.join(tmp, "p.dat")
        open(tpar, "w").write(txt)
        run(["steps", tpar, os.path.join(tmp, "s"), str(nsteps)], tmp)
        setup = read_gdmp(os.path.join(tmp, "s_setup.gdmp"))
        final = read_gdmp(os.path.join(tmp, "s_final.gdmp"))
        out = {"nsteps": np.array([nsteps], dtype=np.int32)}
        for k in ["r", "v", "h", "rho", "u", "a", "t_timestep"]:
            out["setup_" + k] = setup[k]
            out["final_" + k] = final[k]
        np.savez_compressed(os.path.join(GOLD, name + "_steps.npz"), **out)
        print(name, "fromfile ->", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def snapshots():
    """the reference's own snapshot writers (`ref_dump snap`: column, su and sf files of one state, plus that state): the 3-D
    512-particle box after setup and the 1-D shock tube after two steps -> tests/golden/snapshots/{box,sod}.{column,su,sf},
    {box,sod}_state.npz"""
    dst = os.path.join(GOLD, "snapshots")
    os.makedirs(dst, exist_ok=True)
    for tag, par, nsteps in (("box", "box3d_512", 0), ("sod", "adsod_1d", 2)):
        with tempfile.TemporaryDirectory() as tmp:
            run(["snap", os.path.join(ROOT, "tests", "params", par + ".dat"), os.path.join(tmp, tag), str(nsteps)], tmp, threads=1)
            for ext in ("column", "su", "sf"):
                shutil.copy(os.path.join(tmp, "%s.%s" % (tag, ext)), os.path.join(dst, "%s.%s" % (tag, ext)))
            st = read_gdmp(os.path.join(tmp, tag + "_snap.gdmp"))
            keep = {k: st[k] for k in ("r", "v", "m", "h", "rho", "u", "iorig", "snap_t_tsnaplast_mmean_tlitesnaplast_hfac", "snap_Noutsnap_Nsteps_Noutlitesnap")}
            np.savez_compressed(os.path.join(dst, tag + "_state.npz"), **keep)
        print("snapshots:", tag)


def restart():
    """SimulationBase::Run with regular snapshots, then a restart (`ref_dump run`, REF_RESTART=1): the 1-D shock tube with
    dt_snap = 0.003 for 12 steps (snapshots 00001..00003 + ADSOD1D.restart), then - from a directory that holds only the
    restart file and the snapshot it names - 8 more steps as a restart.  -> tests/golden/restart/ADSOD1D.su.00003 (the
    reference's snapshot, a data file) and restart.npz"""
    dst = os.path.join(GOLD, "restart")
    os.makedirs(dst, exist_ok=True)
    src = open(os.path.join(ROOT, "tests", "params", "adsod_1d.dat")).read().replace("dt_snap = 100.0", "dt_snap = 0.003").replace("tsnapfirst = 100.0", "tsnapfirst = 0.0")
    with tempfile.TemporaryDirectory() as tmp:
        par = os.path.join(tmp, "p.dat")
        open(par, "w").write(src)
        run(["run", par, os.path.join(tmp, "a"), "12"], tmp, threads=1)
        names_a = sorted(f for f in os.listdir(tmp) if f.startswith("ADSOD1D.su."))
        text = open(os.path.join(tmp, "ADSOD1D.restart")).read()
        a = read_gdmp(os.path.join(tmp, "a_final.gdmp"))
        last = text.split()[1]
        shutil.copy(os.path.join(tmp, last), os.path.join(dst, last))
        tb = os.path.join(tmp, "b")
        os.makedirs(tb)
        for f in ("p.dat", "ADSOD1D.restart", last):
            shutil.copy(os.path.join(tmp, f), os.path.join(tb, f))
        env = dict(os.environ, REF_RESTART="1")
        run(["run", os.path.join(tb, "p.dat"), os.path.join(tb, "b"), "8"], tb, threads=1, env=env)
        names_b = sorted(f for f in os.listdir(tb) if f.startswith("ADSOD1D.su."))
        bs, b = read_gdmp(os.path.join(tb, "b_setup.gdmp")), read_gdmp(os.path.join(tb, "b_final.gdmp"))
        out = {"names_first_run": np.array(names_a), "restart_text": np.array([text]), "names_restarted_run": np.array(names_b),
               "first_t_tsnaplast_tsnapnext": a["run_t_tsnaplast_tsnapnext"], "first_Noutsnap_Nsteps": a["run_Noutsnap_Nsteps"],
               "restarted_t_tsnaplast_tsnapnext": b["run_t_tsnaplast_tsnapnext"], "restarted_Noutsnap_Nsteps": b["run_Noutsnap_Nsteps"]}
        for k in ("r", "v", "h", "rho", "u", "t_timestep"):
            out["first_final_" + k] = a[k]; out["restart_setup_" + k] = bs[k]; out["restarted_final_" + k] = b[k]
        np.savez_compressed(os.path.join(dst, "restart.npz"), **out)
        print("restart:", names_a, "->", names_b)


NBODY_FIELDS = ["r", "v", "a", "adot", "r0", "v0", "a0", "m", "h", "gpot", "dt", "t_dt"]


def nbody(N, softening, nsteps=3):
    """Star cluster through the reference's NbodyLeapfrogKDK (ref_dump nbody): direct-sum setup + nsteps."""
    name = "nbody_%d_%s" % (N, "soft" if softening else "point")
    with tempfile.TemporaryDirectory() as tmp:
        run(["nbody", str(N), str(softening), str(nsteps), os.path.join(tmp, "n")], tmp)
        setup = read_gdmp(os.path.join(tmp, "n_setup.gdmp"))
        final = read_gdmp(os.path.join(tmp, "n_final.gdmp"))
        out = {"N": np.array([N], dtype=np.int32), "softening": np.array([softening], dtype=np.int32),
               "nsteps": np.array([nsteps], dtype=np.int32), "nbody_mult": np.array([0.1])}
        for k in NBODY_FIELDS:
            out["setup_" + k] = setup[k]
            out["final_" + k] = final[k]
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
        print(name, "->", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def long_run(name, nsteps, tag):
    """final state of a long run (setup + nsteps MainLoop) - inputs are the IC of the parameter file"""
    par = os.path.join(ROOT, "tests", "params", name + ".dat")
    with tempfile.TemporaryDirectory() as tmp:
        run(["steps", par, os.path.join(tmp, "s"), str(nsteps)], tmp)
        setup = read_gdmp(os.path.join(tmp, "s_setup.gdmp"))
        final = read_gdmp(os.path.join(tmp, "s_final.gdmp"))
        out = {"nsteps": np.array([nsteps], dtype=np.int32)}
        for k in ["r", "v", "m", "h", "u"]:
            out["setup_" + k] = setup[k]
        for k in ["r", "v", "h", "rho", "u", "t_timestep"]:
            out["final_" + k] = final[k]
        np.savez_compressed(os.path.join(GOLD, "%s_%s.npz" % (name, tag)), **out)
        print(name, tag, "->", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    cases = sys.argv[1:] or ["box3d_4k", "plummer_4k", "adsod_1d", "nbody"]
    for cfg in cases:
        if cfg == "adsod_mirror_full":
            long_run("adsod_mirror", 1334, "full")          # tend = 5 of the root adsod.dat
        elif cfg.endswith("_fromfile"):
            fromfile(cfg)
        elif cfg == "snapshots":
            snapshots()
        elif cfg == "restart":
            restart()
        elif "_sinks" in cfg:
            sinks(cfg, 40 if cfg.endswith("_levels") else 12)
        elif cfg.endswith("_stars_levels"):
            hybrid_levels(cfg)
        elif cfg.endswith("_levels") or cfg.endswith("_levels_single"):
            levels(cfg)
        elif cfg == "treeerror":
            treeerror()

        elif cfg == "nbody":
            nbody(256, 0)
            nbody(256, 1)
        else:
            passes(cfg)
