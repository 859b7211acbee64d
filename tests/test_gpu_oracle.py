"""GPU parity against the CPU restatement (oracle/) on seeded inputs at sizes beyond the fixtures, and
size-independent properties at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest

from conftest import PARAMS
from gandalf_amd.params import read_params_file

pytestmark = pytest.mark.gpu


def vec_err(a, b):
    nb = np.linalg.norm(b, axis=1)
    return np.max(np.linalg.norm(a - b, axis=1)/np.maximum(nb, nb.mean()))


# ragged sizes: 30000 -> leaves of 3-4 particles, uneven cell splits; 10007 (prime) -> leaves of 4-5 particles, i.e.
# the 6-wide evaluation kernel; 3001 in 2-D (quadrupole, gadget2 bootstrap are covered by the fixtures)
@pytest.mark.parametrize("case,n", [("box3d_4k", 32768), ("plummer_4k", 16384), ("box3d_4k", 30000),
                                    ("plummer_4k", 10007), ("plummer_4k_quadrupole", 9001)])
def test_setup_and_steps_vs_oracle(case, n):
    """whole PostInitialConditionsSetup + 2 MainLoop steps from the raw IC, HIP vs oracle"""
    from gandalf_amd.host import Simulation
    from oracle.pyoracle import Oracle
    pf = os.path.join(PARAMS, case + ".dat")
    sim = Simulation(pf, Nhydro=n)
    ic = sim.generate_ic()
    sim.post_ic_setup()
    sim.main_loop(2)
    dev = sim.device()
    p = read_params_file(pf)
    o = Oracle(p)
    o.set_particles(ic["r"], ic["m"], ic["h"], v=ic["v"], u=ic["u"])
    o.setup(h_provided=ic["initial_h_provided"])
    o.step(2)
    assert abs(sim.t - o.t) <= 1e-12*abs(o.t)
    assert abs(sim.timestep - o.timestep) <= 1e-9*o.timestep
    assert np.max(np.abs(dev.download("r") - o.get("r"))) < 1e-11*np.abs(o.get("r")).max()
    assert np.max(np.abs(dev.download("h")/o.get("h") - 1)) < 1e-10
    assert np.max(np.abs(dev.download("rho")/o.get("rho") - 1)) < 1e-10
    assert vec_err(dev.download("a"), o.get("a")) < 1e-9
    if int(p.get("self_gravity", 0)):
        assert np.max(np.abs(dev.download("gpot")/o.get("gpot") - 1)) < 1e-10


def test_box256k_net_force_and_idempotence():
    """config 2 at full size: pairwise antisymmetry of the SPH force (net force ~ rounding) and
    idempotence of the density pass (a second pass from converged h changes h by < h_converge)"""
    from gandalf_amd.host import Simulation
    sim = Simulation(os.path.join(PARAMS, "box3d_4k.dat"), Nhydro=262144)
    sim.generate_ic()
    sim.post_ic_setup()
    dev = sim.device()
    m, a = dev.download("m"), dev.download("a")
    net = np.abs((m[:, None]*a).sum(axis=0)).max()
    assert net < 1e-11*(m[:, None]*np.abs(a)).sum()
    h0 = dev.download("h")
    dev.build_tree()
    dev.update_density()
    h1 = dev.download("h")
    assert np.max(np.abs(h1/h0 - 1)) < 0.01
    offs, _ = dev.gather_neighbours()
    cnt = np.diff(offs)
    assert 30 < cnt.mean() < 80 and cnt.min() >= 1


@pytest.mark.parametrize("case,n,nsteps", [("box3d_4k", 262144, 2), ("plummer_4k", 131072, 1)])
def test_bench_sizes_vs_oracle(case, n, nsteps):
    """HIP vs oracle at the sizes that are benchmarked: BASELINE configs[1] in full (262 144-particle periodic box, setup +
    2 steps) and the per-GPU share of configs[3] (131 072-particle Plummer sphere with self-gravity, setup + 1 step).
    Deep trees (ltot 16 / 15), the LDS-resident subtree build, the split density path with its list capacities and the
    gravity interaction lists are all exercised at the sizes they run at; same tolerances as the small cases."""
    from gandalf_amd.host import Simulation
    from oracle.pyoracle import Oracle
    pf = os.path.join(PARAMS, case + ".dat")
    sim = Simulation(pf, Nhydro=n)
    ic = sim.generate_ic()
    sim.post_ic_setup()
    sim.main_loop(nsteps)
    dev = sim.device()
    p = read_params_file(pf)
    o = Oracle(p)
    o.set_particles(ic["r"], ic["m"], ic["h"], v=ic["v"], u=ic["u"])
    o.setup(h_provided=ic["initial_h_provided"])
    o.step(nsteps)
    assert abs(sim.t - o.t) <= 1e-12*abs(o.t)
    assert abs(sim.timestep - o.timestep) <= 1e-9*o.timestep
    assert np.max(np.abs(dev.download("r") - o.get("r"))) < 1e-11*np.abs(o.get("r")).max()
    assert np.max(np.abs(dev.download("h")/o.get("h") - 1)) < 1e-10
    assert np.max(np.abs(dev.download("rho")/o.get("rho") - 1)) < 1e-10
    assert vec_err(dev.download("a"), o.get("a")) < 1e-9
    if int(p.get("self_gravity", 0)):
        assert np.max(np.abs(dev.download("gpot")/o.get("gpot") - 1)) < 1e-10


def _tree_force_error(a_tree, a_bf):
    """the reference's tree-accuracy metric, tests/paper_tests/treeerror.py:22-35"""
    return float(np.sqrt(np.mean(np.sum((a_tree - a_bf)**2, axis=1)/np.sum(a_bf**2, axis=1))))


def test_tree_force_error_matches_reference():
    """the RMS force error of the tree against neib_search = bruteforce (the reference's treeerror.py metric),
    on the 32 768-particle Plummer sphere at theta = 0.5, monopole.  The fixture holds the reference's brute-force
    accelerations and the error the reference's own KD-tree run makes against them (4.96e-3); the GPU tree run must make
    the same error to better than two significant digits."""
    from gandalf_amd.host import Simulation
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "plummer_32k_treeerror.npz"))
    n = int(g["Nhydro"][0])
    sim = Simulation(os.path.join(PARAMS, "plummer_4k.dat"), Nhydro=n)
    sim.generate_ic()
    sim.post_ic_setup()
    dev = sim.device()
    a = dev.download("a")
    err = _tree_force_error(a, g["a_bruteforce"].astype(np.float64))
    ref = float(g["ref_force_error"][0])
    assert abs(err/ref - 1.0) < 5e-3, (err, ref)
    gerr = float(np.sqrt(np.mean((dev.download("gpot") - g["gpot_bruteforce"].astype(np.float64))**2)))
    assert abs(gerr/float(g["ref_gpot_error"][0]) - 1.0) < 5e-3


def test_plummer1m_tree_force_error():
    """config 3 at full size: the same metric for 256 sample particles against the O(N) direct sum of the Newtonian far
    field (the softened near field is common to both sums).  At 1M particles the error of the theta = 0.5 monopole tree
    sits where the 32k reference measurement puts it (a few 1e-3), not at the loose 2 % bound of round 1."""
    from gandalf_amd.host import Simulation
    sim = Simulation(os.path.join(PARAMS, "plummer_4k.dat"), Nhydro=1048576)
    sim.generate_ic()
    sim.post_ic_setup()
    dev = sim.device()
    r, m, h = dev.download("r"), dev.download("m"), dev.download("h")
    atree = dev.download("atree")
    rng = np.random.default_rng(1)
    idx = rng.choice(len(m), 256, replace=False)
    num = den = 0.0
    for i in idx:
        dr = r - r[i]
        d2 = (dr*dr).sum(axis=1)
        far = d2 > (2*np.maximum(h, h[i]))**2         # beyond kernel softening: Newtonian in both sums
        afar = (m[far, None]*dr[far]/d2[far, None]**1.5).sum(axis=0)
        # tree value of the same far field: subtract the exactly summed near field (kernel-softened, identical in the tree
        # run: near particles are always direct / SPH neighbours there) by evaluating it with the reference's wgrav
        near = ~far
        near[i] = False
        anear = np.zeros(3)
        if near.any():
            s = np.sqrt(d2[near])
            def wgrav(x):                               # M4Kernel::wgrav, SmoothingKernel.h:205-219
                return np.where(x < 1, 4/3*x - 1.2*x**3 + 0.5*x**4,
                                np.where(x < 2, 8/3*x - 3*x*x + 1.2*x**3 - x**4/6 - 1/(15*x*x), 1/(x*x)))
            hi, hj = h[i], h[near]
            f = 0.5*(wgrav(s/hi)/hi**2 + wgrav(s/hj)/hj**2)      # GradhSph.cpp:538-544 without the zeta terms
            anear = (m[near, None]*dr[near]/s[:, None]*f[:, None]).sum(axis=0)
        num += np.sum((atree[i] - anear - afar)**2)/np.sum((afar + anear)**2)
        den += 1
    err = np.sqrt(num/den)
    assert err < 1.5e-2, err


def test_config1_full_run_vs_reference():
    """BASELINE configs[0] on the GPU path: root adsod.dat (gradhsph, mirror walls) to tend = 5, 1334 steps, against
    the reference's final state.  Rounding differences grow through 1334 steps of a shock tube: 1e-8 on r, rho."""
    from gandalf_amd.host import Simulation
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "adsod_mirror_full.npz"))
    sim = Simulation(os.path.join(PARAMS, "adsod_mirror.dat"))
    sim.generate_ic()
    sim.post_ic_setup()
    sim.main_loop(int(g["nsteps"][0]))
    dev = sim.device()
    assert abs(sim.t - g["final_t_timestep"][0]) < 1e-9
    assert np.max(np.abs(dev.download("r") - g["final_r"])) < 1e-8
    assert np.max(np.abs(dev.download("rho")/g["final_rho"] - 1)) < 1e-7
    assert np.max(np.abs(dev.download("v") - g["final_v"])) < 1e-7
    rho = dev.download("rho")
    assert 0.2 < rho.min() and rho.max() < 1.1          # between the two initial states (rhofluid2 = 0.25, rhofluid1 = 1)
    # the reference's own acceptance criterion for the Sod tube (tests/hydro_tests/test_adsod.py:11-18): L1 error of
    # vx against the exact Riemann solution < 9e-3 (analysis/compute.py:109-146: mean |vx - exact| over the particles;
    # taken over the part of the tube the waves have reached, |x| < 10) - and the same number as the reference's final state
    from riemann import exact_riemann, l1_error
    x, vx = dev.download("r")[:, 0], dev.download("v")[:, 0]
    m = np.abs(x) < 10.0
    _, u_exact, _ = exact_riemann(x[m], sim.t, 1.0, 0.0, 1.0, 0.25, 0.0, 0.1795, 1.4)
    l1 = l1_error(x[m], vx[m], u_exact)
    xr, vr = g["final_r"][:, 0], g["final_v"][:, 0]
    mr = np.abs(xr) < 10.0
    l1_ref = l1_error(xr[mr], vr[mr], exact_riemann(xr[mr], float(g["final_t_timestep"][0]), 1.0, 0.0, 1.0, 0.25, 0.0, 0.1795, 1.4)[1])
    assert l1 < 9e-3 and abs(l1 - l1_ref) < 1e-6, (l1, l1_ref)


@pytest.mark.parametrize("case", ["adsod_1d_levels", "plummer_4k_levels"])
def test_block_timesteps_from_ic(case):
    """whole run with hierarchical block timesteps through the host shell: our IC generator, gh_setup (which builds the
    level structure by a resynchronisation) and 40 MainLoop calls, against the reference's state after the same run"""
    from gandalf_amd.host import Simulation
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", case + "_steps.npz"))
    sim = Simulation(os.path.join(PARAMS, case + ".dat"))
    sim.generate_ic()
    sim.post_ic_setup()
    dev = sim.device()
    for k in ["level", "nstep", "nlast"]:          # the level ladder right after setup
        assert np.array_equal(dev.download(k).astype(np.int64), g["setup_" + k]), k
    clock, dt_max = dev.get_block_clock()
    assert clock[1:] == [int(g["setup_n_Nsteps_nresync"][2])] + [int(x) for x in g["setup_levelmax_levelstep_Nlevels_diffmax"][:2]]
    assert abs(dt_max - float(g["setup_dt_max"][0])) < 1e-10*dt_max
    sim.main_loop(int(g["nsteps"][0]))
    assert abs(sim.t - g["final_t_timestep"][0]) < 1e-10*abs(g["final_t_timestep"][0])
    for k in ["level", "nstep", "nlast"]:
        assert np.array_equal(dev.download(k).astype(np.int64), g["final_" + k]), k
    assert np.max(np.abs(dev.download("r") - g["final_r"])) < 1e-9*np.abs(g["final_r"]).max()
    assert np.max(np.abs(dev.download("rho")/g["final_rho"] - 1)) < 1e-8


def test_run_from_reference_snapshot(tmp_path):
    """ic = file: start from a SEREN-unformatted snapshot the REFERENCE wrote (tests/golden/snapshots/sod.su, t = 0.0015),
    run the setup and 20 steps, compare with the reference's own run from the same file; then write the state back as a
    snapshot and read it again"""
    from gandalf_amd.host import Simulation, read_snapshot
    here = os.path.dirname(__file__)
    g = np.load(os.path.join(here, "golden", "adsod_1d_fromfile_steps.npz"))
    sim = Simulation(os.path.join(PARAMS, "adsod_1d_fromfile.dat"), in_file=os.path.join(here, "golden", "snapshots", "sod.su"))
    sim.setup()
    dev = sim.device()
    assert abs(sim.t - g["setup_t_timestep"][0]) < 1e-15
    assert np.max(np.abs(dev.download("h")/g["setup_h"] - 1)) < 1e-12
    assert abs(sim.timestep - g["setup_t_timestep"][1]) < 1e-12*sim.timestep
    sim.main_loop(int(g["nsteps"][0]))
    assert abs(sim.t - g["final_t_timestep"][0]) < 1e-12
    assert np.max(np.abs(dev.download("r") - g["final_r"])) < 1e-11
    assert np.max(np.abs(dev.download("rho")/g["final_rho"] - 1)) < 1e-10
    assert np.max(np.abs(dev.download("u")/g["final_u"] - 1)) < 1e-10
    out = str(tmp_path / "final.su")
    sim.write_snapshot(out, "su")
    f = read_snapshot(out, "su")
    assert f["t"] == sim.t and f["Nsteps"] == int(g["nsteps"][0])
    assert np.array_equal(f["rho"], dev.download("rho")) and np.array_equal(f["r"].ravel(), dev.download("r").ravel())


def test_run_from_reference_sf_snapshot():
    """ic = file with in_file_form = sf: unlike the column / su readers, the reference's formatted reader leaves the time at
    0 unless the run is a restart (SimulationIO.hpp:677-687); the smoothing lengths are recomputed from scratch as for every
    snapshot (the reader's "h provided" is cleared again by SimulationIC.hpp:91).  The run from tests/golden/snapshots/sod.sf:
    setup + 20 steps against the reference's own run from the same file"""
    from gandalf_amd.host import Simulation
    here = os.path.dirname(__file__)
    g = np.load(os.path.join(here, "golden", "adsod_1d_sf_fromfile_steps.npz"))
    sim = Simulation(os.path.join(PARAMS, "adsod_1d_sf_fromfile.dat"), in_file=os.path.join(here, "golden", "snapshots", "sod.sf"))
    sim.setup()
    dev = sim.device()
    assert sim.t == 0.0 == g["setup_t_timestep"][0]
    assert np.max(np.abs(dev.download("h")/g["setup_h"] - 1)) < 1e-12
    assert abs(sim.timestep - g["setup_t_timestep"][1]) < 1e-12*sim.timestep
    sim.main_loop(int(g["nsteps"][0]))
    assert abs(sim.t - g["final_t_timestep"][0]) < 1e-12
    assert np.max(np.abs(dev.download("r") - g["final_r"])) < 1e-11
    assert np.max(np.abs(dev.download("rho")/g["final_rho"] - 1)) < 1e-10
    assert np.max(np.abs(dev.download("u")/g["final_u"] - 1)) < 1e-10


def test_reference_bossbodenheimer_settings_in_physical_units(tmp_path):
    """BASELINE configs[4]'s parameter file: tests/params/bb_units_1600.dat carries the settings of the reference's own
    tests/astro_tests/bossbodenheimer.dat - physical units (pc, m_sun, myr; the barotropic EOS in K and g cm^-3, sink density
    5e-13 g cm^-3, angular velocity in rad/s), 8 000 particles on a hexagonal lattice sphere, tabulated M4 kernel, fast
    monopoles, sinks with smooth accretion, five block-timestep levels.  The host shell converts to code units the way
    SimUnits::SetupUnits and the parameter processing do (SimUnits.cpp:825-1118; Simulation.cpp:1121-1227;
    SphSimulation.cpp:128-136; BarotropicEOS.cpp:40-42; BossBodenheimerIc.cpp:55-58) and runs setup + 30 steps: against the
    reference's own run of the same file (code units); then a snapshot in OUTPUT units with the 21 unit ids of the header
    against the one the reference wrote at the same state."""
    from gandalf_amd.host import Simulation, read_snapshot
    here = os.path.dirname(__file__)
    g = np.load(os.path.join(here, "golden", "bb_units_1600_steps.npz"))
    sim = Simulation(os.path.join(PARAMS, "bb_units_1600.dat"))
    sim.setup()
    dev = sim.device()
    n = int(g["Nhydro"][0])
    assert dev.N == n and sim.t == 0.0
    assert abs(sim.timestep - g["setup_t_timestep"][1]) < 1e-10*sim.timestep
    assert np.max(np.abs(dev.download("r") - g["setup_r"])) < 1e-12*np.abs(g["setup_r"]).max()
    assert np.max(np.abs(dev.download("m")/g["setup_m"] - 1)) < 1e-13
    assert np.max(np.abs(dev.download("u")/g["setup_u"] - 1)) < 1e-12          # temp0 and rho_bary in code units
    assert np.max(np.abs(dev.download("h")/g["setup_h"] - 1)) < 1e-11 and np.max(np.abs(dev.download("rho")/g["setup_rho"] - 1)) < 1e-11
    a, ar = dev.download("a"), g["setup_a"]
    assert np.max(np.linalg.norm(a - ar, axis=1)/np.maximum(np.linalg.norm(ar, axis=1), np.linalg.norm(ar, axis=1).mean())) < 1e-9
    assert np.array_equal(dev.download("level").astype(np.int64), g["setup_level"])
    # the state after setup as a snapshot in output units
    out = str(tmp_path/"bb.su")
    sim.write_snapshot(out, "su")
    raw = open(out, "rb").read()
    off = 20 + 4*4 + 50*4 + 50*8 + 50*8 + 50*8
    assert [raw[off + 20*i:off + 20*(i + 1)].decode().strip() for i in range(21)] == list(g["snap_units"])
    f = read_snapshot(out, "su")
    for k in ("r", "v", "m", "h", "rho", "u"):
        ref = g["snap_" + k]
        assert np.max(np.abs(f[k][:64] - ref)) <= 1e-10*np.max(np.abs(ref)), k
    assert abs(f["mmean"] - g["snap_t_mmean_hfac"][1]) <= 1e-13*f["mmean"]
    # ... and back: a run that starts from that file (ic = file) divides by the same scales (Simulation::ConvertToCodeUnits)
    again = Simulation(os.path.join(PARAMS, "bb_units_1600.dat"), ic="file", in_file=out, in_file_form="su")
    ic2 = again.generate_ic()
    assert np.max(np.abs(ic2["r"] - g["setup_r"])) < 1e-14*np.abs(g["setup_r"]).max() and np.max(np.abs(ic2["m"]/g["setup_m"] - 1)) < 1e-14
    assert np.max(np.abs(ic2["u"]/g["setup_u"] - 1)) < 1e-12 and np.max(np.abs(ic2["v"] - g["setup_v"])) < 1e-13*np.abs(g["setup_v"]).max()
    sim.main_loop(int(g["nsteps"][0]))
    assert abs(sim.t - g["final_t_timestep"][0]) < 1e-11*sim.t
    for k in ("level", "nstep", "nlast"):
        assert np.array_equal(dev.download(k).astype(np.int64), g["final_" + k]), k
    assert np.max(np.abs(dev.download("r") - g["final_r"])) < 1e-10*np.abs(g["final_r"]).max()
    assert np.max(np.abs(dev.download("rho")/g["final_rho"] - 1)) < 1e-8 and np.max(np.abs(dev.download("u")/g["final_u"] - 1)) < 1e-8


def test_reference_examples_bossbodenheimer_settings(tmp_path):
    """tests/params/bb_units_ex_1600.dat = the settings of the reference's examples/bossbodenheimer.dat: physical units, 1 600
    particles, leaves of 8 particles (Nleafmax = 8), global timestep, sinks, formatted SEREN files.  Setup + 20 steps against
    the reference's run of that file; the sf snapshot of the setup state against the file the reference wrote
    (tests/golden/snapshots/bb_units_ex.sf): header, unit ids and array descriptors line for line, the numbers to the 11
    digits the format keeps."""
    from gandalf_amd.host import Simulation, read_snapshot
    here = os.path.dirname(__file__)
    g = np.load(os.path.join(here, "golden", "bb_units_ex_1600_steps.npz"))
    sim = Simulation(os.path.join(PARAMS, "bb_units_ex_1600.dat"))
    sim.setup()
    dev = sim.device()
    assert dev.N == int(g["Nhydro"][0]) and abs(sim.timestep - g["setup_t_timestep"][1]) < 1e-10*sim.timestep
    assert np.max(np.abs(dev.download("h")/g["setup_h"] - 1)) < 1e-11 and np.max(np.abs(dev.download("rho")/g["setup_rho"] - 1)) < 1e-11
    a, ar = dev.download("a"), g["setup_a"]
    assert np.max(np.linalg.norm(a - ar, axis=1)/np.maximum(np.linalg.norm(ar, axis=1), np.linalg.norm(ar, axis=1).mean())) < 1e-9
    out = str(tmp_path/"bbex.sf")
    sim.write_snapshot(out, "sf")
    ours, ref = open(out).read().splitlines(), open(os.path.join(here, "golden", "snapshots", "bb_units_ex.sf")).read().splitlines()
    nhead = 5 + 200 + 21 + 7 + 7
    assert len(ours) == len(ref) == nhead + 7*dev.N
    for i in range(nhead):
        if ours[i] != ref[i]:                                  # (a header real may differ in its last printed digit: mmean)
            assert abs(float(ours[i]) - float(ref[i])) <= 1e-9*abs(float(ref[i])), (i, ours[i], ref[i])
    assert ours[205:226] == ref[205:226] and ours[226:240] == ref[226:240]         # unit ids; array ids and descriptors
    f, fr = read_snapshot(out, "sf"), read_snapshot(os.path.join(here, "golden", "snapshots", "bb_units_ex.sf"), "sf")
    for k in ("r", "v", "m", "h", "rho", "u"):
        assert np.max(np.abs(f[k] - fr[k])) <= 1e-9*np.max(np.abs(fr[k])), k
    assert sum(x != y for x, y in zip(ours, ref)) < 0.01*len(ref)              # nearly every line is the same text
    sim.main_loop(int(g["nsteps"][0]))
    assert abs(sim.t - g["final_t_timestep"][0]) < 1e-11*sim.t
    assert np.max(np.abs(dev.download("r") - g["final_r"])) < 1e-10*np.abs(g["final_r"]).max()
    assert np.max(np.abs(dev.download("rho")/g["final_rho"] - 1)) < 1e-8 and np.max(np.abs(dev.download("u")/g["final_u"] - 1)) < 1e-8


def test_regular_snapshots_and_restart(tmp_path, monkeypatch):
    """SimulationBase::Run with its regular snapshots, then a restart (Simulation.cpp:382-600, SimulationIC.hpp:64-82): the 1-D
    shock tube with dt_snap = 0.003.  (1) 12 steps from the IC: the snapshot files <run_id>.su.NNNNN the reference wrote, by
    name, its <run_id>.restart text, the header of the last one (Noutsnap, Nsteps, tsnaplast) and its particle data.  (2) A
    directory that holds only the REFERENCE's restart file and the snapshot it names: started as a restart, the run picks up
    t, Nsteps, Noutsnap and the snapshot clock from the file, and after 8 more steps equals the reference's own restarted run -
    including the names of the snapshots written on the way.  Fixture: scripts/make_golden.py restart."""
    import shutil
    from gandalf_amd.host import Simulation, read_snapshot
    here = os.path.dirname(__file__)
    g = np.load(os.path.join(here, "golden", "restart", "restart.npz"))
    over = {"dt_snap": 0.003, "tsnapfirst": 0.0}
    # (1) from the IC
    d1 = tmp_path/"first"; d1.mkdir(); monkeypatch.chdir(d1)
    sim = Simulation(os.path.join(PARAMS, "adsod_1d.dat"), **over)
    sim.set_output(True)
    sim.setup()
    sim.run(12)
    assert sorted(f for f in os.listdir(d1) if ".su." in f) == list(g["names_first_run"])
    assert open(d1/"ADSOD1D.restart").read() == str(g["restart_text"][0])
    assert (sim.Noutsnap, sim.Nsteps) == tuple(int(x) for x in g["first_Noutsnap_Nsteps"])
    assert abs(sim.t - g["first_t_tsnaplast_tsnapnext"][0]) < 1e-12
    last = str(g["restart_text"][0]).split()[1]
    ours, ref = read_snapshot(str(d1/last), "su"), read_snapshot(os.path.join(here, "golden", "restart", last), "su")
    for k in ("Noutsnap", "Nsteps", "tsnaplast", "N"):
        assert ours[k] == ref[k], k
    assert abs(ours["t"] - ref["t"]) < 1e-12 and np.array_equal(ours["iorig"], ref["iorig"])
    for k in ("r", "v", "h", "rho", "u"):
        assert np.max(np.abs(ours[k] - ref[k])) <= 1e-10*np.max(np.abs(ref[k])), k
    # (2) restart from the reference's files
    d2 = tmp_path/"again"; d2.mkdir(); monkeypatch.chdir(d2)
    shutil.copy(os.path.join(here, "golden", "restart", last), d2/last)
    (d2/"ADSOD1D.restart").write_text(str(g["restart_text"][0]))
    sim = Simulation(os.path.join(PARAMS, "adsod_1d.dat"), **over)
    sim.set_output(True); sim.set_restart(True)
    sim.setup()
    dev = sim.device()
    assert sim.t == g["restart_setup_t_timestep"][0] and sim.Nsteps == ref["Nsteps"] and sim.Noutsnap == ref["Noutsnap"]
    assert abs(sim.timestep - g["restart_setup_t_timestep"][1]) < 1e-12*sim.timestep
    assert np.max(np.abs(dev.download("h")/g["restart_setup_h"] - 1)) < 1e-12
    sim.run(8)
    assert (sim.Noutsnap, sim.Nsteps) == tuple(int(x) for x in g["restarted_Noutsnap_Nsteps"])
    assert abs(sim.t - g["restarted_t_tsnaplast_tsnapnext"][0]) < 1e-12
    assert sorted(f for f in os.listdir(d2) if ".su." in f) == list(g["names_restarted_run"])
    assert np.max(np.abs(dev.download("r") - g["restarted_final_r"])) < 1e-11
    assert np.max(np.abs(dev.download("rho")/g["restarted_final_rho"] - 1)) < 1e-10
    assert np.max(np.abs(dev.download("u")/g["restarted_final_u"] - 1)) < 1e-10
    # a temporary restart snapshot every nrestartstep steps (Simulation.cpp:592-632): <run_id>.su.tmp, named in <run_id>.restart
    d4 = tmp_path/"tmpsnap"; d4.mkdir(); monkeypatch.chdir(d4)
    sim = Simulation(os.path.join(PARAMS, "adsod_1d.dat"), nrestartstep=5, **over)
    sim.set_output(True)
    sim.setup()
    sim.run(5)
    assert open(d4/"ADSOD1D.restart").read().split() == ["su", "ADSOD1D.su.tmp"]
    f = read_snapshot(str(d4/"ADSOD1D.su.tmp"), "su")
    assert f["Nsteps"] == 5 and f["t"] == sim.t
    # a restart without a restart file is an ordinary start (SimulationIC.hpp:77-80)
    d3 = tmp_path/"none"; d3.mkdir(); monkeypatch.chdir(d3)
    sim = Simulation(os.path.join(PARAMS, "adsod_1d.dat"), **over)
    sim.set_restart(True)
    sim.setup()
    assert sim.t == 0.0 and sim.Nsteps == 0


@pytest.mark.parametrize("case", ["box3d_4k_tb4", "plummer_4k_tb4", "plummer_4k_ts3"])
def test_restocked_tree_runs_match_reference(case):
    """ntreebuildstep = 4: from the IC through the setup and ten steps - the tree is rebuilt on steps 1, 4, 8 and re-stocked
    (same cells, properties from the moved particles) on the others, as HydroTree::BuildTree does"""
    from gandalf_amd.host import Simulation
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", case + "_steps.npz"))
    sim = Simulation(os.path.join(PARAMS, case + ".dat"))
    sim.setup()
    sim.main_loop(int(g["nsteps"][0]))
    dev = sim.device()
    assert abs(sim.t - g["final_t_timestep"][0]) < 1e-11*abs(g["final_t_timestep"][0])
    assert np.max(np.abs(dev.download("r") - g["final_r"])) < 1e-10*np.abs(g["final_r"]).max()
    assert np.max(np.abs(dev.download("rho")/g["final_rho"] - 1)) < 1e-9
    a, ar = dev.download("a"), g["final_a"]
    assert np.max(np.linalg.norm(a - ar, axis=1)/np.maximum(np.linalg.norm(ar, axis=1), np.linalg.norm(ar, axis=1).mean())) < 1e-8


@pytest.mark.parametrize("nleafmax", [6, 16])
def test_potmin_flags_vs_oracle(nleafmax):
    """(Nleafmax = 16: leaves of up to 16 particles go through the serial kernel - the wave kernel takes 8 slots per leaf.)
    The potential-minimum flag (GradhSph.cpp:270-280, with its stale-distance quirk) for EVERY particle the sink search could
    read it from: the Boss-Bodenheimer cloud with rho_sink lowered so that thousands of particles qualify and Nsinkfixed = 0 so that no
    sink forms; after two steps (real potentials) the flags of all particles with rho >= rho_sink equal the CPU restatement's,
    and they are a non-trivial pattern (neither all set nor all clear)"""
    import gandalf_amd
    from gandalf_amd.capi import NbodyHip
    from oracle.pyoracle import Oracle, NbodyOracle
    from conftest import load_golden
    from test_oracle import bb_initial_h
    g = load_golden("bb_sinks_8k_steps")
    p = read_params_file("%s/bb_sinks_8k.dat" % PARAMS)
    p["rho_sink"] = "0.28"; p["Nsinkfixed"] = "0"      # (h floor 0.092 < the initial h 0.0965: below that every ComputeH call returns -1)
    p["Nleafmax"] = str(nleafmax)
    r, v, m, u = g["setup_r"], g["setup_v"], g["setup_m"], g["setup_u"]
    h0 = bb_initial_h(p, m)
    sim = gandalf_amd.GandalfHip(p)
    sim.upload(r, m, h0, v=v, u=u)
    nb = NbodyHip(ndim=3, softening=1, nbody_mult=float(p["nbody_mult"]))
    nb.hybrid_setup(sim, initial_h_provided=True)
    nb.hybrid_step(sim, 2)
    o = Oracle(p, nthreads=8)
    o.set_particles(r, m, h0, v=v, u=u)
    e = np.zeros(0)
    no = NbodyOracle(e.reshape(0, 3), e.reshape(0, 3), e, e, 1, float(p["nbody_mult"]))
    no.hybrid_setup(o, h_provided=True)
    no.hybrid_step(o, 2)
    assert sim.N == o.num_particles() == len(m) and nb.num_stars() == 0
    rho = o.get("rho")
    assert np.max(np.abs(sim.download("rho") - rho)/rho) < 1e-11
    assert np.max(np.abs(sim.download("h") - o.get("h"))/o.get("h")) < 1e-11
    dense = rho >= 0.28*(1.0 + 1e-9)
    pm_gpu = (sim.download("flags").astype(np.int64)[dense] & 8) != 0
    pm_ref = (o.get_int("flags")[dense] & 8) != 0
    assert dense.sum() > 1500 and 0 < pm_ref.sum() < dense.sum()
    assert np.array_equal(pm_gpu, pm_ref), (int((pm_gpu != pm_ref).sum()), int(pm_ref.sum()))


def test_sink_run_from_parameter_file():
    """the host shell runs the Boss-Bodenheimer sink case from its parameter file alone (ic = bb generated on the host,
    sink_particles = 1 -> star context + gh_hybrid_setup / gh_hybrid_step inside SphSimulation): same particle count, sinks
    and positions as the reference after 12 steps"""
    from conftest import load_golden
    from gandalf_amd.host import Simulation
    g = load_golden("bb_sinks_8k_steps")
    sim = Simulation("%s/bb_sinks_8k.dat" % PARAMS)
    sim.setup()
    assert abs(sim.timestep - g["setup_t_timestep"][1]) <= 1e-10*sim.timestep
    sim.main_loop(int(g["nsteps"][0]))
    dev = sim.device()
    assert dev.N == int(g["final_Nhydro"][0])
    assert abs(sim.t - g["final_t_timestep"][0]) <= 1e-11*sim.t
    assert np.max(np.abs(dev.download("r") - g["final_r"])) < 1e-10*np.abs(g["final_r"]).max()
    sk = dev.sinks()
    assert np.array_equal(sk["Ngas"], g["final_sink_Ngas"])
    assert np.max(np.abs(sk["menc"] - g["final_sink_menc"])/g["final_sink_menc"]) < 1e-9


def test_sinks_64k_vs_oracle():
    """sinks + block timesteps at a size where the exact-mode tree build leaves its LDS-only regime (cells of up to 65 536
    particles: block steps and closed-form chunks on the global arrays): the Boss-Bodenheimer cloud with 65 536 particles,
    Nlevels = 5, setup + 10 steps on the GPU and in the CPU restatement - particle count, the particle order the dead
    particles leave behind, sinkid, levels, potmin of the dense particles and the sinks' gas counts exact, sums to tolerance"""
    from gandalf_amd.host import Simulation
    from oracle.pyoracle import Oracle, NbodyOracle
    par = "%s/bb_sinks_8k_levels.dat" % PARAMS
    sim = Simulation(par, Nhydro=65536, run_id="BB64K")
    ic = sim.generate_ic()
    sim.post_ic_setup()
    sim.main_loop(10)
    p = read_params_file(par)
    p["Nhydro"] = "65536"
    o = Oracle(p, nthreads=16)
    o.set_particles(ic["r"], ic["m"], ic["h"], v=ic["v"], u=ic["u"])
    e = np.zeros(0)
    no = NbodyOracle(e.reshape(0, 3), e.reshape(0, 3), e, e, int(p["nbody_softening"]), float(p["nbody_mult"]))
    no.hybrid_setup(o, h_provided=True)
    no.hybrid_step(o, 10)
    dev = sim.device()
    assert dev.N == o.num_particles() < 65536
    assert abs(sim.t - o.t) <= 1e-11*o.t
    alive = o.get("m") > 0
    for k in ["level", "nstep", "nlast", "sinkid"]:
        assert np.array_equal(dev.download(k).astype(np.int64)[alive], o.get_int(k)[alive]), k
    assert np.array_equal(dev.download("m") == 0.0, ~alive)
    assert np.max(np.abs(dev.download("r") - o.get("r"))) < 1e-10*np.abs(o.get("r")).max()
    assert np.max(np.abs(dev.download("rho")[alive]/o.get("rho")[alive] - 1)) < 1e-9
    sk, so = dev.sinks(), o.sinks()
    assert len(sk["radius"]) == len(so["radius"]) >= 1
    assert np.array_equal(sk["Ngas"], so["Ngas"]) and np.array_equal(sk["istar"], so["istar"])
    assert np.max(np.abs(sk["menc"] - so["menc"])/so["menc"]) < 1e-9


def test_diag_and_timing_files(tmp_path):
    """<run_id>.diag (Simulation::CalculateDiagnostics / RecordDiagnostics, SimAnalysis.hpp:52-300) and <run_id>.timing
    (CodeTiming::ComputeTimingStatistics) from the host shell: the .diag line after the setup of the Boss-Bodenheimer case against
    the line the reference wrote for the same setup (tests/golden/bb_sinks_8k_setup.diag: six significant digits), and the
    layout of the timing table"""
    from gandalf_amd.host import Simulation
    sim = Simulation("%s/bb_sinks_8k.dat" % PARAMS)
    sim.setup()
    f = str(tmp_path/"BB.diag")
    d = sim.diagnostics(f)
    ours = [float(x) for x in open(f).read().split()]
    ref = [float(x) for x in open(os.path.join(os.path.dirname(__file__), "golden", "bb_sinks_8k_setup.diag")).read().split()]
    assert len(ours) == len(ref) == 29
    scale = {14: 0.3, 15: 0.3, 16: 0.3}                      # angular momentum components share the scale of the largest
    for i, (a, b) in enumerate(zip(ours, ref)):
        if i in (17, 18, 19, 20, 21, 22, 23, 24, 25):        # centre of mass / momentum: zero to rounding (com_frame = 1)
            assert abs(a) < 1e-12 and abs(b) < 1e-12, i
        elif i in (26, 27, 28):                              # net force: tree-force error, not conserved exactly
            assert abs(a - b) < 2e-9, (i, a, b)
        else:
            assert abs(a - b) <= 2e-6*max(abs(b), scale.get(i, 0.0)), (i, a, b)
    assert d["Nhydro"] == 8000 and d["Ndead"] == 0 and abs(d["Etot"] - (d["ketot"] + d["utot"] + d["gpetot"])) < 1e-14
    sim.main_loop(3)
    t = str(tmp_path/"BB.timing")
    sim.write_timing(t)
    txt = open(t).read()
    for key in ["Total simulation wall clock time", "Level : 1", "Block", "BUILD_TREE", "SPH_PROPERTIES", "SPH_ALL_FORCES", "REMAINDER"]:
        assert key in txt, key
