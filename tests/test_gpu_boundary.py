"""The drop-in boundary EXECUTED: the reference's own GradhSphSimulation (its IC generator, PostInitialConditionsSetup,
MainLoop, SphLeapfrogKDK, timestep code - compiled from /root/reference by oracle/ref.mk) running with
`sphneib = new HipSphTree<ndim>` (include/reference_shell/HipSphTree.h), i.e. every BuildTree / UpdateAllSphProperties /
UpdateAllSph(Hydro)Forces / GetGatherNeighbourList call of the reference lands in libgandalf_hip.so through the C ABI.

oracle/_ref/ref_hipshell is ref_dump built with -DREF_HIPSHELL (test infrastructure; it travels to the GPU box as a binary).
Its dumps are compared with the fixtures the same driver wrote from the unmodified reference (tests/golden/*_steps.npz)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from test_gpu_parity import relerr, vec_err

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_hipshell")


def run_shell(tmp_path, case, nsteps, mode="hipsteps"):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/ref_hipshell not built (make -f oracle/ref.mk hipshell needs /root/reference)")
    sys.path.insert(0, ROOT)
    from oracle.gdmp import read_gdmp
    prefix = str(tmp_path/"s")
    par = os.path.join(ROOT, "tests", "params", case + ".dat")
    out = subprocess.run([EXE, mode, par, prefix, str(nsteps)], cwd=str(tmp_path), capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "no HIP device" not in out.stdout + out.stderr
    return read_gdmp(prefix + "_setup.gdmp"), read_gdmp(prefix + "_final.gdmp")


@pytest.mark.parametrize("case", ["plummer_4k", "box3d_4k", "adsod_1d"])
def test_reference_mainloop_on_hip_tree(case, tmp_path):
    """setup (the whole PostInitialConditionsSetup from the raw IC) + the fixture's number of MainLoop steps"""
    g = load_golden(case + "_steps")
    nsteps = int(g["nsteps"][0])
    setup, final = run_shell(tmp_path, case, nsteps)
    assert np.array_equal(setup["iorig"], g["setup_iorig"])
    # after the setup: smoothing lengths iterated from the reference's initial guess on the device, forces, first timestep
    assert relerr(setup["h"], g["setup_h"]) < 1e-10
    assert relerr(setup["rho"], g["setup_rho"]) < 1e-10
    assert vec_err(setup["a"], g["setup_a"]) < 1e-9
    assert abs(setup["t_timestep"][1] - g["setup_t_timestep"][1]) <= 1e-9*abs(g["setup_t_timestep"][1])
    # after the steps (the reference's integrator on the host, every force from the device)
    tf, dtf = g["final_t_timestep"]
    assert abs(final["t_timestep"][0] - tf) <= 1e-11*abs(tf)
    assert abs(final["t_timestep"][1] - dtf) <= 1e-8*abs(dtf)
    assert np.max(np.abs(final["r"] - g["final_r"])) < 1e-10*np.abs(g["final_r"]).max()
    assert np.max(np.abs(final["v"] - g["final_v"])) < 1e-9*max(np.abs(g["final_v"]).max(), 1e-3)
    assert relerr(final["h"], g["final_h"]) < 1e-9
    assert relerr(final["rho"], g["final_rho"]) < 1e-9
    assert vec_err(final["a"], g["final_a"]) < 1e-8
    assert relerr(final["u"], g["final_u"]) < 1e-9
    # the reference's point query (Sinks, IC regularisation: Tree::ComputeGatherNeighbourList(rp, rsearch), Tree.cpp:208-280)
    # answered by gh_gather_neighbours_at - the driver asks for kernrange*h_i around every particle; against brute force
    if "gather_offsets" in setup:
        off, ids = setup["gather_offsets"], setup["gather_ids"]
        r, h = setup["r"].reshape(len(setup["h"]), -1), setup["h"]
        box = 1.0 if case.startswith("box3d") else 0.0          # periodic unit cube: minimum image
        for i in range(0, len(h), 97):
            dx = r - r[i]
            if box:
                dx -= box*np.round(dx/box)
            d2 = np.sum(dx*dx, axis=1)
            want = set(np.nonzero(d2 <= (2.0*h[i])**2)[0].tolist())
            got = set(ids[off[i]:off[i + 1]].tolist())
            edge = {j for j in want ^ got if abs(d2[j] - (2.0*h[i])**2) > 1e-9*d2[j]}
            assert not edge, (i, sorted(edge))


@pytest.mark.parametrize("case", ["plummer_4k_levels", "box3d_4k_levels", "adsod_1d_ts3_levels", "plummer_4k_tb4"])
def test_reference_mainloop_on_hip_tree_levels_and_schedules(case, tmp_path):
    """Block timesteps (Nlevels = 5) and the tree schedule through the seam: the reference's integrator decides levels and
    active flags on the host (they go down before every pass: HipSphTree::UploadLevels / UpdateActiveParticleCounters),
    the device raises levelneib (comes back after the force pass), HydroTree::BuildTree's rebuild / re-stock / extrapolate
    schedule reaches the device with its own arguments (gh_build_tree_scheduled; adsod_1d_ts3_levels: ntreebuildstep = 8,
    ntreestockstep = 3; plummer_4k_tb4: ntreebuildstep = 4).  40 (10) MainLoop steps against the reference's own run."""
    g = load_golden(case + "_steps")
    nsteps = int(g["nsteps"][0])
    setup, final = run_shell(tmp_path, case, nsteps)
    if "final_level" in g:
        for k in ("level", "nstep", "nlast"):
            assert np.array_equal(final[k], g["final_" + k]), k
        assert np.array_equal(final["n_Nsteps_nresync"], g["final_n_Nsteps_nresync"])
    tf, dtf = g["final_t_timestep"]
    assert abs(final["t_timestep"][0] - tf) <= 1e-11*abs(tf)
    assert np.max(np.abs(final["r"] - g["final_r"])) < 1e-9*np.abs(g["final_r"]).max()
    assert relerr(final["h"], g["final_h"]) < 1e-8
    assert relerr(final["rho"], g["final_rho"]) < 1e-8
    assert vec_err(final["a"], g["final_a"]) < 1e-7
    assert relerr(final["u"], g["final_u"]) < 1e-8
