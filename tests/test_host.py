"""Host logic that needs no GPU: parameter files, the reference's RNG + IC generators (bit-exact against
the reference's own ICs), and the C ABI library: it loads and exports every symbol the header declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import PARAMS, ROOT, load_golden


def test_capi_exports_every_declared_symbol():
    import gandalf_amd
    from gandalf_amd import capi
    lib = gandalf_amd.load_library()
    header = open(os.path.join(ROOT, "include", "gandalf_hip.h")).read()
    declared = set(re.findall(r"\b(gh_[a-z_0-9]+)\s*\(", header))
    declared -= {"gh_ctx", "gh_config", "gh_stats"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), "libgandalf_hip.so does not export " + name
        assert name in capi.SYMBOLS, "ctypes binding misses " + name
    assert ctypes.sizeof(capi.Config) == 12*4 + 6*4 + 6*4 + 6*8 + 14*8 + 6*4 + 5*8      # ... + the sink block


def test_no_cpu_fallback_without_gpu():
    """without a HIP device the product path must fail loudly, not compute on the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gandalf_amd
    from gandalf_amd.params import read_params_file
    with pytest.raises(gandalf_amd.GhError):
        gandalf_amd.GandalfHip(read_params_file(os.path.join(PARAMS, "box3d_4k.dat")))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gandalf_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower(), os.path.join(dirpath, f)


def test_params_grammar_and_defaults():
    from gandalf_amd.host import Simulation
    sim = Simulation(os.path.join(PARAMS, "plummer_4k.dat"))
    assert sim.get_param("Nhydro") == "4096"
    assert float(sim.get_param("thetamaxsqd")) == 0.25
    assert sim.get_param("multipole") == "monopole"
    assert float(sim.get_param("h_converge")) == 0.01     # reference default, Parameters.cpp:256
    assert float(sim.get_param("energy_mult")) == 0.4
    # the reference's own file, with its "Comment : key = value" lines
    ref = os.path.join(os.path.dirname(PARAMS), "params", "box3d_4k.dat")
    sim2 = Simulation(ref, Nhydro=128)
    assert sim2.get_param("Nhydro") == "128" and sim2.get_param("boundary_lhs[0]") == "periodic"


def test_boss_bodenheimer_ic_bitwise():
    """ic = bb (BossBodenheimerIc.cpp:71-150): hexagonal lattice cut to a sphere, random Euler rotation (with the reference's
    single-precision acos), m = 2 azimuthal perturbation, solid-body rotation - the particles the reference generated"""
    from gandalf_amd.host import Simulation
    g = load_golden("bb_sinks_8k_steps")
    ic = Simulation(os.path.join(PARAMS, "bb_sinks_8k.dat")).generate_ic()
    assert np.array_equal(ic["r"], g["setup_r"]) and np.array_equal(ic["v"], g["setup_v"]) and np.array_equal(ic["m"], g["setup_m"])
    assert ic["initial_h_provided"]


@pytest.mark.parametrize("case", ["box3d_4k", "plummer_4k", "adsod_1d", "lattice3d_cubic_grav", "lattice3d_hex_grav"])
def test_ic_generators_bitwise(case):
    from gandalf_amd.host import Simulation
    g = load_golden(case + "_passes")
    ic = Simulation(os.path.join(PARAMS, case + ".dat")).generate_ic()
    assert np.array_equal(ic["r"], g["in_r"])
    assert np.array_equal(ic["m"], g["in_m"])
    assert np.array_equal(ic["u"], g["in_u"])
    assert np.array_equal(ic["v"], g["in_v"])
    assert ic["initial_h_provided"] == (case != "plummer_4k")


def test_physical_units_scale_factors():
    """SimUnits::SetupUnits restated in the host shell (SimUnits.cpp:825-1118): code units are pc and m_sun with G = 1 for the
    settings of the reference's bossbodenheimer.dat; every other scale follows from them - checked against the formulas
    evaluated here with the constants of Constants.h:34-51 - and the bb cloud the host generates carries them (angular
    velocity in rad/s, temperature in K -> u in code units)."""
    import numpy as np
    from gandalf_amd.host import Simulation
    here = os.path.dirname(os.path.abspath(__file__))
    sim = Simulation(os.path.join(here, "params", "bb_units_1600.dat"), Nhydro=1000)
    ic = sim.generate_ic()
    pc, msun, myr, G, mH, kB = 3.08568025E16, 1.98892E30, 3.1556952E13, 6.67384E-11, 1.66054E-27, 1.3806503E-23
    t_unit = pc**1.5/np.sqrt(msun*G)                       # seconds per code time
    want = {"r": 1.0, "m": 1.0, "t": t_unit/myr, "v": pc/t_unit/1000.0, "rho": msun/pc**3/1000.0, "u": (pc/t_unit)**2,
            "temp": mH*(pc/t_unit)**2/kB, "angvel": 1.0/t_unit}
    for k, v in want.items():
        assert abs(sim.unit_outscale(k) - v) <= 1e-14*abs(v), (k, sim.unit_outscale(k), v)
    assert abs(sim.unit_outscale("t") - 14.9) < 0.1        # a code time is ~15 Myr
    # the cloud: mass 1 m_sun, u = temp0/(gamma - 1)/mu_bar with temp0 = 10 K in code units, solid-body rotation at 1.6e-12 rad/s
    assert abs(ic["m"].sum() - 1.0) < 1e-12
    u0 = 10.0/want["temp"]/(5.0/3.0 - 1.0)/2.35
    assert np.max(np.abs(ic["u"]/u0 - 1)) < 1e-12
    r, v = ic["r"], ic["v"]
    w = 1.6e-12*t_unit                                     # code units; v = w x r about z in the frame before the COM shift
    vz = v[:, 2]
    assert np.max(np.abs(vz)) < 1e-12*np.abs(v).max()
    dv = v - v.mean(axis=0); dr = r - r.mean(axis=0)
    assert np.max(np.abs(dv[:, 0] + w*dr[:, 1])) < 1e-9*np.abs(v).max() and np.max(np.abs(dv[:, 1] - w*dr[:, 0])) < 1e-9*np.abs(v).max()
