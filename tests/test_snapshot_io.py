"""Snapshot formats: the host shell's `column`, SEREN-unformatted (`su`) and SEREN-formatted (`sf`)
readers / writers against files written by the reference's own writers (tests/golden/snapshots/*.column, *.su, *.sf:
`ref_dump snap`, i.e. Simulation::WriteColumnSnapshotFile / WriteSerenUnformSnapshotFile / WriteSerenFormSnapshotFile on a
3-D box after setup and on the 1-D shock tube after two steps; scripts/make_golden.py snapshots) and the state the
reference held when it wrote them (*_state.npz)."""
import filecmp
import os

import numpy as np
import pytest

from gandalf_amd.host import HostError, read_snapshot, write_snapshot

SNAP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "snapshots")


def state(name):
    g = np.load(os.path.join(SNAP, name + "_state.npz"))
    t, tsnaplast, mmean, tlite, h_fac = g["snap_t_tsnaplast_mmean_tlitesnaplast_hfac"]
    nout, nsteps, nlite = [int(x) for x in g["snap_Noutsnap_Nsteps_Noutlitesnap"]]
    s = {k: g[k] for k in ("r", "v", "m", "h", "rho", "u", "iorig")}
    s.update(t=float(t), tsnaplast=float(tsnaplast), mmean=float(mmean), tlitesnaplast=float(tlite), h_fac=float(h_fac),
             Noutsnap=nout, Nsteps=nsteps, Noutlitesnap=nlite)
    return s


@pytest.mark.parametrize("name", ["box", "sod"])
def test_su_reader_returns_the_reference_state(name):
    s = state(name)
    f = read_snapshot(os.path.join(SNAP, name + ".su"), "su")
    assert f["N"] == len(s["m"]) and f["ndim"] == s["r"].reshape(len(s["m"]), -1).shape[1]
    assert f["t"] == s["t"] and f["Nsteps"] == s["Nsteps"] and f["mmean"] == s["mmean"] and f["h_fac"] == s["h_fac"]
    for k in ("r", "v"):
        assert np.array_equal(f[k], s[k].reshape(f["N"], -1)), k
    for k in ("m", "h", "rho", "u", "iorig"):
        assert np.array_equal(f[k], s[k]), k


@pytest.mark.parametrize("name", ["box", "sod"])
@pytest.mark.parametrize("form", ["su", "column", "sf"])
def test_writer_is_byte_identical_to_the_reference(name, form, tmp_path):
    out = str(tmp_path / ("%s.%s" % (name, form)))
    write_snapshot(out, form, state(name))
    assert filecmp.cmp(out, os.path.join(SNAP, "%s.%s" % (name, form)), shallow=False)


@pytest.mark.parametrize("name", ["box", "sod"])
def test_column_reader(name):
    s = state(name)
    f = read_snapshot(os.path.join(SNAP, name + ".column"), "column")
    assert f["N"] == len(s["m"])
    assert abs(f["t"] - s["t"]) <= 1e-5*abs(s["t"])
    for k in ("r", "v", "m", "h", "rho", "u"):                 # the format keeps 6 significant digits
        a, b = f[k].reshape(f["N"], -1), s[k].reshape(f["N"], -1)
        assert np.all(np.abs(a - b) <= 5e-6*np.abs(b) + 1e-300), k


@pytest.mark.parametrize("name", ["box", "sod"])
def test_sf_reader(name):
    """the formatted SEREN file keeps 11 significant digits (scientific, 10 decimals); header words and ids exact"""
    s = state(name)
    f = read_snapshot(os.path.join(SNAP, name + ".sf"), "seren_form")
    assert f["N"] == len(s["m"]) and f["Nsteps"] == s["Nsteps"] and np.array_equal(f["iorig"], s["iorig"])
    assert abs(f["t"] - s["t"]) <= 1e-10*abs(s["t"]) and abs(f["h_fac"] - s["h_fac"]) <= 1e-10*s["h_fac"]
    for k in ("r", "v", "m", "h", "rho", "u"):
        a, b = f[k].reshape(f["N"], -1), s[k].reshape(f["N"], -1)
        assert np.all(np.abs(a - b) <= 1e-10*np.abs(b) + 1e-300), k


def test_round_trip_and_errors(tmp_path):
    s = state("box")
    out = str(tmp_path / "rt.su")
    write_snapshot(out, "seren_unform", s)
    f = read_snapshot(out, "su")
    assert np.array_equal(f["r"], s["r"]) and np.array_equal(f["u"], s["u"])
    with pytest.raises(HostError):
        write_snapshot(out, "slite", s)                        # the lite format is not built
    with pytest.raises(HostError):
        read_snapshot(os.path.join(SNAP, "box.su"), "sf")      # wrong tag
    with pytest.raises(HostError):
        read_snapshot(os.path.join(SNAP, "box.column"), "su")  # wrong tag
    with pytest.raises(HostError):
        read_snapshot(str(tmp_path / "missing.su"), "su")
