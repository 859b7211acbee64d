"""ctypes binding of libgandalf_host.so: the C++ host shell (parameter files, IC generators,
SetupSimulation / MainLoop) that sits above the C ABI.  No logic of its own."""
import ctypes as C
import os

import numpy as np

from . import capi

HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libgandalf_host.so")
_PD = C.POINTER(C.c_double)
_H = C.c_void_p
_hlib = None

HOST_SYMBOLS = {
    "gah_create": (_H, []),
    "gah_destroy": (None, [_H]),
    "gah_last_error": (C.c_char_p, [_H]),
    "gah_read_params": (C.c_int, [_H, C.c_char_p]),
    "gah_set_param": (C.c_int, [_H, C.c_char_p, C.c_char_p]),
    "gah_get_param": (C.c_int, [_H, C.c_char_p, C.c_char_p, C.c_int]),
    "gah_generate_ic": (C.c_int, [_H]),
    "gah_num_particles": (C.c_int, [_H]),
    "gah_initial_h_provided": (C.c_int, [_H]),
    "gah_get_ic": (C.c_int, [_H, _PD, _PD, _PD, _PD, _PD]),
    "gah_post_ic_setup": (C.c_int, [_H]),
    "gah_init_comm": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "gah_run": (C.c_int, [_H, C.c_int]),
    "gah_set_restart": (C.c_int, [_H, C.c_int]),
    "gah_set_output": (C.c_int, [_H, C.c_int]),
    "gah_nsteps": (C.c_int, [_H]),
    "gah_unit_outscale": (C.c_double, [_H, C.c_char_p]),
    "gah_noutsnap": (C.c_int, [_H]),
    "gah_upload_ic": (C.c_int, [_H]),
    "gah_setup": (C.c_int, [_H]),
    "gah_main_loop": (C.c_int, [_H, C.c_int]),
    "gah_time": (C.c_double, [_H]),
    "gah_timestep": (C.c_double, [_H]),
    "gah_ctx": (C.c_void_p, [_H]),
    "gah_write_snapshot": (C.c_int, [_H, C.c_char_p, C.c_char_p]),
    "gah_diagnostics": (C.c_int, [_H, _PD, C.c_char_p]),
    "gah_write_timing": (C.c_int, [_H, C.c_char_p]),
    "gah_snapshot_error": (C.c_char_p, []),
    "gah_snapshot_write": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double, _PD, _PD, _PD, _PD, _PD, _PD,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_long), _PD]),
    "gah_snapshot_open": (C.c_void_p, [C.c_char_p, C.c_char_p]),
    "gah_snapshot_info": (None, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _PD, C.POINTER(C.c_long), _PD]),
    "gah_snapshot_data": (None, [C.c_void_p, _PD, _PD, _PD, _PD, _PD, _PD, C.POINTER(C.c_int32)]),
    "gah_snapshot_close": (None, [C.c_void_p]),
}


def load_host_library(path=HOST_LIB_PATH):
    global _hlib
    if _hlib is not None:
        return _hlib
    if not os.path.exists(path):
        raise ImportError("%s not built (run __graft_entry__.build())" % path)
    capi.load_library()          # libgandalf_hip.so first (the host library links against it)
    lib = C.CDLL(path)
    for name, (res, args) in HOST_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _hlib = lib
    return lib


class HostError(RuntimeError):
    pass


def _dp(a):
    return a.ctypes.data_as(_PD)


def write_snapshot(filename, fileform, snap):
    """GANDALF snapshot file (fileform = column | su) from a dict with r, v [N][ndim], m, h, rho, u [N], t and, for su,
    optionally iorig and the header words Noutsnap, Nsteps, Noutlitesnap, tsnaplast, mmean, tlitesnaplast, h_fac
    (gandalf_amd/host/SnapshotIO.cpp; reference SimulationIO.hpp:444-540, 2009-2254)"""
    lib = load_host_library()
    c = lambda k: np.ascontiguousarray(snap[k], dtype=np.float64)  # noqa: E731
    r = c("r")
    N = len(c("m"))
    r = r.reshape(N, -1)
    v = c("v").reshape(N, -1)
    io = np.ascontiguousarray(snap.get("iorig", np.arange(N)), dtype=np.int32)
    hl = np.array([snap.get("Noutsnap", 0), snap.get("Nsteps", 0), snap.get("Noutlitesnap", 0)], dtype=np.int64)
    hd = np.array([snap.get("tsnaplast", 0.0), snap.get("mmean", 0.0), snap.get("tlitesnaplast", 0.0), snap.get("h_fac", 1.2)])
    rc = lib.gah_snapshot_write(filename.encode(), fileform.encode(), r.shape[1], N, float(snap["t"]), _dp(r), _dp(v), _dp(c("m")),
                                _dp(c("h")), _dp(c("rho")), _dp(c("u")), io.ctypes.data_as(C.POINTER(C.c_int32)),
                                hl.ctypes.data_as(C.POINTER(C.c_long)), _dp(hd))
    if rc:
        raise HostError(lib.gah_snapshot_error().decode())


def read_snapshot(filename, fileform):
    """inverse of write_snapshot: dict with ndim, N, t, r, v, m, h, rho, u, iorig and the su header words"""
    lib = load_host_library()
    h = lib.gah_snapshot_open(filename.encode(), fileform.encode())
    if not h:
        raise HostError(lib.gah_snapshot_error().decode())
    try:
        nd, N, t = C.c_int(), C.c_int(), C.c_double()
        hl = np.zeros(3, dtype=np.int64)
        hd = np.zeros(4)
        lib.gah_snapshot_info(h, C.byref(nd), C.byref(N), C.byref(t), hl.ctypes.data_as(C.POINTER(C.c_long)), _dp(hd))
        out = {"ndim": nd.value, "N": N.value, "t": t.value, "r": np.zeros((N.value, nd.value)), "v": np.zeros((N.value, nd.value)),
               "iorig": np.zeros(N.value, dtype=np.int32)}
        for k in ("m", "h", "rho", "u"):
            out[k] = np.zeros(N.value)
        lib.gah_snapshot_data(h, _dp(out["r"]), _dp(out["v"]), _dp(out["m"]), _dp(out["h"]), _dp(out["rho"]), _dp(out["u"]),
                              out["iorig"].ctypes.data_as(C.POINTER(C.c_int32)))
        out.update(Noutsnap=int(hl[0]), Nsteps=int(hl[1]), Noutlitesnap=int(hl[2]), tsnaplast=float(hd[0]), mmean=float(hd[1]),
                   tlitesnaplast=float(hd[2]), h_fac=float(hd[3]))
        return out
    finally:
        lib.gah_snapshot_close(h)


class Simulation:
    """GANDALF-style driver: Simulation(paramfile, **overrides); .generate_ic(); .setup(); .main_loop(n)."""

    def __init__(self, paramfile=None, **overrides):
        self.lib = load_host_library()
        self.h = self.lib.gah_create()
        if paramfile is not None:
            self._chk(self.lib.gah_read_params(self.h, paramfile.encode()))
        for k, v in overrides.items():
            self.set_param(k, v)

    def _chk(self, rc):
        if rc:
            raise HostError(self.lib.gah_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.gah_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_param(self, key, value):
        self._chk(self.lib.gah_set_param(self.h, str(key).encode(), str(value).encode()))

    def get_param(self, key):
        buf = C.create_string_buffer(256)
        self._chk(self.lib.gah_get_param(self.h, key.encode(), buf, 256))
        return buf.value.decode()

    def generate_ic(self):
        """ProcessParameters + GenerateIC (+ SetComFrame, + InitialSmoothingLengthGuess); host only."""
        self._chk(self.lib.gah_generate_ic(self.h))
        n = self.lib.gah_num_particles(self.h)
        nd = int(self.get_param("ndim"))
        ic = {"r": np.empty((n, nd)), "v": np.empty((n, nd)), "m": np.empty(n), "h": np.empty(n), "u": np.empty(n)}
        dp = lambda a: a.ctypes.data_as(_PD)  # noqa: E731
        self._chk(self.lib.gah_get_ic(self.h, dp(ic["r"]), dp(ic["v"]), dp(ic["m"]), dp(ic["h"]), dp(ic["u"])))
        ic["initial_h_provided"] = bool(self.lib.gah_initial_h_provided(self.h))
        return ic

    def upload_ic(self):
        self._chk(self.lib.gah_upload_ic(self.h))

    def initial_h_provided(self):
        return bool(self.lib.gah_initial_h_provided(self.h))

    def init_comm(self, rank, nranks, ops_ptr):
        """multi-GPU: rank, size and the collectives (address of a gh_comm_ops, multigpu.CommOps) - before the setup"""
        self._chk(self.lib.gah_init_comm(self.h, rank, nranks, ops_ptr))

    def post_ic_setup(self):
        self._chk(self.lib.gah_post_ic_setup(self.h))

    def setup(self):
        self._chk(self.lib.gah_setup(self.h))

    def write_snapshot(self, filename, fileform="su"):
        """SimulationBase::WriteSnapshotFile: the current device state as a column / su snapshot"""
        self._chk(self.lib.gah_write_snapshot(self.h, filename.encode(), fileform.encode()))

    DIAG = ["t", "Nsteps", "timestep", "dt_min_hydro", "dt_min_nbody", "level_max", "Nhydro", "Nstar", "Ndead", "mtot", "Etot", "ketot",
            "utot", "gpetot"]

    def diagnostics(self, filename=None):
        """Simulation::CalculateDiagnostics (SimAnalysis.hpp:52-200); with a filename also appends the <run_id>.diag line"""
        o = np.zeros(29)
        self._chk(self.lib.gah_diagnostics(self.h, o.ctypes.data_as(_PD), filename.encode() if filename else None))
        d = {k: o[i] for i, k in enumerate(self.DIAG)}
        d.update(angmom=o[14:17].copy(), rcom=o[17:20].copy(), vcom=o[20:23].copy(), mom=o[23:26].copy(), force=o[26:29].copy())
        return d

    def write_timing(self, filename):
        """CodeTiming::ComputeTimingStatistics: the <run_id>.timing table (device phases)"""
        self._chk(self.lib.gah_write_timing(self.h, filename.encode()))

    def main_loop(self, nsteps=1):
        self._chk(self.lib.gah_main_loop(self.h, nsteps))

    def run(self, nsteps=-1):
        """SimulationBase::Run: MainLoop + Output until tend / Nstepsmax (or nsteps more steps)"""
        self._chk(self.lib.gah_run(self.h, nsteps))

    def set_output(self, on=True):
        """regular snapshots <run_id>.<out_file_form>.NNNNN + <run_id>.restart in the working directory (before setup())"""
        self.lib.gah_set_output(self.h, 1 if on else 0)

    def set_restart(self, on=True):
        """continue from the snapshot named in <run_id>.restart (the reference's `gandalf -r`); before setup()"""
        self.lib.gah_set_restart(self.h, 1 if on else 0)

    def unit_outscale(self, quantity):
        """SimUnits: x[code units] = x[output units] / scale, for r, m, t, v, a, rho, u, temp, angvel (1 when dimensionless)"""
        return self.lib.gah_unit_outscale(self.h, quantity.encode())

    @property
    def Nsteps(self):
        return self.lib.gah_nsteps(self.h)

    @property
    def Noutsnap(self):
        return self.lib.gah_noutsnap(self.h)

    @property
    def t(self):
        return self.lib.gah_time(self.h)

    @property
    def timestep(self):
        return self.lib.gah_timestep(self.h)

    def device(self):
        """GandalfHip view (no ownership) of the device context of this simulation."""
        ctx = self.lib.gah_ctx(self.h)
        if not ctx:
            raise HostError("simulation has no device context yet (call setup())")
        return capi.GandalfHip.borrow(ctx, int(self.get_param("ndim")), int(self.get_param("self_gravity")))
