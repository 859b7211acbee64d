"""ctypes binding of oracle/libgandalf_oracle.so (TEST INFRASTRUCTURE - see gandalf_oracle.cpp)."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgandalf_oracle.so")
_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)
_lib = None


class OrcParams(C.Structure):
    _fields_ = [("ndim", C.c_int32), ("Nleafmax", C.c_int32), ("self_gravity", C.c_int32), ("periodic", C.c_int32*3),
                ("energy_integration", C.c_int32), ("nthreads", C.c_int32),
                ("kernel", C.c_int32), ("multipole", C.c_int32), ("acond", C.c_int32), ("gravity_mac", C.c_int32), ("tdavisc", C.c_int32),
                ("Nlevels", C.c_int32), ("level_diff_max", C.c_int32), ("sph_single_timestep", C.c_int32),
                ("gas_eos", C.c_int32), ("ntreebuildstep", C.c_int32), ("ntreestockstep", C.c_int32), ("pad3_", C.c_int32),
                ("boxmin", C.c_double*3), ("boxmax", C.c_double*3), ("h_fac", C.c_double), ("h_converge", C.c_double),
                ("alpha_visc", C.c_double), ("beta_visc", C.c_double), ("gamma_eos", C.c_double), ("thetamaxsqd", C.c_double),
                ("courant_mult", C.c_double), ("accel_mult", C.c_double), ("energy_mult", C.c_double), ("macerror", C.c_double), ("alpha_visc_min", C.c_double),
                ("temp0", C.c_double), ("mu_bar", C.c_double), ("rho_bary", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("%s not built (make -C oracle)" % _PATH)
        L = C.CDLL(_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcParams)]
        L.orc_time.restype = C.c_double
        L.orc_timestep.restype = C.c_double
        L.orc_gather_neighbours.restype = C.c_long
        for name in ("orc_destroy", "orc_set_particles", "orc_get", "orc_set", "orc_set_time", "orc_time", "orc_timestep",
                     "orc_set_all_active", "orc_build_tree", "orc_tree_size", "orc_export_tree", "orc_density",
                     "orc_zero_accelerations", "orc_forces", "orc_setup", "orc_step", "orc_num_ghosts", "orc_gather_neighbours"):
            getattr(L, name)
        L.orc_nbody_create.restype = C.c_void_p
        L.orc_nbody_create.argtypes = [C.c_int, C.c_int, C.c_double, _PD, _PD, _PD, _PD]
        L.orc_nbody_time.restype = C.c_double
        L.orc_nbody_timestep.restype = C.c_double
        for name in ("orc_nbody_destroy", "orc_nbody_forces", "orc_nbody_setup", "orc_nbody_step"):
            getattr(L, name).argtypes = [C.c_void_p] + ([C.c_int] if name == "orc_nbody_step" else [])
        L.orc_nbody_time.argtypes = [C.c_void_p]
        L.orc_nbody_timestep.argtypes = [C.c_void_p]
        L.orc_nbody_get.argtypes = [C.c_void_p, C.c_int, _PD]
        L.orc_nbody_set.argtypes = [C.c_void_p, C.c_int, _PD]
        L.orc_hybrid_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_nbody_count.argtypes = [C.c_void_p]
        L.orc_nbody_get_scalar.argtypes = [C.c_void_p, C.c_int, _PD]
        L.orc_hybrid_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_set_stars.argtypes = [C.c_void_p, C.c_int, _PD, _PD, _PD, C.c_int]
        L.orc_star_gas_forces.argtypes = [C.c_void_p, _PD, _PD]
        _lib = L
    return _lib


VEC = {"r", "v", "a", "atree", "r0", "v0", "a0"}


class Oracle:
    def __init__(self, p, nthreads=None):
        """p: dict with the reference's parameter-file keys"""
        L = lib()
        q = OrcParams()
        q.ndim = int(p.get("ndim", 3))
        q.Nleafmax = int(p.get("Nleafmax", 6))
        q.self_gravity = int(p.get("self_gravity", 0))
        q.energy_integration = 1 if p.get("gas_eos", "energy_eqn") == "energy_eqn" else 0
        q.nthreads = nthreads or min(os.cpu_count() or 1, 16)
        for k in range(3):
            lhs, rhs = p.get("boundary_lhs[%d]" % k, "open"), p.get("boundary_rhs[%d]" % k, "open")
            # 1 = periodic (both faces); bit 1 / bit 2 = mirror wall at the lhs / rhs face
            q.periodic[k] = 1 if lhs == "periodic" else ((2 if lhs == "mirror" else 0) | (4 if rhs == "mirror" else 0))
            q.boxmin[k] = float(p.get("boxmin[%d]" % k, 0.0))
            q.boxmax[k] = float(p.get("boxmax[%d]" % k, 0.0))
        q.h_fac = float(p.get("h_fac", 1.2)); q.h_converge = float(p.get("h_converge", 0.01))
        q.alpha_visc = float(p.get("alpha_visc", 1.0)); q.beta_visc = float(p.get("beta_visc", 2.0))
        q.gamma_eos = float(p.get("gamma_eos", 1.66666666666666)); q.thetamaxsqd = float(p.get("thetamaxsqd", 0.1))
        q.courant_mult = float(p.get("courant_mult", 0.15)); q.accel_mult = float(p.get("accel_mult", 0.3))
        q.energy_mult = float(p.get("energy_mult", 0.4))
        assert p.get("kernel", "m4") in ("m4", "quintic")
        # bit 0: quintic, bit 1: tabulated_kernel
        q.kernel = (1 if p.get("kernel", "m4") == "quintic" else 0) | (2 if int(p.get("tabulated_kernel", 0)) else 0)
        assert p.get("multipole", "quadrupole") in ("monopole", "quadrupole", "fast_monopole", "fast_quadrupole") or not q.self_gravity
        q.multipole = {"monopole": 0, "quadrupole": 1, "fast_monopole": 2, "fast_quadrupole": 3}.get(p.get("multipole", "quadrupole"), 0)
        assert p.get("avisc", "mon97") == "mon97"
        q.acond = {"none": 0, "wadsley2008": 1, "price2008": 2}[p.get("acond", "none")]
        q.gravity_mac = {"geometric": 0, "gadget2": 1, "eigenmac": 2}[p.get("gravity_mac", "geometric")]
        q.macerror = float(p.get("macerror", 0.0001))
        assert p.get("time_dependent_avisc", "none") in ("none", "mm97", "cd2010")
        q.tdavisc = {"none": 0, "mm97": 1, "cd2010": 2}[p.get("time_dependent_avisc", "none")]
        q.alpha_visc_min = float(p.get("alpha_visc_min", 0.1))
        q.gas_eos = {"energy_eqn": 0, "isothermal": 1, "barotropic": 2}[p.get("gas_eos", "energy_eqn")]
        q.temp0 = float(p.get("temp0", 1.0)); q.mu_bar = float(p.get("mu_bar", 1.0)); q.rho_bary = float(p.get("rho_bary", 1.0e-14))
        q.ntreebuildstep = int(p.get("ntreebuildstep", 1))
        q.ntreestockstep = int(p.get("ntreestockstep", 1))
        q.Nlevels = int(p.get("Nlevels", 1)); q.level_diff_max = int(p.get("level_diff_max", 1))
        q.sph_single_timestep = int(p.get("sph_single_timestep", 0))
        self.L, self.ndim, self.N = L, q.ndim, 0
        self.h = C.c_void_p(L.orc_create(C.byref(q)))
        if int(p.get("sink_particles", 0)):
            mode = {"fixed": 0, "hmult": 1}.get(p.get("sink_radius_mode", "hmult"), 2)
            v = np.array([1, int(p.get("create_sinks", 0)), int(p.get("smooth_accretion", 0)), mode, int(p.get("Nsinkfixed", -1)),
                          float(p.get("rho_sink", 1.0e-12)), float(p.get("sink_radius", 2.0)), float(p.get("alpha_ss", 0.01)),
                          float(p.get("smooth_accrete_frac", 0.01)), float(p.get("smooth_accrete_dt", 0.01))])
            L.orc_set_sink_params(self.h, v.ctypes.data_as(_PD))

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @staticmethod
    def _dp(a):
        return a.ctypes.data_as(_PD) if a is not None else None

    def set_particles(self, r, m, h, v=None, u=None):
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1, self.ndim)
        self.N = r.shape[0]
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        v, m, h, u = c(v), c(m), c(h), c(u)
        self.L.orc_set_particles(self.h, self.N, self._dp(r), self._dp(v), self._dp(m), self._dp(h), self._dp(u))

    def num_particles(self):
        """Nhydro now (sink runs delete accreted particles at every tree build)"""
        self.N = int(self.L.orc_num_particles(self.h))
        return self.N

    def sinks(self):
        """SinkParticle records: dict of arrays (radius, mmax, menc, ..., angmom[3], istar, Ngas)"""
        n = int(self.L.orc_num_sinks(self.h))
        d, i = np.zeros((max(n, 1), 15)), np.zeros((max(n, 1), 2), dtype=np.int32)
        self.L.orc_get_sinks(self.h, self._dp(d), i.ctypes.data_as(C.c_void_p))
        d, i = d[:n], i[:n]
        names = ["radius", "mmax", "menc", "dmdt", "ketot", "gpetot", "rotketot", "utot", "taccrete", "trad", "trot", "tvisc"]
        out = {k: d[:, j].copy() for j, k in enumerate(names)}
        out["angmom"] = d[:, 12:15].copy(); out["istar"] = i[:, 0].copy(); out["Ngas"] = i[:, 1].copy()
        self.L.orc_mmean.restype = C.c_double
        out["mmean"] = float(self.L.orc_mmean(self.h))
        return out

    def get(self, name):
        self.num_particles()
        out = np.empty((self.N, self.ndim) if name in VEC else (self.N,))
        rc = self.L.orc_get(self.h, name.encode(), self._dp(out))
        assert rc == 0, name
        return out

    def set(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert self.L.orc_set(self.h, name.encode(), self._dp(arr)) == 0, name

    def get_int(self, name):
        """level / levelneib / nstep / nlast / flags (1 dead, 2 active, 4 end_timestep, 8 potmin) / sinkid / iorig"""
        self.num_particles()
        out = np.empty(self.N, dtype=np.int32)
        assert self.L.orc_get_int(self.h, name.encode(), out.ctypes.data_as(C.c_void_p)) == 0, name
        return out

    def set_int(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.int32)
        assert self.L.orc_set_int(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p)) == 0, name

    def set_block(self, n, nresync, level_max, level_step, dt_max):
        """block-timestep clock of Simulation (n, nresync, level_max, level_step, dt_max)"""
        v = np.array([n, nresync, level_max, level_step], dtype=np.int32)
        self.L.orc_set_block(self.h, v.ctypes.data_as(C.c_void_p), C.c_double(dt_max))

    def get_block(self):
        v = np.zeros(4, dtype=np.int32)
        self.L.orc_get_block.restype = C.c_double
        dt_max = self.L.orc_get_block(self.h, v.ctypes.data_as(C.c_void_p))
        return [int(x) for x in v], float(dt_max)

    def set_stars(self, r, m, h, nbody_softening=1):
        """stars of a hybrid gas + N-body run (their gravity acts on the gas in forces())"""
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1, self.ndim)
        self.Nstar = r.shape[0]
        m = np.ascontiguousarray(m, dtype=np.float64); h = np.ascontiguousarray(h, dtype=np.float64)
        self.L.orc_set_stars(self.h, self.Nstar, self._dp(r), self._dp(m), self._dp(h), int(nbody_softening))

    def star_gas_forces(self):
        """stars <- gas through the gas tree (UpdateAllStarGasForces): (a [Nstar][ndim], gpot [Nstar])"""
        a = np.zeros((self.Nstar, self.ndim)); g = np.zeros(self.Nstar)
        self.L.orc_star_gas_forces(self.h, self._dp(a), self._dp(g))
        return a, g

    def set_time(self, t, dt):
        self.L.orc_set_time(self.h, C.c_double(t), C.c_double(dt))

    @property
    def t(self):
        return self.L.orc_time(self.h)

    @property
    def timestep(self):
        return self.L.orc_timestep(self.h)

    def set_all_active(self):
        self.L.orc_set_all_active(self.h)

    def build_tree(self):
        self.L.orc_build_tree(self.h)

    def export_tree(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_tree_size(self.h, C.byref(a), C.byref(b), C.byref(c))
        nc, nd = a.value, self.ndim
        t = {"Ncell": nc, "ltot": b.value, "gtot": c.value}
        for k in ("level", "ifirst", "ilast", "N"):
            t[k] = np.zeros(nc, dtype=np.int32)
        t["inext"] = np.zeros(self.N, dtype=np.int32)
        for k in ("bbmin", "bbmax", "hboxmin", "hboxmax", "rcell", "com"):
            t[k] = np.zeros((nc, nd))
        for k in ("m", "rmax", "hmax", "cdistsqd"):
            t[k] = np.zeros(nc)
        ip = lambda x: x.ctypes.data_as(_PI)  # noqa: E731
        self.L.orc_export_tree(self.h, ip(t["level"]), ip(t["ifirst"]), ip(t["ilast"]), ip(t["N"]), ip(t["inext"]),
                               self._dp(t["bbmin"]), self._dp(t["bbmax"]), self._dp(t["hboxmin"]), self._dp(t["hboxmax"]),
                               self._dp(t["rcell"]), self._dp(t["com"]), self._dp(t["m"]), self._dp(t["rmax"]),
                               self._dp(t["hmax"]), self._dp(t["cdistsqd"]))
        return t

    def density(self):
        self.L.orc_density(self.h)

    def zero_accelerations(self):
        self.L.orc_zero_accelerations(self.h)

    def forces(self):
        self.L.orc_forces(self.h)

    def setup(self, h_provided=True):
        self.L.orc_setup(self.h, 1 if h_provided else 0)

    def step(self, n=1):
        self.L.orc_step(self.h, n)

    def num_ghosts(self):
        return int(self.L.orc_num_ghosts(self.h))

    def gather_neighbours(self):
        offs = np.zeros(self.N + 1, dtype=np.int64)
        tot = self.L.orc_gather_neighbours(self.h, offs.ctypes.data_as(C.POINTER(C.c_long)), None)
        ids = np.zeros(max(tot, 1), dtype=np.int32)
        self.L.orc_gather_neighbours(self.h, offs.ctypes.data_as(C.POINTER(C.c_long)), ids.ctypes.data_as(_PI))
        return offs, ids[:tot]


def smoke_check(sim, g):
    """used by __graft_entry__.smoke(): the HIP density against the CPU restatement on the same inputs"""
    from gandalf_amd.params import read_params_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    o = Oracle(read_params_file(os.path.join(root, "tests", "params", "plummer_4k.dat")))
    o.set_particles(g["in_r"], g["in_m"], g["in_h"], v=g["in_v"], u=g["in_u"])
    o.build_tree()
    o.density()
    e = np.max(np.abs(sim.download("rho") - o.get("rho"))/o.get("rho"))
    assert e < 1e-12, e
    return "(vs CPU restatement: rho %.1e)" % e


class NbodyOracle:
    """Stars: direct-sum forces + leapfrog KDK (reference Nbody.cpp:233-287, NbodyLeapfrogKDK.cpp:78-400)."""
    FIELDS = {"r": 0, "v": 1, "a": 2, "adot": 3, "gpot": 4, "r0": 5, "v0": 6, "a0": 7}

    def __init__(self, r, v, m, h, softening, nbody_mult=0.1):
        """no stars (a sink run before its first sink): r = v = m = h = empty arrays"""
        self.L = lib()
        self.N = len(m)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (r, v, m, h)]
        self.o = self.L.orc_nbody_create(self.N, int(softening), float(nbody_mult), *[x.ctypes.data_as(_PD) for x in a])

    def __del__(self):
        if getattr(self, "o", None):
            self.L.orc_nbody_destroy(self.o)
            self.o = None

    def forces(self):
        self.L.orc_nbody_forces(self.o)

    def setup(self):
        self.L.orc_nbody_setup(self.o)

    def step(self, n=1):
        self.L.orc_nbody_step(self.o, int(n))

    def t(self):
        return self.L.orc_nbody_time(self.o)

    def timestep(self):
        return self.L.orc_nbody_timestep(self.o)

    SET_FIELDS = {"r": 0, "v": 1, "a": 2, "adot": 3, "gpot": 4, "r0": 5, "v0": 6, "a0": 7, "adot0": 8, "dt": 9, "tlast": 10}

    def set(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self.L.orc_nbody_set(self.o, self.SET_FIELDS[name], arr.ctypes.data_as(_PD))

    def hybrid_setup(self, gas, h_provided=False):
        """PostInitialConditionsSetup of a hybrid run: `gas` is the Oracle holding the gas IC, self the star IC"""
        self.L.orc_hybrid_setup(gas.h, self.o, 1 if h_provided else 0)

    def hybrid_step(self, gas, n=1):
        """n MainLoop calls of a hybrid run: `gas` is the Oracle holding the gas, self the stars"""
        self.L.orc_hybrid_step(gas.h, self.o, int(n))

    SCALARS = {"m": 0, "h": 1, "dt_internal": 2, "invh": 3, "gpot": 4, "dt": 5, "tlast": 6}

    def get(self, name):
        self.N = int(self.L.orc_nbody_count(self.o))          # sinks add stars
        if name in self.SCALARS:
            out = np.empty(self.N)
            self.L.orc_nbody_get_scalar(self.o, self.SCALARS[name], out.ctypes.data_as(_PD))
            return out
        out = np.empty((self.N, 3))
        self.L.orc_nbody_get(self.o, self.FIELDS[name], out.ctypes.data_as(_PD))
        return out
