"""ctypes binding of include/gandalf_hip.h (one-to-one; no logic of its own)."""
import ctypes as C
import os

import numpy as np

# GANDALF_HIP_LIB: another build of the same library (A/B timing of kernel variants); never a different implementation
LIB_PATH = os.environ.get("GANDALF_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libgandalf_hip.so")

GH_T_NAMES = ["BUILD_TREE", "SPH_PROPERTIES", "SPH_FORCES", "KDK", "GRAV_WALK"]

FIELDS = {name: i for i, name in enumerate(
    ["r", "v", "a", "atree", "r0", "v0", "a0",
     "m", "h", "u", "u0", "dudt", "dudt0", "rho", "invomega", "zeta", "hfactor", "hrangesqd", "sound",
     "pressure", "div_v", "gpot", "gpot_hydro", "alpha", "dalphadt", "dt", "dt_next", "tlast",
     "level", "levelneib", "nstep", "nlast", "flags",        # block timesteps: integers carried as doubles
     "sinkid"])}                                             # sink runs (flags then: 1 active, 2 end_timestep, 4 dead, 8 potmin)
VECTOR_FIELDS = {"r", "v", "a", "atree", "r0", "v0", "a0"}


class GhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gandalf_hip error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32), ("kernel", C.c_int32), ("gas_eos", C.c_int32), ("avisc", C.c_int32),
        ("acond", C.c_int32), ("self_gravity", C.c_int32), ("hydro_forces", C.c_int32),
        ("multipole", C.c_int32), ("gravity_mac", C.c_int32), ("Nleafmax", C.c_int32),
        ("energy_integration", C.c_int32), ("device", C.c_int32),
        ("boundary_lhs", C.c_int32 * 3), ("boundary_rhs", C.c_int32 * 3),
        ("Nlevels", C.c_int32), ("level_diff_max", C.c_int32), ("ntreebuildstep", C.c_int32), ("ntreestockstep", C.c_int32),
        ("sph_single_timestep", C.c_int32), ("reserved_", C.c_int32),
        ("boxmin", C.c_double * 3), ("boxmax", C.c_double * 3),
        ("h_fac", C.c_double), ("h_converge", C.c_double), ("alpha_visc", C.c_double),
        ("beta_visc", C.c_double), ("gamma_eos", C.c_double), ("temp0", C.c_double),
        ("mu_bar", C.c_double), ("rho_bary", C.c_double), ("thetamaxsqd", C.c_double),
        ("courant_mult", C.c_double), ("accel_mult", C.c_double), ("energy_mult", C.c_double),
        ("macerror", C.c_double), ("alpha_visc_min", C.c_double),
        ("sink_particles", C.c_int32), ("create_sinks", C.c_int32), ("smooth_accretion", C.c_int32),
        ("sink_radius_mode", C.c_int32), ("Nsinkfixed", C.c_int32), ("reserved2_", C.c_int32),
        ("rho_sink", C.c_double), ("sink_radius", C.c_double), ("alpha_ss", C.c_double),
        ("smooth_accrete_frac", C.c_double), ("smooth_accrete_dt", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [("n_particles", C.c_int64), ("n_iterations", C.c_int64), ("n_candidates", C.c_int64),
                ("n_direct", C.c_int64), ("n_cells", C.c_int64), ("n_retries", C.c_int64),
                ("kernel_ms", C.c_double), ("n_leaf_cells", C.c_int64), ("n_leaf_direct", C.c_int64), ("n_leaf_cand", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)
_PL = C.POINTER(C.c_int64)
_CTX = C.c_void_p

# every symbol include/gandalf_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "gh_create": (C.c_int, [C.POINTER(Config), C.POINTER(_CTX)]),
    "gh_destroy": (None, [_CTX]),
    "gh_last_error": (C.c_char_p, [_CTX]),
    "gh_upload_particles": (C.c_int, [_CTX, C.c_int64, _PD, _PD, _PD, _PD, _PD]),
    "gh_upload_field": (C.c_int, [_CTX, C.c_int, _PD]),
    "gh_download": (C.c_int, [_CTX, C.c_int, _PD]),
    "gh_num_particles": (C.c_int64, [_CTX]),
    "gh_build_tree": (C.c_int, [_CTX]),
    "gh_build_tree_scheduled": (C.c_int, [_CTX, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]),
    "gh_tree_size": (C.c_int, [_CTX, _PI, _PI, _PI]),
    "gh_export_tree": (C.c_int, [_CTX, _PI, _PI, _PI, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PI]),
    "gh_update_density": (C.c_int, [_CTX, C.POINTER(Stats)]),
    "gh_zero_accelerations": (C.c_int, [_CTX]),
    "gh_update_hydro_forces": (C.c_int, [_CTX, C.POINTER(Stats)]),
    "gh_update_all_forces": (C.c_int, [_CTX, C.POINTER(Stats)]),
    "gh_kdk_advance": (C.c_int, [_CTX, C.c_int, C.c_double, C.c_double]),
    "gh_compute_global_timestep": (C.c_int, [_CTX, _PD]),
    "gh_kdk_end": (C.c_int, [_CTX, C.c_int, C.c_double, C.c_double]),
    "gh_setup": (C.c_int, [_CTX, C.c_int, _PD]),
    "gh_set_time": (C.c_int, [_CTX, C.c_double, C.c_double]),
    "gh_step": (C.c_int, [_CTX, C.c_int, _PD, _PD]),
    "gh_set_stars": (C.c_int, [_CTX, C.c_int64, _PD, _PD, _PD, C.c_int]),
    "gh_star_gas_forces": (C.c_int, [_CTX, _PD, _PD]),
    "gh_set_block_clock": (C.c_int, [_CTX, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]),
    "gh_get_block_clock": (C.c_int, [_CTX, _PI, _PD]),
    "gh_get_active_count": (C.c_int, [_CTX, _PL, C.c_int]),
    "gh_gather_neighbours": (C.c_int, [_CTX, C.c_int64, _PL, _PI]),
    "gh_gather_neighbours_at": (C.c_int, [_CTX, _PD, C.c_double, _PI, C.c_int32]),
    "gh_get_timers": (C.c_int, [_CTX, _PD, C.POINTER(Stats), C.POINTER(Stats)]),
    "gh_reset_timers": (C.c_int, [_CTX]),
    "gh_comm_init": (C.c_int, [_CTX, C.c_int, C.c_int, C.c_void_p]),
    "gh_exchange_halo": (C.c_int, [_CTX, C.c_int]),
    "gh_allgather_multipoles": (C.c_int, [_CTX]),
    "gh_comm_info": (C.c_int, [_CTX, _PL, _PL, _PL]),
    "gh_field_dev": (C.c_void_p, [_CTX, C.c_int, C.c_int]),
    "gh_stream": (C.c_void_p, [_CTX]),
    "gh_update_hmax": (C.c_int, [_CTX]),
    "gh_nbody_create": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(_CTX)]),
    "gh_nbody_destroy": (None, [_CTX]),
    "gh_nbody_last_error": (C.c_char_p, [_CTX]),
    "gh_nbody_upload": (C.c_int, [_CTX, C.c_int64, _PD, _PD, _PD, _PD]),
    "gh_nbody_download": (C.c_int, [_CTX, C.c_int, _PD]),
    "gh_nbody_forces": (C.c_int, [_CTX]),
    "gh_nbody_setup": (C.c_int, [_CTX, _PD]),
    "gh_nbody_step": (C.c_int, [_CTX, C.c_int, _PD, _PD]),
    "gh_nbody_upload_field": (C.c_int, [_CTX, C.c_int, _PD]),
    "gh_hybrid_step": (C.c_int, [_CTX, _CTX, C.c_int, _PD, _PD]),
    "gh_get_sinks": (C.c_int, [_CTX, _PI, _PD, _PI]),
    "gh_nbody_num_stars": (C.c_int64, [_CTX]),
    "gh_nbody_download_scalar": (C.c_int, [_CTX, C.c_int, _PD]),
    "gh_hybrid_setup": (C.c_int, [_CTX, _CTX, C.c_int, _PD]),
    # native RCCL transport (csrc/rccl_comm.hip)
    "gh_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "gh_rccl_create": (C.c_int, [C.POINTER(_CTX), C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "gh_rccl_create_all": (C.c_int, [C.POINTER(_CTX), C.c_int, _PI]),
    "gh_rccl_ops": (C.c_void_p, [_CTX]),
    "gh_rccl_last_error": (C.c_char_p, [_CTX]),
    "gh_rccl_load_error": (C.c_char_p, []),
    "gh_rccl_counters": (C.c_int, [_CTX, _PL, _PL, _PL, C.c_int]),
    "gh_rccl_destroy": (None, [_CTX]),
}

_lib = None


def load_library(path=LIB_PATH):
    """Load libgandalf_hip.so and bind every declared symbol.  Raises if it is missing: there is no
    CPU fallback for the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise ImportError("%s not built (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(_PD) if a is not None else None


_ENUMS = {
    "boundary": {"open": 0, "periodic": 1, "mirror": 2},
    "kernel": {"m4": 0, "quintic": 1},
    "gas_eos": {"energy_eqn": 0, "isothermal": 1, "barotropic": 2},
    "avisc": {"none": 0, "mon97": 1},
    "acond": {"none": 0, "wadsley2008": 1, "price2008": 2},
    "multipole": {"monopole": 0, "quadrupole": 1, "fast_monopole": 2, "fast_quadrupole": 3},
    "gravity_mac": {"geometric": 0, "gadget2": 1, "eigenmac": 2},
}


def config_from_params(p, device=0):
    """Build a gh_config from a dict that uses the reference's parameter-file keys."""
    c = Config()
    nd = int(p.get("ndim", 3))
    c.ndim = nd
    c.kernel = _ENUMS["kernel"][p.get("kernel", "m4")]
    if int(p.get("tabulated_kernel", 1)) != 0:          # the reference's default is 1 (Parameters.cpp:254)
        c.kernel = 3 if p.get("kernel", "m4") == "quintic" else 2     # GH_KERNEL_QUINTIC_TAB / GH_KERNEL_M4_TAB
    c.gas_eos = _ENUMS["gas_eos"][p.get("gas_eos", "energy_eqn")]
    c.avisc = _ENUMS["avisc"][p.get("avisc", "mon97")]
    tdav = p.get("time_dependent_avisc", "none")
    if tdav == "mm97" and c.avisc == 1:
        c.avisc = 2                       # GH_AVISC_MON97MM97
    elif tdav == "cd2010" and c.avisc == 1:
        c.avisc = 3                       # GH_AVISC_MON97CD2010
    elif tdav != "none":
        raise ValueError("time_dependent_avisc = %s is not built (none | mm97 | cd2010 with avisc = mon97)" % tdav)
    c.alpha_visc_min = float(p.get("alpha_visc_min", 0.1))
    c.acond = _ENUMS["acond"][p.get("acond", "none")]
    c.self_gravity = int(p.get("self_gravity", 0))
    c.hydro_forces = int(p.get("hydro_forces", 1))
    c.multipole = _ENUMS["multipole"][p.get("multipole", "quadrupole")]
    c.gravity_mac = _ENUMS["gravity_mac"][p.get("gravity_mac", "geometric")]
    c.macerror = float(p.get("macerror", 0.0001))
    c.Nleafmax = int(p.get("Nleafmax", 6))
    c.Nlevels = int(p.get("Nlevels", 1))
    c.level_diff_max = int(p.get("level_diff_max", 1))
    c.ntreebuildstep = int(p.get("ntreebuildstep", 1))
    c.ntreestockstep = int(p.get("ntreestockstep", 1))
    c.sph_single_timestep = int(p.get("sph_single_timestep", 0))
    c.energy_integration = 1 if p.get("gas_eos", "energy_eqn") == "energy_eqn" else 0
    c.device = device
    for k in range(3):
        c.boundary_lhs[k] = _ENUMS["boundary"][p.get("boundary_lhs[%d]" % k, "open")]
        c.boundary_rhs[k] = _ENUMS["boundary"][p.get("boundary_rhs[%d]" % k, "open")]
        c.boxmin[k] = float(p.get("boxmin[%d]" % k, 0.0))
        c.boxmax[k] = float(p.get("boxmax[%d]" % k, 0.0))
    c.h_fac = float(p.get("h_fac", 1.2))
    c.h_converge = float(p.get("h_converge", 0.01))
    c.alpha_visc = float(p.get("alpha_visc", 1.0))
    c.beta_visc = float(p.get("beta_visc", 2.0))
    c.gamma_eos = float(p.get("gamma_eos", 1.66666666666666))
    c.temp0 = float(p.get("temp0", 1.0))
    c.mu_bar = float(p.get("mu_bar", 1.0))
    c.rho_bary = float(p.get("rho_bary", 1.0e-14))
    c.thetamaxsqd = float(p.get("thetamaxsqd", 0.1))
    c.courant_mult = float(p.get("courant_mult", 0.15))
    c.accel_mult = float(p.get("accel_mult", 0.3))
    c.energy_mult = float(p.get("energy_mult", 0.4))
    # sink particles (dimensionless = 1: rho_sink and sink_radius are in code units already, SphSimulation.cpp:128-136)
    c.sink_particles = int(p.get("sink_particles", 0))
    c.create_sinks = int(p.get("create_sinks", 0)) if c.sink_particles else 0
    c.smooth_accretion = int(p.get("smooth_accretion", 0))
    c.sink_radius_mode = {"fixed": 0, "hmult": 1}.get(p.get("sink_radius_mode", "hmult"), 2)
    c.Nsinkfixed = int(p.get("Nsinkfixed", -1))
    c.rho_sink = float(p.get("rho_sink", 1.0e-12))
    c.sink_radius = float(p.get("sink_radius", 2.0))
    c.alpha_ss = float(p.get("alpha_ss", 0.01))
    c.smooth_accrete_frac = float(p.get("smooth_accrete_frac", 0.01))
    c.smooth_accrete_dt = float(p.get("smooth_accrete_dt", 0.01))
    return c


class GandalfHip:
    """One device context (= one GPU).  Methods map one-to-one onto the C ABI."""

    def __init__(self, params, device=0):
        self.lib = load_library()
        self.cfg = config_from_params(params, device) if not isinstance(params, Config) else params
        self.ndim = self.cfg.ndim
        self.ctx = _CTX()
        rc = self.lib.gh_create(C.byref(self.cfg), C.byref(self.ctx))
        if rc:
            msg = self.lib.gh_last_error(self.ctx).decode() if self.ctx else "gh_create failed"
            if self.ctx:
                self.lib.gh_destroy(self.ctx)
                self.ctx = None
            raise GhError(rc, msg)

    @classmethod
    def borrow(cls, ctx_ptr, ndim, self_gravity=0):
        """wrap a gh_ctx* owned by someone else (the C++ host shell)"""
        self = cls.__new__(cls)
        self.lib = load_library()
        self.cfg = Config()
        self.cfg.ndim = ndim
        self.cfg.self_gravity = self_gravity
        self.ndim = ndim
        self.ctx = _CTX(ctx_ptr)
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "_borrowed", False):
            self.ctx = None
            return
        if getattr(self, "ctx", None):
            self.lib.gh_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise GhError(rc, self.lib.gh_last_error(self.ctx).decode())

    @property
    def N(self):
        return int(self.lib.gh_num_particles(self.ctx))

    def upload(self, r, m, h, v=None, u=None):
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1, self.ndim)
        n = r.shape[0]
        m = np.ascontiguousarray(m, dtype=np.float64)
        h = np.ascontiguousarray(h, dtype=np.float64)
        v = None if v is None else np.ascontiguousarray(v, dtype=np.float64).reshape(n, self.ndim)
        u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        assert m.shape == (n,) and h.shape == (n,)
        self._chk(self.lib.gh_upload_particles(self.ctx, n, _dp(r), _dp(v), _dp(m), _dp(h), _dp(u)))

    def upload_field(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self._chk(self.lib.gh_upload_field(self.ctx, FIELDS[name], _dp(arr)))

    def download(self, name):
        n = self.N
        shape = (n, self.ndim) if name in VECTOR_FIELDS else (n,)
        out = np.full(shape, np.nan, dtype=np.float64)      # multi-GPU: only this rank's own particles are written
        self._chk(self.lib.gh_download(self.ctx, FIELDS[name], _dp(out)))
        return out

    def sinks(self):
        """SinkParticle records of a sink run: dict of arrays (radius, mmax, ..., angmom[3], invh, istar, Ngas) + mmean"""
        n = C.c_int32()
        self._chk(self.lib.gh_get_sinks(self.ctx, C.byref(n), None, None))
        n = n.value
        d, i = np.zeros((max(n, 1), 17)), np.zeros((max(n, 1), 2), dtype=np.int32)
        self._chk(self.lib.gh_get_sinks(self.ctx, C.byref(C.c_int32()), _dp(d), i.ctypes.data_as(_PI)))
        names = ["radius", "mmax", "menc", "dmdt", "ketot", "gpetot", "rotketot", "utot", "taccrete", "trad", "trot", "tvisc"]
        out = {k: d[:n, j].copy() for j, k in enumerate(names)}
        out["angmom"] = d[:n, 12:15].copy(); out["invh"] = d[:n, 15].copy(); out["mmean"] = float(d[0, 16]) if n else 0.0
        out["istar"] = i[:n, 0].copy(); out["Ngas"] = i[:n, 1].copy()
        return out

    def build_tree(self):
        self._chk(self.lib.gh_build_tree(self.ctx))

    def tree_size(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self.lib.gh_tree_size(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def export_tree(self):
        ncell, ltot, gtot = self.tree_size()
        nd, n = self.ndim, self.N
        t = {"Ncell": ncell, "ltot": ltot, "gtot": gtot}
        for k in ("level", "first", "N"):
            t[k] = np.zeros(ncell, dtype=np.int32)
        for k in ("bbmin", "bbmax", "hboxmin", "hboxmax", "rcell", "com"):
            t[k] = np.zeros((ncell, nd))
        for k in ("m", "rmax", "hmax", "cdistsqd"):
            t[k] = np.zeros(ncell)
        t["order"] = np.zeros(n, dtype=np.int32)
        ip = lambda a: a.ctypes.data_as(_PI)  # noqa: E731
        self._chk(self.lib.gh_export_tree(self.ctx, ip(t["level"]), ip(t["first"]), ip(t["N"]), _dp(t["bbmin"]),
                                          _dp(t["bbmax"]), _dp(t["hboxmin"]), _dp(t["hboxmax"]), _dp(t["rcell"]),
                                          _dp(t["com"]), _dp(t["m"]), _dp(t["rmax"]), _dp(t["hmax"]),
                                          _dp(t["cdistsqd"]), ip(t["order"])))
        return t

    def update_density(self, stats=False):
        st = Stats()
        self._chk(self.lib.gh_update_density(self.ctx, C.byref(st) if stats else None))
        return st.as_dict() if stats else None

    def zero_accelerations(self):
        self._chk(self.lib.gh_zero_accelerations(self.ctx))

    def update_hydro_forces(self, stats=False):
        st = Stats()
        self._chk(self.lib.gh_update_hydro_forces(self.ctx, C.byref(st) if stats else None))
        return st.as_dict() if stats else None

    def update_all_forces(self, stats=False):
        st = Stats()
        self._chk(self.lib.gh_update_all_forces(self.ctx, C.byref(st) if stats else None))
        return st.as_dict() if stats else None

    def update_forces(self, stats=False):
        return self.update_all_forces(stats) if self.cfg.self_gravity else self.update_hydro_forces(stats)

    def kdk_advance(self, n, t, timestep):
        self._chk(self.lib.gh_kdk_advance(self.ctx, n, t, timestep))

    def compute_global_timestep(self):
        dt = C.c_double()
        self._chk(self.lib.gh_compute_global_timestep(self.ctx, C.byref(dt)))
        return dt.value

    def kdk_end(self, n, t, timestep):
        self._chk(self.lib.gh_kdk_end(self.ctx, n, t, timestep))

    def setup(self, initial_h_provided=True):
        dt = C.c_double()
        self._chk(self.lib.gh_setup(self.ctx, 1 if initial_h_provided else 0, C.byref(dt)))
        return dt.value

    def set_block_clock(self, n, nresync, level_max, level_step, dt_max):
        """integer clock of a block-timestep run (Simulation: n, nresync, level_max, level_step) and dt_max"""
        self._chk(self.lib.gh_set_block_clock(self.ctx, n, nresync, level_max, level_step, dt_max))

    def set_stars(self, r, m, h, nbody_softening=1):
        """stars of a hybrid gas + N-body run (their gravity acts on the gas; see gh_set_stars)"""
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1, self.ndim)
        m = np.ascontiguousarray(m, dtype=np.float64); h = np.ascontiguousarray(h, dtype=np.float64)
        self.nstars = r.shape[0]
        self._chk(self.lib.gh_set_stars(self.ctx, self.nstars, _dp(r), _dp(m), _dp(h), int(nbody_softening)))

    def star_gas_forces(self):
        """gravity of the gas on every star through the gas tree: (a [nstars][ndim], gpot [nstars])"""
        a = np.zeros((self.nstars, self.ndim)); g = np.zeros(self.nstars)
        self._chk(self.lib.gh_star_gas_forces(self.ctx, _dp(a), _dp(g)))
        return a, g

    def active_count(self, reset=False):
        """particle force evaluations since the last reset (block-timestep runs)"""
        v = np.zeros(1, dtype=np.int64)
        self._chk(self.lib.gh_get_active_count(self.ctx, v.ctypes.data_as(_PL), 1 if reset else 0))
        return int(v[0])

    def get_block_clock(self):
        v = np.zeros(4, dtype=np.int32)
        dt_max = np.zeros(1)
        self._chk(self.lib.gh_get_block_clock(self.ctx, v.ctypes.data_as(_PI), _dp(dt_max)))
        return [int(x) for x in v], float(dt_max[0])

    def set_time(self, t, timestep):
        self._chk(self.lib.gh_set_time(self.ctx, t, timestep))

    def step(self, nsteps=1):
        t, dt = C.c_double(), C.c_double()
        self._chk(self.lib.gh_step(self.ctx, nsteps, C.byref(t), C.byref(dt)))
        return t.value, dt.value

    def gather_neighbours(self):
        n = self.N
        offs = np.zeros(n + 1, dtype=np.int64)
        cap = 128 * n
        ids = np.zeros(cap, dtype=np.int32)
        rc = self.lib.gh_gather_neighbours(self.ctx, cap, offs.ctypes.data_as(_PL), ids.ctypes.data_as(_PI))
        if rc == 1:
            cap = int(offs[n])
            ids = np.zeros(cap, dtype=np.int32)
            rc = self.lib.gh_gather_neighbours(self.ctx, cap, offs.ctypes.data_as(_PL), ids.ctypes.data_as(_PI))
        self._chk(rc)
        return offs, ids[:offs[n]]

    def gather_neighbours_at(self, rp, rsearch, cap=4096):
        """point query (Tree.cpp:208-280): ids of the particles within rsearch of rp; None if more than cap"""
        rp = np.ascontiguousarray(rp, dtype=np.float64)
        out = np.zeros(cap, dtype=np.int32)
        n = self.lib.gh_gather_neighbours_at(self.ctx, _dp(rp), float(rsearch), out.ctypes.data_as(_PI), cap)
        if n == -1:
            return None
        if n < 0:
            raise GhError(n, self.lib.gh_last_error(self.ctx).decode())
        return out[:n]

    def timers(self):
        ms = (C.c_double * len(GH_T_NAMES))()
        d, f = Stats(), Stats()
        self._chk(self.lib.gh_get_timers(self.ctx, ms, C.byref(d), C.byref(f)))
        return dict(zip(GH_T_NAMES, list(ms))), d.as_dict(), f.as_dict()

    def reset_timers(self):
        self._chk(self.lib.gh_reset_timers(self.ctx))

    def comm_init(self, rank, nranks, ops_ptr):
        """gh_comm_init: ops_ptr = address of a gh_comm_ops (multigpu.CommOps), or None for a single rank"""
        self._chk(self.lib.gh_comm_init(self.ctx, rank, nranks, ops_ptr))

    def exchange_halo(self, phase):
        self._chk(self.lib.gh_exchange_halo(self.ctx, phase))

    def allgather_multipoles(self):
        self._chk(self.lib.gh_allgather_multipoles(self.ctx))

    def comm_info(self):
        """(own_first, own_count, held): this rank's range of the global tree order; particles held = own + imported"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.lib.gh_comm_info(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def stream_handle(self):
        """hipStream_t of the context as an integer (for torch.cuda.ExternalStream)"""
        return int(self.lib.gh_stream(self.ctx) or 0)

    def update_hmax(self):
        self._chk(self.lib.gh_update_hmax(self.ctx))

    def field_dev(self, name, k=0):
        return self.lib.gh_field_dev(self.ctx, FIELDS[name], k)


class NbodyHip:
    """Stars: direct-sum forces + leapfrog KDK on the GPU (gh_nbody_* of include/gandalf_hip.h)."""
    FIELDS = {"r": 0, "v": 1, "a": 2, "adot": 3, "gpot": 4, "r0": 5, "v0": 6, "a0": 7, "tlast": 8}

    def __init__(self, ndim=3, softening=0, nbody_mult=0.1, device=0):
        self.lib = load_library()
        self.ndim = ndim
        self.N = 0
        ctx = _CTX()
        rc = self.lib.gh_nbody_create(ndim, int(softening), float(nbody_mult), device, C.byref(ctx))
        self.ctx = ctx
        self._chk(rc)

    def _chk(self, rc):
        if rc < 0:
            msg = self.lib.gh_nbody_last_error(self.ctx) if self.ctx else b"gh_nbody_create failed"
            raise GhError(rc, (msg or b"").decode())
        return rc

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.gh_nbody_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, r, v, m, h):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (r, v, m, h)]
        self.N = len(a[2])
        self._chk(self.lib.gh_nbody_upload(self.ctx, self.N, *[_dp(x) for x in a]))

    def num_stars(self):
        """sink runs create stars"""
        self.N = int(self.lib.gh_nbody_num_stars(self.ctx))
        return self.N

    def download(self, name):
        self.num_stars()
        if name in ("m", "h", "dt_internal"):
            out = np.empty(self.N)
            self._chk(self.lib.gh_nbody_download_scalar(self.ctx, {"m": 0, "h": 1, "dt_internal": 2}[name], _dp(out)))
            return out
        out = np.empty(self.N if name in ("gpot", "tlast") else (self.N, self.ndim))
        if self.N:
            self._chk(self.lib.gh_nbody_download(self.ctx, self.FIELDS[name], _dp(out)))
        return out

    def upload_field(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self._chk(self.lib.gh_nbody_upload_field(self.ctx, self.FIELDS[name], _dp(arr)))

    def hybrid_setup(self, gas, initial_h_provided=False):
        """PostInitialConditionsSetup of a hybrid run; returns the first timestep"""
        dt = C.c_double()
        rc = self.lib.gh_hybrid_setup(gas.ctx, self.ctx, 1 if initial_h_provided else 0, C.byref(dt))
        if rc:
            raise GhError(rc, gas.lib.gh_last_error(gas.ctx).decode() or self.lib.gh_nbody_last_error(self.ctx).decode())
        return dt.value

    def hybrid_step(self, gas, nsteps=1):
        """nsteps MainLoop calls of a hybrid gas + stars run; `gas` is the GandalfHip context holding the gas"""
        t, dt = C.c_double(), C.c_double()
        rc = self.lib.gh_hybrid_step(gas.ctx, self.ctx, int(nsteps), C.byref(t), C.byref(dt))
        if rc:
            msg = gas.lib.gh_last_error(gas.ctx).decode() or self.lib.gh_nbody_last_error(self.ctx).decode()
            raise GhError(rc, msg)
        return t.value, dt.value

    def forces(self):
        self._chk(self.lib.gh_nbody_forces(self.ctx))

    def setup(self):
        dt = C.c_double()
        self._chk(self.lib.gh_nbody_setup(self.ctx, C.byref(dt)))
        return dt.value

    def step(self, nsteps=1):
        t, dt = C.c_double(), C.c_double()
        self._chk(self.lib.gh_nbody_step(self.ctx, int(nsteps), C.byref(t), C.byref(dt)))
        return t.value, dt.value
