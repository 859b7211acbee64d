"""Domain-decomposed multi-rank runs (gandalf_amd/csrc/comm.hip) against the single-rank run of the same problem.
Two / four ranks share GPU 0 and use gloo as the transport (RCCL refuses two ranks on one device; the library-side
code - decomposition, migration, halo selection, pack / unpack - is the same whatever carries the bytes).

Every rank owns one top-level cell of the SAME global KD-tree (exact distributed median splits) and imports the halo
its walks need, so the per-particle results must equal the single-rank ones to rounding - and do so bit for bit."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from gandalf_amd.host import Simulation
from gandalf_amd.multigpu import DistributedRunner
case, out, nsteps = sys.argv[2], sys.argv[3], int(sys.argv[4])
over = json.loads(sys.argv[5])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
sim = Simulation(os.path.join(sys.argv[1], "tests", "params", case + ".dat"), **over)
sim.generate_ic()
run = DistributedRunner(sim, rank, world, transport=os.environ.get("GH_TEST_TRANSPORT", "torch"))
run.setup()
if os.environ.get("GH_TEST_TRANSPORT"):
    print("TRANSPORT rank %d: %s" % (rank, run.transport), flush=True)
run.steps(nsteps)
fields = ["r", "v", "h", "rho", "a", "u", "dudt", "gpot"] + (["level", "levelneib", "nstep", "nlast"] if int(sim.get_param("Nlevels")) > 1 else [])
sinks = int(sim.get_param("sink_particles")) == 1
if sinks:
    fields += ["m", "sinkid", "flags"]
res = {k: run.gather(k) for k in fields}
if sinks:
    sk = run.dev.sinks()
    for k, v in sk.items():
        res["sink_" + k] = np.atleast_1d(np.asarray(v))
    res["Nhydro"] = np.array([run.dev.N])
own_first, own_count, held = run.dev.comm_info()
info = np.array([own_count, held, sim.t, sim.timestep])
if world > 1:
    allinfo = [None]*world
    dist.all_gather_object(allinfo, info.tolist())
    info = np.array(allinfo)
if rank == 0:
    np.savez(out, info=info, **res)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(tmp_path, case, world, nsteps, over, extra_env=None, timeout=900):
    wf = tmp_path/"worker.py"
    wf.write_text(WORKER)
    out = str(tmp_path/("w%d.npz" % world))
    port = _port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(wf), ROOT, case, out, str(nsteps), json.dumps(over)], env=env))
    try:
        for p in procs:
            assert p.wait(timeout=timeout) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return dict(np.load(out))


def _relerr(a, b):
    """max |a - b| relative to max(|a_i|, mean |a|); 0 for two all-zero arrays (gpot of a hydro-only run)"""
    mag = np.abs(a) if a.ndim == 1 else np.linalg.norm(a, axis=1)
    scale = np.maximum(mag, mag.mean())
    if not np.any(scale > 0):
        return float(np.max(np.abs(b)))
    diff = np.abs(a - b) if a.ndim == 1 else np.linalg.norm(a - b, axis=1)
    return float(np.max(diff/scale))


CASES = {
    "gravity": ("plummer_4k", {}),                                         # hydro + self-gravity (tree walk with MAC)
    "hydro": ("plummer_4k", {"self_gravity": 0, "run_id": "PLUMHYD"}),      # hydro only (scatter-gather walk)
    "periodic": ("box3d_4k", {}),                                          # periodic box: the halo selection tests the images too
    "periodic32k": ("box3d_4k", {"Nhydro": 32768, "run_id": "BOXP32K"}),    # ... with a halo that is a layer, not the whole box
}


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("case", sorted(CASES))
def test_ranks_equal_one_rank(case, world, tmp_path):
    par, over = CASES[case]
    one = _run(tmp_path, par, 1, 3, over)
    many = _run(tmp_path, par, world, 3, over)
    assert many["info"].shape == (world, 4)
    assert np.all(many["info"][:, 2] == one["info"][2]) and np.all(many["info"][:, 3] == one["info"][3])      # t, dt
    for k in ("r", "v", "h", "rho", "a", "u", "dudt", "gpot"):
        a, b = one[k], many[k]
        assert np.all(np.isfinite(b)), k
        assert _relerr(a, b) <= 1e-13, (k, _relerr(a, b))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("case", ["plummer_4k_levels", "box3d_4k_levels"])
def test_block_timesteps_on_ranks_equal_one_rank(case, world, tmp_path):
    """Hierarchical block timesteps (Nlevels = 5) under the domain decomposition: level / nstep / nlast / flags migrate with
    the particles, the minimum timestep, the highest occupied level and CheckTimesteps' wake-up count are reduced over the
    ranks (Simulation.cpp:1843-1847, 2016-2080; SphSimulation.cpp:753), the halo carries the neighbours' levels and the
    levelneib raised on halo copies returns to the owners.  40 steps: the level structure and the integer clock must be
    EXACTLY those of the one-rank run, the fields equal to rounding."""
    one = _run(tmp_path, case, 1, 40, {})
    many = _run(tmp_path, case, world, 40, {})
    assert np.all(many["info"][:, 2] == one["info"][2]) and np.all(many["info"][:, 3] == one["info"][3])      # t, dt
    for k in ("level", "levelneib", "nstep", "nlast"):
        assert np.array_equal(one[k], many[k]), k
    for k in ("r", "v", "h", "rho", "a", "u", "dudt", "gpot"):
        assert np.all(np.isfinite(many[k])), k
        assert _relerr(one[k], many[k]) <= 1e-12, (k, _relerr(one[k], many[k]))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("case,nsteps", [("bb_sinks_8k", 12), ("bb_sinks_8k_levels", 40)])
def test_sinks_on_ranks_equal_one_rank(case, nsteps, world, tmp_path):
    """Sink particles under the domain decomposition (BASELINE configs[4]'s physics: Boss-Bodenheimer cloud, sink creation,
    smooth accretion; `_levels`: on the block-timestep ladder).  The sinks' stars live on every rank; the candidate search is a
    gather of the ranks' best (density, slot), the chosen particle's row and the gas inside the sink radii are gathered to all
    ranks, which run the serial accretion alike; accreted particles leave the arrays by the reference's slot rule applied to the
    gathered dead slots, every cell's particle count follows the new N and the next migration evens the ranks out; the star
    forces are partial sums over the ranks' subtrees.  Two sinks form and accrete in these steps: which particles they were,
    Ngas, sinkid and dead flag of every particle, the compacted order, levels and clock are EXACTLY the one-rank run's; sums to
    rounding.  (One difference is inherent: inside a leaf cell the one-rank sink run keeps the reference's quick-select order,
    several ranks the coordinate order - the potential-minimum test, which the reference makes order dependent, sees its
    neighbours in that order.)"""
    one = _run(tmp_path, case, 1, nsteps, {})
    many = _run(tmp_path, case, world, nsteps, {})
    assert one["Nhydro"][0] == many["Nhydro"][0] < 8192 and len(one["sink_istar"]) == len(many["sink_istar"]) == 2
    # (the stars' accelerations are sums of the ranks' partial sums: equal to rounding, and so are the timesteps they set)
    assert np.all(np.abs(many["info"][:, 2] - one["info"][2]) <= 1e-12*one["info"][2])                       # t
    assert np.all(np.abs(many["info"][:, 3] - one["info"][3]) <= 1e-10*one["info"][3])                       # dt
    assert many["info"][:, 0].sum() == many["Nhydro"][0]                                                     # the ranks' cells hold the new N
    for k in ("sinkid", "flags") + (("level", "levelneib", "nstep", "nlast") if case.endswith("levels") else ()):
        a, b = one[k].astype(np.int64), many[k].astype(np.int64)
        if k == "flags":
            a, b = a & 4, b & 4                                                                               # dead (potmin: where the search reads it)
        assert np.array_equal(a, b), k
    assert np.array_equal(one["m"] == 0.0, many["m"] == 0.0)
    for k in ("r", "v", "h", "rho", "a", "u", "gpot", "m"):
        assert np.all(np.isfinite(many[k])), k
        assert _relerr(one[k], many[k]) <= 1e-11, (k, _relerr(one[k], many[k]))
    assert np.array_equal(one["sink_istar"], many["sink_istar"]) and np.array_equal(one["sink_Ngas"], many["sink_Ngas"])
    for k in ("radius", "mmax", "menc", "dmdt", "ketot", "gpetot", "rotketot", "utot", "taccrete", "trad", "trot", "tvisc", "angmom", "mmean"):
        a, b = one["sink_" + k], many["sink_" + k]
        assert np.max(np.abs(a - b)) <= 1e-10*max(np.max(np.abs(a)), 1e-300), k


def test_two_ranks_at_the_benchmark_size(tmp_path):
    """configs[2] itself (1 048 576-particle Plummer sphere) on 2 ranks: 4 096 leaves below every published cell of a rank's
    subtree - the deepest marking walk of the halo selection any of the 2 / 4 / 8-rank runs of the scaling bench needs
    (a fixed 2 048-flag array refused exactly this case) - equal to the one-rank run after setup + 1 step."""
    over = {"Nhydro": 1048576, "run_id": "PLUM1M2R"}
    one = _run(tmp_path, "plummer_4k", 1, 1, over)
    two = _run(tmp_path, "plummer_4k", 2, 1, over)
    assert np.all(two["info"][:, 0] == 1048576//2)
    for k in ("h", "rho", "a", "gpot", "dudt"):
        assert _relerr(one[k], two[k]) <= 1e-13, (k, _relerr(one[k], two[k]))


@pytest.mark.parametrize("world,n,par", [(8, 65536, "plummer_4k"), (8, 1048576, "plummer_4k"), (8, 65536, "box3d_4k"), (8, 65536, "bb_sinks_8k_levels")])
def test_eight_ranks_in_one_process(world, n, par):
    """The 8-rank case of the scaling bench (L = 3 shared levels) cannot run as 8 processes on a one-GPU box (process
    guard); scripts/probe/threaded_ranks.py runs the ranks as threads of one process, all contexts on GPU 0, with the two
    collectives as device-to-device copies behind a barrier.  Results against the one-rank run, at 65 536 particles after
    setup + 2 steps (Plummer sphere with self-gravity; periodic box) and at the benchmark's 1 048 576 after setup + 1 step."""
    steps = "10" if par.startswith("bb_sinks") else ("2" if n < 1000000 else "1")       # sink run: two sinks form on the first step and accrete
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probe", "threaded_ranks.py"), str(world), str(n), steps, par],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "OK" in out.stdout


def test_ranks_hold_only_their_share(tmp_path):
    """65 536-particle uniform cube (open boundaries, self-gravity) on 2 and 4 ranks: every rank owns N / world particles
    and holds, with the imported halo, well under the whole set (the replicated scheme of round 1 held N on every rank).
    (Not the Plummer sphere: there the reference's own cell-overlap test, Tree.cpp:659-672, lets a handful of sparse halo
    leaves - rmax + kernrange*hmax of 6 to 7 length units - reach 85 % of the other rank's leaves at this N;
    scripts/probe/let_need.py counts that need from the reference's tests alone.)"""
    N = 65536
    over = {"Nhydro": N}
    one = _run(tmp_path, "box3d_open_grav", 1, 1, over)
    for world in (2, 4):
        many = _run(tmp_path, "box3d_open_grav", world, 1, over)
        own, held = many["info"][:, 0], many["info"][:, 1]
        assert np.all(own == N//world)
        assert np.all(held < 0.7*N), held
        assert np.all(held < own + 0.6*N/world + 0.1*N), held        # the halo is a layer, not the neighbour's whole subtree
        for k in ("rho", "a", "gpot"):
            a, b = one[k], many[k]
            assert _relerr(a, b) <= 1e-13, k


@pytest.mark.parametrize("case", ["plummer_4k_quadrupole", "plummer_4k_fastmono", "plummer_4k_fastquad", "plummer_4k_nl8", "plummer_4k_quintic", "box3d_4k_isothermal",
                                  "bb_units_1600"])
def test_other_modes_on_two_ranks(case, tmp_path):
    """Modes the domain decomposition allows besides the monopole / M4 default, each on 2 ranks against 1: quadrupole moments
    (the published subtree tops and the halo cells carry them), the fast-multipole variants, leaves of 8 particles (halo leaf
    records as wide as the leaf), the quintic kernel (kernrange 3 in every halo test), the isothermal periodic box; and
    the settings of the reference's bossbodenheimer.dat (physical units, tabulated kernel, fast monopoles, block timesteps, a
    sink run before its first sink; the one-rank run builds its first trees in the reference's leaf order, hence 1e-12)."""
    one = _run(tmp_path, case, 1, 3, {})
    two = _run(tmp_path, case, 2, 3, {})
    tol = 1e-12 if case.startswith("bb_") else 1e-13
    assert np.all(np.abs(two["info"][:, 2] - one["info"][2]) <= tol*abs(one["info"][2])) and np.all(np.abs(two["info"][:, 3] - one["info"][3]) <= 1e-10*one["info"][3])
    for k in ("r", "v", "h", "rho", "a", "u", "dudt", "gpot"):
        assert np.all(np.isfinite(two[k])), k
        assert _relerr(one[k], two[k]) <= tol, (k, _relerr(one[k], two[k]))


def test_speculative_splits_fall_back_collectively(tmp_path, monkeypatch):
    """The one-collective-per-level median search (windows around last step's medians) must hand over to the exact search
    whenever a window misses - on all ranks together.  GH_DD_WINSCALE=0 empties every window, so every step after the first
    takes: speculative attempt -> status word in the migration counts -> exact search; results as ever."""
    one = _run(tmp_path, "plummer_4k", 1, 3, {})
    monkeypatch.setenv("GH_DD_WINSCALE", "0")
    many = _run(tmp_path, "plummer_4k", 4, 3, {})
    for k in ("r", "v", "h", "rho", "a", "u", "dudt", "gpot"):
        assert _relerr(one[k], many[k]) <= 1e-13, (k, _relerr(one[k], many[k]))


def test_lattice_ties_are_refused_on_more_than_one_rank(tmp_path):
    """16^3 cubic lattice (equal coordinates at every median): one rank rebuilds the tree with the reference's own
    quick-select tie order (lattice3d_cubic_grav fixture); across ranks the shared top levels order ties by particle id,
    which would be a different tree than the reference's - every rank learns of the tie (the flag travels with the subtree
    tops) and all of them stop with the same error instead of computing on."""
    wf = tmp_path/"worker.py"
    wf.write_text(WORKER)
    port = _port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(wf), ROOT, "lattice3d_cubic_grav", str(tmp_path/"l.npz"), "1", "{}"], env=env,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode != 0
        assert "equal coordinates" in err, err[-1500:]


NCCL_WORKER = r'''
import ctypes as C, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from gandalf_amd.multigpu import CommOps
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ops = CommOps("device")
st = ops.struct
side = torch.cuda.Stream()                       # the library enqueues on a stream of its own, not torch's current one
a = torch.arange(1000, dtype=torch.float64, device="cuda")
b = torch.zeros(1000, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
rc = st.allgather(None, a.data_ptr(), b.data_ptr(), a.numel()*8, side.cuda_stream)
side.synchronize()
assert rc == 0, ops.last_error
assert torch.equal(a, b)
c = torch.zeros(1000, dtype=torch.float64, device="cuda")
n = (C.c_int64*1)(777*8)
rc = st.alltoallv(None, a.data_ptr(), n, c.data_ptr(), n, side.cuda_stream)
side.synchronize()
assert rc == 0, ops.last_error
assert torch.equal(c[:777], a[:777]) and float(c[777:].abs().sum()) == 0.0
dist.destroy_process_group()
print("NCCL_OPS_OK")
'''


def test_comm_ops_on_rccl(tmp_path):
    """the RCCL leg of CommOps (raw device pointers through the CUDA array interface, collectives enqueued on a
    foreign HIP stream) with the one rank a one-GPU box allows"""
    wf = tmp_path/"nccl_worker.py"
    wf.write_text(NCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()))
    out = subprocess.run([sys.executable, str(wf), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "NCCL_OPS_OK" in out.stdout, out.stderr[-2000:]


RCCL_WORKER = r'''
import ctypes as C, os, sys
import torch
sys.path.insert(0, sys.argv[1])
from gandalf_amd.multigpu import RcclOps, _OpsStruct
torch.cuda.set_device(0)
ops = RcclOps(0, 1, 0)                            # ncclGetUniqueId + ncclCommInitRank inside libgandalf_hip.so
st = _OpsStruct.from_address(ops.ptr)
side = torch.cuda.Stream()
a = torch.arange(1000, dtype=torch.float64, device="cuda")
b = torch.zeros(1000, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
rc = st.allgather(st.user, a.data_ptr(), b.data_ptr(), a.numel()*8, side.cuda_stream)
side.synchronize()
assert rc == 0, ops.error()
assert torch.equal(a, b)
c = torch.zeros(1000, dtype=torch.float64, device="cuda")
n = (C.c_int64*1)(777*8)
rc = st.alltoallv(st.user, a.data_ptr(), n, c.data_ptr(), n, side.cuda_stream)
side.synchronize()
assert rc == 0, ops.error()
assert torch.equal(c[:777], a[:777]) and float(c[777:].abs().sum()) == 0.0
assert ops.counters() == {"allgather": 1, "alltoallv": 1, "bytes": 8000 + 777*8}
from gandalf_amd.multigpu import _selftest_ops
assert _selftest_ops(ops.ptr, 0, 1, 0) == ""          # the check DistributedRunner makes before it trusts the transport
ops.close()
print("RCCL_NATIVE_OK")
'''


def test_native_transport_falls_back_on_all_ranks_together(tmp_path):
    """bench.py --gpus N asks for the library's own RCCL transport.  DistributedRunner brings it up, tries both collectives
    on known data and reduces the verdict over the ranks: if it cannot work on ANY rank, EVERY rank takes the torch.distributed
    callbacks instead, before the first collective of the run.  Two ranks on ONE GPU is such a case (RCCL refuses a second rank
    on a device): the run must go through on the fallback and equal the one-rank run."""
    one = _run(tmp_path, "plummer_4k", 1, 2, {})
    two = _run(tmp_path, "plummer_4k", 2, 2, {}, extra_env={"GH_TEST_TRANSPORT": "rccl"}, timeout=300)
    for k in ("h", "rho", "a", "gpot"):
        assert _relerr(one[k], two[k]) <= 1e-13, (k, _relerr(one[k], two[k]))


def test_native_rccl_ops(tmp_path):
    """csrc/rccl_comm.hip: the library's own RCCL binding (what bench.py --gpus N uses), with the one rank a one-GPU box
    allows: communicator from ncclGetUniqueId / ncclCommInitRank, ncclAllGather and the send / receive group on a foreign
    stream"""
    wf = tmp_path/"rccl_worker.py"
    wf.write_text(RCCL_WORKER)
    out = subprocess.run([sys.executable, str(wf), ROOT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_NATIVE_OK" in out.stdout, out.stdout[-1000:] + out.stderr[-2000:]
