#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the SPH + tree-gravity hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload plummer1m|box256k] [--no-cpu]

A step is one full pass of the hot path (SphSimulation::MainLoop of the reference: KDK predict, tree
rebuild, density + h, hydro + gravity forces, timestep, KDK correct) over all particles, with the
particles resident in HBM.  Default workload: BASELINE.json configs[2], the 1 048 576-particle Plummer
gas sphere with KD-tree self-gravity (theta = 0.5, monopole) - the configuration the metric is quoted on.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_PEAK_TFLOPS = 78.6    # fp64 vector: half the guide's 157.3 TFLOP/s fp32 vector rate (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
F64_PEAK_TINSTR = 39.3    # ... as issue slots: 256 CU x 4 SIMD x 16 fp64 lanes/clk x 2.4 GHz = 3.93e13 lane-instructions/s


def _sha16(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


WORKLOADS = {
    "plummer1m": dict(params="plummer_4k.dat", overrides={"Nhydro": 1048576, "run_id": "PLUM1M"},
                      desc="3-D Plummer gas sphere, N=1048576, self-gravity KD-tree theta=0.5 monopole (BASELINE configs[2])"),
    "box256k": dict(params="box3d_4k.dat", overrides={"Nhydro": 262144, "run_id": "BOX256K"},
                    desc="3-D uniform-random periodic box, N=262144, hydro only (BASELINE configs[1])"),
    "plummer1m_fastmono": dict(params="plummer_4k.dat", overrides={"Nhydro": 1048576, "run_id": "PLUM1MFM", "multipole": "fast_monopole"},
                               desc="3-D Plummer gas sphere, N=1048576, multipole = fast_monopole (not the metric's config)"),
    "plummer1m_quad": dict(params="plummer_4k.dat", overrides={"Nhydro": 1048576, "run_id": "PLUM1MQ", "multipole": "quadrupole"},
                           desc="3-D Plummer gas sphere, N=1048576, multipole = quadrupole, the reference's default (not the metric's config)"),
    "plummer1m_tb8": dict(params="plummer_4k.dat", overrides={"Nhydro": 1048576, "run_id": "PLUM1MTB8", "ntreebuildstep": 8},
                          desc="3-D Plummer gas sphere, N=1048576, ntreebuildstep = 8: tree re-stocked on 7 of 8 steps (not the metric's config)"),
    "plummer64k": dict(params="plummer_4k.dat", overrides={"Nhydro": 65536, "run_id": "PLUM64K"},
                       desc="3-D Plummer gas sphere, N=65536 (reduced; not the metric's config)"),
}


def cpu_baseline(workload, sim=None):
    """The compiled reference (oracle/_ref/ref_dump, OpenMP) timed on this box's host cores on a bounded sample of the
    same workload.  Reported beside the GPU number, never as part of it.

    Same-size sample: the state the GPU run has reached is written as a SEREN `su` snapshot
    (the reference's own format; SnapshotIO.cpp) and the reference starts from it with `ic = file` - it then carries
    converged smoothing lengths (SimulationIO.hpp:794 sets initial_h_provided), which avoids the reference's ~500 s
    first density pass from one global h guess at 1M Plummer particles."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(exe):
        return None
    cores = min(os.cpu_count() or 1, int(os.environ.get("GH_CPU_THREADS", "16")))
    wl = WORKLOADS[workload]
    n = int(wl["overrides"].get("Nhydro", 4096))
    if workload.startswith("plummer"):
        steps, warm = (2, 1) if n > 300000 else (8, 1)
        shape = "Plummer gas sphere self-gravity theta=0.5 monopole"
    else:
        steps, warm = 10, 1
        shape = "uniform-random periodic box, hydro only"
    with tempfile.TemporaryDirectory() as tmp:
        src = open(os.path.join(ROOT, "tests", "params", wl["params"])).read()
        start = "its own IC generator"
        if sim is not None and n > 300000:
            snap = os.path.join(tmp, "state.su")
            sim.write_snapshot(snap, "su")
            src = "\n".join(l for l in src.splitlines() if not l.startswith("ic ")) + "\nic = file\nin_file = %s\nin_file_form = su\n" % snap
            start = "the GPU run's state as an su snapshot (h provided)"
        else:
            src = src.replace("Nhydro = 4096", "Nhydro = %d" % n)
        for k, v in wl["overrides"].items():
            if k not in ("Nhydro", "run_id"):
                src = "\n".join(l for l in src.splitlines() if not l.startswith(k + " ")) + "\n%s = %s\n" % (k, v)
        pf = os.path.join(tmp, "p.dat")
        open(pf, "w").write(src)
        env = dict(os.environ, OMP_NUM_THREADS=str(cores))
        if "ic = file" in src:
            env["REF_H_PROVIDED"] = "1"       # the su reader reads h but does not say so (only the sf reader does, SimulationIO.hpp:794)
        t0 = time.time()
        out = subprocess.run([exe, "time", pf, str(steps), str(warm)], cwd=tmp, env=env, capture_output=True, text=True)
        wall = time.time() - t0
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not line:
        sys.stderr.write("cpu_baseline: ref_dump failed (%d): %s\n" % (out.returncode, out.stderr[-500:]))
        return None
    r = json.loads(line[-1])
    return {"value": r["particle_steps_per_s"], "unit": "particle-steps/s", "cores": r["threads"], "kind": "reference",
            "sample": "%s, N=%d, %d timed steps after %d warm-up, started from %s (GANDALF reference built -O3 OpenMP by "
                      "oracle/ref.mk, %d threads; whole run incl. setup %.0f s)" % (shape, r["N"], steps, warm, start, r["threads"], wall)}


def spawn_ranks(n, argv):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <argv>` as a child process
    (this process never initialises the GPU; nothing is exec'ed)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="plummer1m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched as plain `python bench.py --gpus N`: start N ranks (one per GPU) as FRESH child processes
        # before this process has touched the GPU, relay rank 0's JSON line and exit with the children's code
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GH_BENCH_BACKEND=gloo: functional rehearsal of the N-rank path on a box with ONE GPU (all ranks share device 0, the
    # collectives go through host memory) - never a measurement
    backend = os.environ.get("GH_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)

    from gandalf_amd.host import Simulation
    from gandalf_amd import multigpu

    wl = WORKLOADS[args.workload]
    sim = Simulation(os.path.join(ROOT, "tests", "params", wl["params"]), **wl["overrides"])
    sim.set_param("device", local_rank)
    ic = sim.generate_ic()
    N = ic["r"].shape[0]
    # N > 1: the library's own RCCL binding carries every collective of the stepped loop (csrc/rccl_comm.hip);
    # GH_COMM=torch routes them through torch.distributed callbacks instead (the gloo rehearsal always does)
    transport = os.environ.get("GH_COMM", "rccl" if (backend == "nccl" and world > 1) else "torch")
    runner = multigpu.DistributedRunner(sim, rank, world, transport=transport, device=local_rank)       # world == 1: plain single-GPU stepping
    runner.setup()

    def comm_calls():
        if runner.ops is None:
            return {"allgather": 0, "alltoallv": 0, "bytes": 0}
        return runner.ops.counters() if runner.transport == "rccl" else dict(runner.ops.calls)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    runner.steps(args.warmup)
    dev = sim.device()
    dev.reset_timers()
    sync()
    c0 = comm_calls()
    t0 = time.perf_counter()
    runner.steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    c1 = comm_calls()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    timers, dst, fst = dev.timers()
    # per-rank device time of the walk + evaluation phases: max / mean over the ranks is the work imbalance of the
    # count-balanced decomposition (the reference balances on measured work, MpiKDTreeDecomposition.cpp:282-360)
    imbalance = None
    if world > 1:
        mine = {k: float(v)/args.steps for k, v in timers.items()}
        allt = [None]*world
        dist.all_gather_object(allt, mine)
        work = [t.get("SPH_PROPERTIES", 0.0) + t.get("GRAV_WALK", 0.0) + t.get("SPH_FORCES", 0.0) for t in allt]
        imbalance = {"work_ms_per_rank": work, "max_over_mean": max(work)/(sum(work)/world) if sum(work) > 0 else None,
                     "phase_ms_per_rank": allt}

    # ---- roofline accounting.  Interaction counts come from the kernels' own counters (one
    #      instrumented pass on the final state), durations from the HIP events the library records on its stream
    #      around every launch of the timed steps.
    dens = runner.count_density()
    forc = runner.count_forces()
    nown = N/world                                           # particles this rank computes for
    grav = int(sim.get_param("self_gravity"))
    dens_ms = dst["kernel_ms"]
    forc_ms = fst["kernel_ms"]

    # measured HBM bytes / fp64 VALU instructions per launch from separate rocprofv3 --pmc passes (profiles/pmc_traffic.json);
    # ignored when older than the library it was measured on, or for another workload / rank count
    pmc = {}
    try:
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        so = os.path.join(ROOT, "gandalf_amd", "csrc", "libgandalf_hip.so")
        if world == 1:
            with open(pj) as f:
                allp = json.load(f)
            if allp.get("_library_sha16") == _sha16(so):
                pmc = allp.get(args.workload, {})
    except (OSError, ValueError):
        pass

    # (1) density pass: SURVEY 8(d)'s edge model - every (particle, candidate inside the reference's cull) moves 32 B per
    #     h iteration, every particle 112 B in + out - against the HBM peak.  The north_star target (>= 0.6) is on this number.
    dens_bytes = dens["n_candidates"]*32.0 + nown*112.0
    gbs = dens_bytes/(dens_ms*1e-3)/1e9 if dens_ms > 0 else 0.0
    r_d = {"kernel": "density pass (k_dens_walk + k_dens_eval + k_density fallback)", "bound": "hbm", "achieved": gbs,
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs/HBM_PEAK_GBS, "traffic": pmc.get("density_pass_hbm_bytes"),
           "algorithmic_bytes_per_launch": dens_bytes, "avg_launch_ms": dens_ms,
           "model": "N_it*N_cand*32 B + 112 B per particle (SURVEY 8d)"}
    if grav:
        # (2) gravity evaluation.  It applies every loaded list entry to the 4-6 particles of a leaf from registers, so
        #     SURVEY 8(d)'s per-(particle, entry) byte count is not a bound for it (it came out at 2.1x the HBM peak in
        #     round 1).  Its two real ceilings:
        #     a. fp64 VALU issue: lane-instructions the interactions need (counted in the ISA of the kernel: 19 per
        #        point-mass term, 20 per classified hydro candidate, +115 per SPH pair) against
        #        256 CU x 4 SIMD x 16 fp64 lanes/clk x 2.4 GHz = 3.93e13 lane-instr/s (= 78.6 TFLOP/s counting an FMA
        #        slot as 2 flop, half the guide's 157.3 TFLOP/s fp32 vector peak);
        #     b. HBM: bytes the kernel actually requests, counted per (leaf, entry) as it loads them.
        instr = 19.0*(forc["n_cells"] + forc["n_direct"]) + 20.0*forc["n_candidates"] + 115.0*forc["n_candidates"]
        tin = instr/(forc_ms*1e-3)/1e12 if forc_ms > 0 else 0.0          # tera lane-instructions per second
        leaf_bytes = (forc["n_leaf_cells"]*36.0 + forc["n_leaf_direct"]*32.0 + forc["n_leaf_cand"]*40.0
                      + forc["n_candidates"]*128.0 + nown*(136.0 + 80.0))
        hbm_gbs = leaf_bytes/(forc_ms*1e-3)/1e9 if forc_ms > 0 else 0.0
        meas = pmc.get("k_grav_eval_f64_flops")
        r_valu = {"kernel": "k_grav_eval", "bound": "fp64_valu", "achieved": tin, "peak": F64_PEAK_TINSTR, "unit": "T lane-instr/s",
                  "frac": tin/F64_PEAK_TINSTR, "traffic": pmc.get("k_grav_eval"),
                  "measured_fp64_tflops": (meas/(forc_ms*1e-3)/1e12 if (meas and forc_ms > 0) else None), "fp64_peak_tflops": F64_PEAK_TFLOPS,
                  "algorithmic_fp64_lane_instructions_per_launch": instr, "avg_launch_ms": forc_ms,
                  "measured_fp64_lane_instructions_per_launch": pmc.get("k_grav_eval_f64_lane_instr"),
                  "model": "19*(cell + direct terms) + 135*SPH pairs fp64 VALU lane-instructions against the issue peak 3.93e13 lane-instr/s (one slot per instruction whatever it is; measured_fp64_tflops counts an FMA as 2 flop, the others as 1, from the PMC pass)",
                  "hbm_leaf_model": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs/HBM_PEAK_GBS,
                                     "bytes_per_launch": leaf_bytes,
                                     "model": "36 B per (leaf, cell entry) + 32 B per (leaf, direct particle) + 40 B per (leaf, hydro candidate) + 128 B per SPH pair + 216 B per particle"},
                  "walk_ms": timers.get("GRAV_WALK", 0.0)/args.steps}
        # the requested bytes are served by L2 / MALL to ~95 % (16.8 MB of cell records, measured HBM traffic = `traffic`), so
        # they do not bound the kernel by the HBM peak; the VALU issue roof is the binding one
        r_valu["hbm_leaf_model"]["note"] = "bytes requested from the memory system (L2 + MALL + HBM), not HBM traffic"
        r_f = r_valu
    else:
        f_bytes = forc["n_candidates"]*112.0 + nown*(112.0 + 56.0)
        gbs = f_bytes/(forc_ms*1e-3)/1e9 if forc_ms > 0 else 0.0
        r_f = {"kernel": "k_hydro_forces", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": gbs/HBM_PEAK_GBS, "traffic": pmc.get("k_hydro_forces"), "algorithmic_bytes_per_launch": f_bytes,
               "avg_launch_ms": forc_ms, "model": "N_pair*112 B + 168 B per particle (SURVEY 8d)"}
    dominant = r_f if forc_ms >= dens_ms else r_d

    if rank == 0:
        value = N*args.steps/elapsed
        out = {
            "metric": "particle-steps/s (density+force+gravity)" if int(sim.get_param("self_gravity")) else "particle-steps/s (density+force)",
            "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3*elapsed/args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["desc"], "N": N, "ic": "reference xorshift RNG seed 1 (bit-exact IC generators)",
                       "parallelism": ("domain-decomposed x%d: rank r owns top-level KD cell r, halo + multipole exchange over RCCL" % world) if world > 1 else "single GPU"},
            "roofline": dominant,
            "roofline_density": r_d,
            "roofline_forces": r_f,
            "phase_ms_per_step": {k: v/args.steps for k, v in timers.items()},
            "counters": {"density": dens, "forces": forc},
        }
        if world > 1:
            out["multi_gpu"] = {"transport": "RCCL bound natively in libgandalf_hip.so (ncclAllGather, grouped ncclSend/ncclRecv)" if runner.transport == "rccl"
                                else "torch.distributed callbacks (%s)" % backend,
                                "collectives_per_step": {k: (c1[k] - c0[k])/args.steps for k in ("allgather", "alltoallv")},
                                "bytes_per_step": (c1["bytes"] - c0["bytes"])/args.steps, "imbalance": imbalance}
        if world > 1 and backend != "nccl":
            out["note"] = "REHEARSAL: %d ranks sharing one GPU over gloo - not a measurement" % world
        cb = None if (args.no_cpu or world > 1) else cpu_baseline(args.workload, sim)
        out["cpu_baseline"] = cb
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
