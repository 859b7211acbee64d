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


def cpu_baseline(workload, verbose=False):
    """The compiled reference (oracle/_ref/ref_dump, OpenMP) timed on this box's host cores on a bounded
    sample of the same workload shape.  Reported beside the GPU number, never as part of it."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(exe):
        return None
    cores = min(os.cpu_count() or 1, int(os.environ.get("GH_CPU_THREADS", "16")))
    # bounded samples of the same workload shape: ~10-30 s of CPU work in all (the 1M Plummer setup alone
    # takes the reference ~500 s on 8 cores because of its first density pass)
    if workload.startswith("plummer"):
        par, n, steps, warm = "plummer_4k.dat", 131072, 8, 1
        shape = "Plummer gas sphere self-gravity theta=0.5 monopole"
    else:
        par, n, steps, warm = "box3d_4k.dat", 262144, 10, 1
        shape = "uniform-random periodic box, hydro only"
    with tempfile.TemporaryDirectory() as tmp:
        src = open(os.path.join(ROOT, "tests", "params", par)).read().replace("Nhydro = 4096", "Nhydro = %d" % n)
        pf = os.path.join(tmp, "p.dat")
        open(pf, "w").write(src)
        env = dict(os.environ, OMP_NUM_THREADS=str(cores))
        t0 = time.time()
        out = subprocess.run([exe, "time", pf, str(steps), str(warm)], cwd=tmp, env=env, capture_output=True, text=True)
        wall = time.time() - t0
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not line:
        return None
    r = json.loads(line[-1])
    return {"value": r["particle_steps_per_s"], "unit": "particle-steps/s", "cores": r["threads"], "kind": "reference",
            "sample": "%s, N=%d, %d timed steps after %d warm-up (GANDALF reference built -O3 OpenMP by oracle/ref.mk; "
                      "whole run incl. setup %.0f s)" % (shape, n, steps, warm, wall)}


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
    runner = multigpu.DistributedRunner(sim, rank, world)       # world == 1: plain single-GPU stepping
    runner.setup()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    runner.steps(args.warmup)
    dev = sim.device()
    dev.reset_timers()
    sync()
    t0 = time.perf_counter()
    runner.steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    timers, dst, fst = dev.timers()

    # ---- roofline accounting: algorithmic bytes of SURVEY.md 8(d) from the kernels' own counters
    dens = runner.count_density()
    forc = runner.count_forces()
    ngpu_share = 1.0/world
    dens_bytes = (dens["n_candidates"]*32.0 + N*ngpu_share*(48.0 + 64.0))
    f_bytes = (forc["n_candidates"]*(112.0 + 8.0) + forc["n_direct"]*32.0 + forc["n_cells"]*32.0
               + N*ngpu_share*(112.0 + 56.0 + 40.0)) if int(sim.get_param("self_gravity")) else \
              (forc["n_candidates"]*112.0 + N*ngpu_share*(112.0 + 56.0))
    dens_ms = dst["kernel_ms"]
    forc_ms = fst["kernel_ms"]

    # HBM bytes per launch measured with PMC counters in a separate rocprofv3 pass (profiles/pmc_traffic.json);
    # only valid for the single-GPU workload it was collected on
    pmc = {}
    try:
        if world == 1:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f).get(args.workload, {})
    except OSError:
        pass

    def roof(name, nbytes, ms):
        gbs = nbytes/(ms*1e-3)/1e9 if ms > 0 else 0.0
        return {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs/HBM_PEAK_GBS, "traffic": pmc.get(name.split(" ")[0]),
                "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms}

    r_d = roof("k_density", dens_bytes, dens_ms)
    grav = int(sim.get_param("self_gravity"))
    r_f = roof("k_grav_eval (interaction lists from k_grav_walk)" if grav else "k_hydro_forces", f_bytes, forc_ms)
    if grav:
        r_f["walk_ms"] = timers.get("GRAV_WALK", 0.0)/args.steps
        r_f["note"] = ("algorithmic bytes are counted per (particle, list entry) interaction (SURVEY 8d); the kernel "
                       "evaluates every loaded entry against the 4-6 particles of a leaf, so the model rate may exceed "
                       "the HBM peak - real HBM traffic is in profiles/")
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
        if world > 1 and backend != "nccl":
            out["note"] = "REHEARSAL: %d ranks sharing one GPU over gloo - not a measurement" % world
        cb = None if (args.no_cpu or world > 1) else cpu_baseline(args.workload)
        out["cpu_baseline"] = cb
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
