#!/usr/bin/env python3
"""W ranks of the domain-decomposed path inside ONE process, one thread per rank, all contexts on GPU 0.

A one-GPU box allows at most 6 processes on the card, so the 8-rank case of the scaling bench cannot be rehearsed with
one process per rank.  The library only asks its host for two collectives over device buffers (gh_comm_ops); here they
are device-to-device copies between the ranks' buffers behind a thread barrier.  Functional check only (results of W
ranks against the one-rank run), never a measurement.

    python scripts/probe/threaded_ranks.py 8 65536 [nsteps] [params]
"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gandalf_amd.host import Simulation                                   # noqa: E402
from gandalf_amd.multigpu import _ALLGATHER_T, _ALLTOALLV_T, _OpsStruct, _DevPtr   # noqa: E402


class Shared:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        # THR_SERIAL=1: only the rank that holds the token enqueues GPU work between two collectives, so that a kernel trace
        # of the run shows every kernel with the duration it has on a GPU of its own (profiles/r03_multirank_*)
        self.token = threading.Lock() if os.environ.get("THR_SERIAL") else None
        self.ncoll = 0
        self.send = [0]*world
        self.sbytes = [None]*world
        self.errors = []


def dev_view(ptr, nbytes):
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device="cuda:0")
    return torch.as_tensor(_DevPtr(ptr, nbytes), device="cuda:0")


class ThreadOps:
    def __init__(self, shared, rank):
        self.sh, self.rank = shared, rank
        self._ag = _ALLGATHER_T(self._allgather)
        self._a2a = _ALLTOALLV_T(self._alltoallv)
        self.struct = _OpsStruct(None, self._ag, self._a2a)
        self.ptr = C.addressof(self.struct)

    def _allgather(self, user, send, recv, nbytes, stream):
        try:
            sh, W = self.sh, self.sh.world
            torch.cuda.synchronize()
            sh.send[self.rank] = int(send)
            if sh.token:
                sh.token.release()
            if self.rank == 0:
                sh.ncoll += 1
            sh.barrier.wait()
            out = dev_view(recv, nbytes*W)
            for r in range(W):
                out[r*nbytes:(r + 1)*nbytes].copy_(dev_view(sh.send[r], nbytes))
            torch.cuda.synchronize()
            sh.barrier.wait()
            if sh.token:
                sh.token.acquire()
            return 0
        except Exception as e:          # noqa: BLE001
            self.sh.errors.append(e)
            return 1

    def _alltoallv(self, user, send, send_bytes, recv, recv_bytes, stream):
        try:
            sh, W = self.sh, self.sh.world
            torch.cuda.synchronize()
            sh.send[self.rank] = int(send)
            sh.sbytes[self.rank] = [int(send_bytes[r]) for r in range(W)]
            rb = [int(recv_bytes[r]) for r in range(W)]
            if sh.token:
                sh.token.release()
            if self.rank == 0:
                sh.ncoll += 1
            sh.barrier.wait()
            out = dev_view(recv, sum(rb))
            ro = 0
            for r in range(W):
                sb = sh.sbytes[r]
                assert sb[self.rank] == rb[r], ("block size mismatch", r, self.rank, sb[self.rank], rb[r])
                if rb[r] > 0:
                    so = sum(sb[:self.rank])
                    out[ro:ro + rb[r]].copy_(dev_view(sh.send[r] + so, rb[r]))
                ro += rb[r]
            torch.cuda.synchronize()
            sh.barrier.wait()
            if sh.token:
                sh.token.acquire()
            return 0
        except Exception as e:          # noqa: BLE001
            self.sh.errors.append(e)
            try:
                self.sh.barrier.abort()
            except Exception:           # noqa: BLE001
                pass
            return 1


def run(world, N, nsteps, params):
    over = {"Nhydro": N, "run_id": "THR%d" % world}
    sims = []
    for r in range(world):
        sim = Simulation(os.path.join(ROOT, "tests", "params", params + ".dat"), **over)
        sim.generate_ic()
        sims.append(sim)
    shared = Shared(world)
    ops = [ThreadOps(shared, r) for r in range(world)]
    out = [None]*world
    wall = [0.0]*world
    fail = []

    def work(r):
        try:
            torch.cuda.set_device(0)
            sim = sims[r]
            if shared.token:
                shared.token.acquire()
            if world > 1:
                sim.init_comm(r, world, ops[r].ptr)
            sim.post_ic_setup()
            if nsteps > 0:
                torch.cuda.synchronize()
                if world > 1:
                    if shared.token:
                        shared.token.release()
                    shared.barrier.wait()
                    if r == 0:           # sentinel kernels ("flip") bracket the timed steps in a kernel trace
                        torch.arange(7, device="cuda").flip(0)
                        torch.cuda.synchronize()
                        shared.ncoll = 0
                    shared.barrier.wait()
                    if shared.token:
                        shared.token.acquire()
                t0 = time.perf_counter()
                sim.main_loop(nsteps)
                torch.cuda.synchronize()
                wall[r] = (time.perf_counter() - t0)/nsteps
                if world > 1:
                    if shared.token:
                        shared.token.release()
                    shared.barrier.wait()
                    if r == 0:
                        torch.arange(7, device="cuda").flip(0)
                        torch.cuda.synchronize()
                        print("world %d: %d collectives in %d steps = %.1f per step" % (world, shared.ncoll, nsteps, shared.ncoll/nsteps))
                    shared.barrier.wait()
                    if shared.token:
                        shared.token.acquire()
            dev = sim.device()
            sinks = sim.get_param("sink_particles") == "1"
            names = ("h", "rho", "a", "gpot", "dudt") + (("m", "sinkid", "flags") if sinks else ())
            out[r] = {k: np.nan_to_num(dev.download(k), nan=0.0) for k in names}
            out[r]["info"] = dev.comm_info()
            if sinks:
                out[r]["sinks"] = dev.sinks()
                out[r]["N"] = dev.N
            if shared.token:
                shared.token.release()
        except Exception as e:          # noqa: BLE001
            fail.append((r, e))
            try:
                shared.barrier.abort()
            except Exception:           # noqa: BLE001
                pass
            try:
                if shared.token:
                    shared.token.release()
            except Exception:           # noqa: BLE001
                pass

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if fail or shared.errors:
        raise RuntimeError("ranks failed: %r %r" % (fail, shared.errors))
    res = {k: sum(o[k] for o in out) for k in out[0] if k not in ("info", "sinks", "N")}     # every particle is owned by one rank
    if "sinks" in out[0]:
        for o in out[1:]:       # the sinks (and the particle count) are everybody's: identical on all ranks
            assert o["N"] == out[0]["N"] and all(np.array_equal(o["sinks"][k], out[0]["sinks"][k]) for k in out[0]["sinks"])
        res["sinks"], res["N"] = out[0]["sinks"], out[0]["N"]
    # all ranks share ONE GPU: the wall time of a step is (roughly) the SUM of the ranks' work plus the harness's device-wide
    # synchronisations - an upper bound of the mean work per rank when divided by the rank count, not a multi-GPU timing
    print("world %d: %.2f ms per step on one shared GPU = %.2f ms per rank" % (world, 1e3*max(wall), 1e3*max(wall)/world))
    return res, [o["info"] for o in out]


def relerr(a, b):
    mag = np.abs(a) if a.ndim == 1 else np.linalg.norm(a, axis=1)
    scale = np.maximum(mag, mag.mean())
    diff = np.abs(a - b) if a.ndim == 1 else np.linalg.norm(a - b, axis=1)
    return float(np.max(diff/scale)) if np.any(scale > 0) else float(np.max(np.abs(b)))


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    params = sys.argv[4] if len(sys.argv) > 4 else "plummer_4k"
    one, _ = run(1, N, nsteps, params)
    many, info = run(world, N, nsteps, params)
    sinks = "sinks" in one
    errs = {k: relerr(one[k], many[k]) for k in one if k not in ("sinks", "N", "sinkid", "flags")}
    own = [int(i[1]) for i in info]
    held = [int(i[2]) for i in info]
    print("world %d N %d steps %d %s: max rel err vs one rank %s" % (world, N, nsteps, params, {k: "%.2e" % v for k, v in errs.items()}))
    print("own", own, "held", held)
    if sinks:
        # sink run: which particles became sinks or were accreted, sinkid and dead flag of everybody, Ngas - exact; sums to rounding
        # (the stars' forces are sums of the ranks' partial sums)
        a, b = one["sinks"], many["sinks"]
        print("sinks: N %d -> %d (one rank: %d), %d sinks, Ngas %s" % (N, many["N"], one["N"], len(b["istar"]), list(b["Ngas"])))
        assert one["N"] == many["N"] == sum(own) and len(a["istar"]) == len(b["istar"])
        assert np.array_equal(a["Ngas"], b["Ngas"]) and np.array_equal(one["sinkid"], many["sinkid"])
        assert np.array_equal(one["flags"].astype(np.int64) & 4, many["flags"].astype(np.int64) & 4)
        assert np.array_equal(one["m"] == 0.0, many["m"] == 0.0)
        for k in ("mmax", "menc", "dmdt", "utot"):
            assert len(a[k]) == 0 or np.max(np.abs(a[k] - b[k])) <= 1e-10*np.max(np.abs(a[k])), k
    assert all(v <= (1e-11 if sinks else 1e-13) for v in errs.values()), errs
    print("OK")


if __name__ == "__main__":
    main()
