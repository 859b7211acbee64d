"""Multi-GPU runs: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI) as the transport.

libgandalf_hip does the domain decomposition, halo selection and packing itself (gandalf_amd/csrc/comm.hip,
DESIGN.md section 7) and asks its host for exactly two collectives over device buffers (include/gandalf_hip.h:
gh_comm_ops): an all-gather of equal blocks and an all-to-all of ragged blocks.  This module supplies them:

  CommOps            the gh_comm_ops table as ctypes callbacks around torch.distributed
                       nccl : all_gather_into_tensor / all_to_all_single on the library's own HIP stream
                              (torch.cuda.ExternalStream), raw device pointers wrapped through __cuda_array_interface__
                       gloo : the same calls staged through host memory - used by the CPU tests (host pointers) and by
                              the functional test that runs two ranks on one GPU
  DistributedRunner  bench.py / the tests' driver: registers the ops with the C++ host shell and runs setup / steps

The reference's counterpart is its MPI layer (src/Mpi/MpiControl.cpp: MPI_Allgather :329-337, MPI_Alltoallv :1073-1150)."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

_ALLGATHER_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
_ALLTOALLV_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_void_p)


class _OpsStruct(C.Structure):          # gh_comm_ops
    _fields_ = [("user", C.c_void_p), ("allgather", _ALLGATHER_T), ("alltoallv", _ALLTOALLV_T)]


class _DevPtr:
    """raw device memory as a uint8 array (torch.as_tensor accepts the CUDA array interface on ROCm too)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _host_view(ptr, nbytes):
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8)
    return torch.from_numpy(np.ctypeslib.as_array((C.c_ubyte*int(nbytes)).from_address(int(ptr))))


class CommOps:
    """gh_comm_ops on torch.distributed.  memory = "device" (pointers are HIP device memory) or "host" (CPU tests)."""

    def __init__(self, memory="device", device=None, group=None):
        self.memory = memory
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if memory == "device" else torch.device("cpu"))
        self._streams = {}
        self.calls = {"allgather": 0, "alltoallv": 0, "bytes": 0}
        self.last_error = None
        self._ag = _ALLGATHER_T(self._allgather)          # keep the callback objects alive
        self._a2a = _ALLTOALLV_T(self._alltoallv)
        self.struct = _OpsStruct(None, self._ag, self._a2a)

    @property
    def ptr(self):
        return C.addressof(self.struct)

    # ---- views of the library's buffers
    def _view(self, ptr, nbytes):
        if self.memory == "host":
            return _host_view(ptr, nbytes)
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.device)
        return torch.as_tensor(_DevPtr(ptr, nbytes), device=self.device)

    def _stream(self, handle):
        if handle not in self._streams:
            self._streams[handle] = torch.cuda.ExternalStream(int(handle), device=self.device)
        return self._streams[handle]

    # ---- gh_comm_ops.allgather
    def _allgather(self, user, send, recv, nbytes, stream):
        try:
            self.calls["allgather"] += 1
            self.calls["bytes"] += int(nbytes)*self.world
            inp, out = self._view(send, nbytes), self._view(recv, nbytes*self.world)
            if self.memory == "device" and self.backend == "nccl":
                with torch.cuda.stream(self._stream(stream)):          # RCCL orders its work against the current stream
                    dist.all_gather_into_tensor(out, inp, group=self.group)
                return 0
            if self.memory == "device":                                  # gloo moves host memory
                self._stream(stream).synchronize()
                hi, ho = inp.cpu(), torch.empty(out.shape, dtype=torch.uint8)
                dist.all_gather_into_tensor(ho, hi, group=self.group)
                out.copy_(ho)
                torch.cuda.synchronize(self.device)
                return 0
            dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group)
            return 0
        except Exception as e:      # noqa: BLE001 - a Python exception must not unwind through the C caller
            self.last_error = e
            return 1

    # ---- gh_comm_ops.alltoallv
    def _alltoallv(self, user, send, send_bytes, recv, recv_bytes, stream):
        try:
            self.calls["alltoallv"] += 1
            sb = [int(send_bytes[r]) for r in range(self.world)]
            rb = [int(recv_bytes[r]) for r in range(self.world)]
            self.calls["bytes"] += sum(sb)
            inp, out = self._view(send, sum(sb)), self._view(recv, sum(rb))
            if self.memory == "device" and self.backend == "nccl":
                with torch.cuda.stream(self._stream(stream)):
                    dist.all_to_all_single(out, inp, rb, sb, group=self.group)
                return 0
            if self.memory == "device":
                self._stream(stream).synchronize()
                hi, ho = inp.cpu(), torch.empty(out.shape, dtype=torch.uint8)
                self._host_alltoallv(ho, hi, rb, sb)
                out.copy_(ho)
                torch.cuda.synchronize(self.device)
                return 0
            self._host_alltoallv(out, inp, rb, sb)
            return 0
        except Exception as e:      # noqa: BLE001
            self.last_error = e
            return 1

    def _host_alltoallv(self, out, inp, rb, sb):
        """ragged all-to-all on host tensors with point-to-point messages (gloo has no all_to_all on every build)"""
        so = np.concatenate(([0], np.cumsum(sb))).astype(np.int64)
        ro = np.concatenate(([0], np.cumsum(rb))).astype(np.int64)
        out[ro[self.rank]:ro[self.rank + 1]] = inp[so[self.rank]:so[self.rank + 1]]
        reqs, keep = [], []
        for r in range(self.world):
            if r == self.rank:
                continue
            if rb[r] > 0:
                buf = torch.empty(rb[r], dtype=torch.uint8)
                keep.append((r, buf))
                reqs.append(dist.irecv(buf, src=r, group=self.group))
            if sb[r] > 0:
                reqs.append(dist.isend(inp[so[r]:so[r + 1]].contiguous(), dst=r, group=self.group))
        for q in reqs:
            q.wait()
        for r, buf in keep:
            out[ro[r]:ro[r + 1]] = buf


class RcclOps:
    """gh_comm_ops bound to RCCL inside libgandalf_hip.so (csrc/rccl_comm.hip): ncclAllGather and grouped
    ncclSend / ncclRecv on the library's own stream - no collective of the stepped loop comes back into Python.
    torch.distributed is only the launcher's rendezvous here: rank 0's ncclUniqueId (128 bytes) reaches the other ranks
    through `dist.broadcast_object_list` (any backend), like MPI_Bcast in a C++ host."""

    def __init__(self, rank, world, device):
        from . import capi
        self.lib = capi.load_library()
        err = self.lib.gh_rccl_load_error()
        if err:
            raise RuntimeError(err.decode())
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            if self.lib.gh_rccl_unique_id(buf) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
            ident = [buf.raw]
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        self.rank, self.world = rank, world
        self.handle = C.c_void_p()
        rc = self.lib.gh_rccl_create(C.byref(self.handle), rank, world, C.create_string_buffer(ident[0], 128), int(device))
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed (rank %d of %d)" % (rank, world))
        self.ptr = self.lib.gh_rccl_ops(self.handle)
        self.last_error = None

    def counters(self, reset=False):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self.lib.gh_rccl_counters(self.handle, C.byref(a), C.byref(b), C.byref(c), int(reset))
        return {"allgather": a.value, "alltoallv": b.value, "bytes": c.value}

    def error(self):
        return self.lib.gh_rccl_last_error(self.handle).decode()

    def close(self):
        if self.handle:
            self.lib.gh_rccl_destroy(self.handle)
            self.handle = C.c_void_p()


def _selftest_ops(ops_ptr, rank, world, device):
    """one all-gather and one ragged all-to-all through a gh_comm_ops table on known data; "" if both came out right"""
    ops = C.cast(ops_ptr, C.POINTER(_OpsStruct)).contents
    dev = torch.device("cuda", device)
    stream = torch.cuda.current_stream(dev)
    send = torch.full((4,), rank + 1, dtype=torch.int64, device=dev)
    recv = torch.zeros(4*world, dtype=torch.int64, device=dev)
    stream.synchronize()
    if ops.allgather(ops.user, send.data_ptr(), recv.data_ptr(), 32, stream.cuda_stream) != 0:
        return "all-gather returned an error"
    stream.synchronize()
    if not torch.equal(recv.cpu(), torch.arange(1, world + 1, dtype=torch.int64).repeat_interleave(4)):
        return "all-gather delivered wrong data"
    # rank r sends (d + 1) words of value 100 r + d to rank d
    sb = [(d + 1)*8 for d in range(world)]
    rb = [(rank + 1)*8]*world
    out = torch.cat([torch.full((d + 1,), 100*rank + d, dtype=torch.int64) for d in range(world)]).to(dev)
    inn = torch.zeros((rank + 1)*world, dtype=torch.int64, device=dev)
    stream.synchronize()
    if ops.alltoallv(ops.user, out.data_ptr(), (C.c_int64*world)(*sb), inn.data_ptr(), (C.c_int64*world)(*rb), stream.cuda_stream) != 0:
        return "all-to-all returned an error"
    stream.synchronize()
    want = torch.cat([torch.full((rank + 1,), 100*s + rank, dtype=torch.int64) for s in range(world)])
    if not torch.equal(inn.cpu(), want):
        return "all-to-all delivered wrong data"
    return ""


class DistributedRunner:
    """setup() / steps(n) of a gandalf_amd.host.Simulation on `world` ranks (rank r owns top-level KD cell r).
    world == 1 is the plain single-GPU run."""

    def __init__(self, sim, rank=0, world=1, transport="torch", device=0):
        """transport: "rccl" = the library's own RCCL binding (RcclOps; what bench.py uses on real multi-GPU nodes),
        "torch" = torch.distributed callbacks (CommOps; gloo for several ranks on one GPU and for the CPU tests)"""
        self.sim, self.rank, self.world = sim, rank, world
        self.transport, self.device = transport, device
        self.ops = None
        self.dev = None

    def setup(self):
        if self.transport == "rccl":
            # RCCL cannot be loaded at all (no librccl on this host): every rank sees the same and takes the torch transport
            # instead - decided before the first collective, so no rank waits for another
            from . import capi
            err = capi.load_library().gh_rccl_load_error()
            if err:
                import sys
                sys.stderr.write("gandalf_amd: native RCCL transport unavailable (%s): using torch.distributed\n" % err.decode())
                self.transport = "torch"
        if self.transport == "rccl" and self.world > 1:
            # bring the native transport up and try both collectives once on known data; the verdict is reduced over the
            # ranks (torch's process group), so either every rank steps on RCCL inside the library or every rank takes the
            # torch.distributed callbacks - never a mixture, never a failure in the middle of the stepped loop
            ok, why = 1, ""
            try:
                self.ops = RcclOps(self.rank, self.world, self.device)
                why = _selftest_ops(self.ops.ptr, self.rank, self.world, self.device)
                ok = 0 if why else 1
            except Exception as e:      # noqa: BLE001
                ok, why = 0, repr(e)
            flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", self.device) if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                import sys
                if why:
                    sys.stderr.write("gandalf_amd: rank %d: native RCCL transport failed its self-test (%s)\n" % (self.rank, why))
                if self.rank == 0:
                    sys.stderr.write("gandalf_amd: using torch.distributed for the collectives\n")
                self.transport, self.ops = "torch", None
        if self.ops is None and (self.world > 1 or self.transport == "rccl"):
            self.ops = RcclOps(self.rank, self.world, self.device) if self.transport == "rccl" else CommOps("device")
        if self.world > 1:
            self.sim.init_comm(self.rank, self.world, self.ops.ptr)
        self.sim.post_ic_setup()
        self.dev = self.sim.device()
        self._check()

    def steps(self, n):
        if n > 0:
            self.sim.main_loop(n)
            self._check()

    def _check(self):
        if self.ops is not None and self.ops.last_error is not None:
            raise self.ops.last_error

    def gather(self, name):
        """a field of all particles in caller order, merged over the ranks (every rank gets the whole array)"""
        a = self.dev.download(name)
        if self.world == 1:
            return a
        t = torch.from_numpy(np.nan_to_num(a, nan=0.0))
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t)               # every particle is owned by exactly one rank: the others contribute zeros
        return t.cpu().numpy()

    def count_density(self):
        """counters of one density pass over this rank's particles, on the current state (collective)"""
        dev = self.dev
        h = dev.download("h")
        st = dev.update_density(stats=True)
        dev.upload_field("h", np.nan_to_num(h))       # leave h as it was; the next step rebuilds everything else
        return st

    def count_forces(self):
        dev = self.dev
        saved = {k: np.nan_to_num(dev.download(k)) for k in ("a", "atree", "gpot", "gpot_hydro", "dudt", "div_v")}
        dev.zero_accelerations()
        st = dev.update_forces(stats=True)
        for k, v in saved.items():
            dev.upload_field(k, v)
        return st
