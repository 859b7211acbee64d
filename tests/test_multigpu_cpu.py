"""N > 1 path on CPU: the slice exchange of gandalf_amd.multigpu (pack -> all_gather -> unpack) with
world_size 2 and 3 over gloo, against a stand-in device that keeps its arrays in host memory."""
import ctypes
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gandalf_amd.multigpu import Exchanger


class HostDevice:
    """same interface as GandalfHip's shard methods, arrays in numpy (tree order)"""

    def __init__(self, n, world, narr, rank, seed=0):
        self.n, self.world, self.narr = n, world, narr
        # uneven slices, like slices of tree groups
        cuts = np.linspace(0, n, world + 1).astype(np.int64)
        cuts[1:-1] += np.arange(1, world)*3
        self.cuts = cuts
        rng = np.random.default_rng(seed)
        self.truth = rng.standard_normal((narr, n))
        self.arr = np.zeros((narr, n))
        a, b = cuts[rank], cuts[rank + 1]
        self.arr[:, a:b] = self.truth[:, a:b]       # each rank only knows its own slice

    def shard_range(self, r):
        return int(self.cuts[r]), int(self.cuts[r + 1] - self.cuts[r])

    def exchange_narrays(self, xset):
        return self.narr

    def _view(self, ptr, count):
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(count,))

    def shard_pack(self, xset, dst_ptr, stride):
        a, c = self.shard_range(self.rank)
        buf = self._view(dst_ptr, self.narr*stride)
        for k in range(self.narr):
            buf[k*stride:k*stride + c] = self.arr[k, a:a + c]

    def shard_unpack(self, xset, r, src_ptr, stride):
        a, c = self.shard_range(r)
        buf = self._view(src_ptr, self.narr*stride)
        for k in range(self.narr):
            self.arr[k, a:a + c] = buf[k*stride:k*stride + c]

    def shard_unpack_all(self, xset, src_ptr, stride):
        """src: [rank][array][stride], as all_gather_into_tensor lays it out"""
        for r in range(self.world):
            if r != self.rank:
                self.shard_unpack(xset, r, src_ptr + r*self.narr*stride*8, stride)


def _worker(rank, world, port, n, narr, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = HostDevice(n, world, narr, rank)
    dev.rank = rank
    x = Exchanger(dev, rank, world, torch.device("cpu"))
    x.exchange(0)
    x.exchange(0)            # buffers are reused
    q.put((rank, bool(np.array_equal(dev.arr, dev.truth))))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, n, narr):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, narr, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_exchange_world2():
    _run(2, 1000, 10)


def test_exchange_world3_uneven():
    _run(3, 517, 4)
