"""N > 1 path on CPU: the two collectives libgandalf_hip asks its host for (include/gandalf_hip.h: gh_comm_ops) as
gandalf_amd.multigpu.CommOps implements them on torch.distributed - world sizes 2 and 4 over gloo, called through the
same C function pointers the library calls, on host buffers (memory="host"; the GPU test in test_gpu_multirank.py
drives the same class on device buffers)."""
import ctypes as C
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from gandalf_amd.multigpu import CommOps


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops = CommOps("host")
    st = ops.struct
    ok = True
    # ---- allgather: blocks of 6 doubles (the root-box exchange) and of 0 bytes
    mine = np.arange(6, dtype=np.float64) + 100.0*rank
    allb = np.zeros(6*world)
    rc = st.allgather(None, mine.ctypes.data, allb.ctypes.data, mine.nbytes, None)
    ok &= rc == 0 and np.array_equal(allb, np.concatenate([np.arange(6) + 100.0*r for r in range(world)]))
    rc = st.allgather(None, mine.ctypes.data, allb.ctypes.data, 0, None)
    ok &= rc == 0
    # ---- alltoallv: ragged blocks, some empty (rank r sends (r + 2*t) % 5 records of 43 doubles to rank t: the migration)
    rec = 43
    nsend = [(rank + 2*t) % 5 if t != rank else 0 for t in range(world)]
    nrecv = [(s + 2*rank) % 5 if s != rank else 0 for s in range(world)]
    send = np.concatenate([np.full(n*rec, 1000.0*rank + t) for t, n in enumerate(nsend)] + [np.zeros(0)])
    recv = np.full(sum(nrecv)*rec + 1, -1.0)
    sb = (C.c_int64*world)(*[n*rec*8 for n in nsend]); rb = (C.c_int64*world)(*[n*rec*8 for n in nrecv])
    rc = st.alltoallv(None, send.ctypes.data if send.size else None, sb, recv.ctypes.data, rb, None)
    want = np.concatenate([np.full(n*rec, 1000.0*s + rank) for s, n in enumerate(nrecv)] + [np.full(1, -1.0)])
    ok &= rc == 0 and np.array_equal(recv, want)
    # ---- a collective that fails inside Python reports through the return code, not an exception through C
    bad = (C.c_int64*world)(*[8]*world)
    rc = st.alltoallv(None, None, bad, None, bad, None)
    ok &= rc != 0 and ops.last_error is not None
    q.put((rank, bool(ok), ops.calls))
    dist.barrier()
    dist.destroy_process_group()


def _run(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in res) == list(range(world))
    assert all(ok for _, ok, _ in res), res
    assert all(c["allgather"] == 2 and c["alltoallv"] == 2 for _, _, c in res)


def test_comm_ops_world2():
    _run(2)


def test_comm_ops_world4():
    _run(4)
