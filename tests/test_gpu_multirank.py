"""Sharded (multi-rank) stepping must give bit-identical results to single-rank stepping: two ranks
share GPU 0 and exchange their slices over gloo (RCCL refuses two ranks on one device; the exchange
code path is otherwise the same as with backend nccl)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from gandalf_amd.host import Simulation
from gandalf_amd.multigpu import ShardedRunner
case, out = sys.argv[2], sys.argv[3]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
sim = Simulation(os.path.join(sys.argv[1], "tests", "params", case + ".dat"))
sim.generate_ic()
run = ShardedRunner(sim, rank, world)
run.setup()
run.steps(2)
dev = sim.device()
if rank == 0:
    np.savez(out, **{k: dev.download(k) for k in ("r", "v", "h", "rho", "a", "u", "dudt")})
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("case", ["plummer_4k", "box3d_4k"])
def test_two_ranks_equal_one_rank(case, tmp_path):
    wf = tmp_path/"worker.py"
    wf.write_text(WORKER)
    outs = {}
    for world in (1, 2):
        out = str(tmp_path/("w%d.npz" % world))
        port = _port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, str(wf), ROOT, case, out], env=env))
        for p in procs:
            assert p.wait(timeout=600) == 0
        outs[world] = dict(np.load(out))
    for k in outs[1]:
        assert np.array_equal(outs[1][k], outs[2][k]), k
