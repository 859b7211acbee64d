"""Multi-GPU stepping: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Round-1 scheme: every rank keeps the full particle set and builds the same tree
(288 GB of HBM per GPU makes replication free at these sizes); the expensive phases - density and
forces - are sharded by contiguous slices of tree groups (= top-level tree cells), and their outputs are
all-gathered.  With world == 1 this is plain gh_step."""
import numpy as np


class ShardedRunner:
    def __init__(self, sim, rank, world):
        self.sim, self.rank, self.world = sim, rank, world

    def setup(self):
        if self.world == 1:
            self.sim.post_ic_setup()
            return
        raise NotImplementedError("multi-GPU stepping lands with the sharded step API")

    def steps(self, n):
        if n <= 0:
            return
        if self.world == 1:
            self.sim.main_loop(n)
            return
        raise NotImplementedError

    def count_density(self):
        """counters of one density pass on the current state (instrumented kernel build)"""
        dev = self.sim.device()
        h = dev.download("h")
        st = dev.update_density(stats=True)
        dev.upload_field("h", h)       # leave the state as it was
        return st

    def count_forces(self):
        dev = self.sim.device()
        saved = {k: dev.download(k) for k in ("a", "atree", "gpot", "gpot_hydro", "dudt", "div_v")}
        dev.zero_accelerations()
        st = dev.update_forces(stats=True)
        for k, v in saved.items():
            dev.upload_field(k, v)
        return st
