"""Multi-GPU stepping: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Round-1 scheme: every rank keeps the full particle set and builds the same tree
(288 GB of HBM per GPU makes replication free at these sizes); the two expensive phases - density and
forces - are sharded by contiguous slices of tree groups (= top-level KD-tree cells), and the slices'
outputs are all-gathered (RCCL) after each phase.  With world == 1 this is plain gh_step.

The exchange is written against a small "device" interface (shard_range / exchange_narrays / shard_pack /
shard_unpack) so that the slicing and gather logic can be exercised on CPU tensors with the gloo
backend (tests/test_multigpu_cpu.py)."""
import torch
import torch.distributed as dist


class Exchanger:
    """all-gather of per-rank contiguous slices of several equally long arrays"""

    def __init__(self, dev, rank, world, torch_device):
        self.dev, self.rank, self.world, self.tdev = dev, rank, world, torch_device
        self.buf = {}
        self.ranges = None
        self.ext = None
        self.simulate = False                      # timing estimate only: skip the collective (see bench.py --simulate-world)

    def exchange(self, xset):
        """pack -> all-gather -> unpack, all ordered on the device context's own stream: no host synchronisation"""
        if self.world == 1:
            return
        if self.ranges is None:                     # the slices depend only on N and the rank count
            self.ranges = [self.dev.shard_range(r) for r in range(self.world)]
        stride = max(c for _, c in self.ranges)
        stride = (stride + 63)//64*64
        na = self.dev.exchange_narrays(xset)
        key = (xset, na, stride)
        if key not in self.buf:
            self.buf[key] = (torch.zeros(na*stride, dtype=torch.float64, device=self.tdev),
                             torch.zeros(self.world*na*stride, dtype=torch.float64, device=self.tdev))
        mine, allb = self.buf[key]
        if self.tdev.type == "cuda":
            if self.ext is None:
                self.ext = torch.cuda.ExternalStream(self.dev.stream_handle(), device=self.tdev)
            with torch.cuda.stream(self.ext):       # RCCL orders its work against the current stream
                self.dev.shard_pack(xset, mine.data_ptr(), stride)
                if self.simulate:
                    pass
                elif dist.get_backend() == "gloo":
                    # functional-test path (several ranks sharing one GPU): gloo moves host memory
                    hm, ha = mine.cpu(), torch.empty(allb.shape, dtype=allb.dtype)
                    dist.all_gather_into_tensor(ha, hm)
                    allb.copy_(ha)
                else:
                    dist.all_gather_into_tensor(allb, mine)
                self.dev.shard_unpack_all(xset, allb.data_ptr(), stride)
            return
        self.dev.shard_pack(xset, mine.data_ptr(), stride)
        dist.all_gather_into_tensor(allb, mine)
        self.dev.shard_unpack_all(xset, allb.data_ptr(), stride)


class ShardedRunner:
    def __init__(self, sim, rank, world, simulate=False):
        self.sim, self.rank, self.world = sim, rank, world
        self.simulate = simulate
        self.dev = None
        self.x = None

    def _attach(self):
        self.dev = self.sim.device()
        self.dev.set_shard(self.rank, self.world)
        self.x = Exchanger(self.dev, self.rank, self.world, torch.device("cuda", torch.cuda.current_device()))
        self.x.simulate = self.simulate

    def setup(self):
        """SphSimulation::PostInitialConditionsSetup (SphSimulation.cpp:204-565), sliced"""
        if self.world == 1:
            self.sim.post_ic_setup()
            self._attach()
            return
        ic = self.sim.generate_ic() if self.sim.lib.gah_num_particles(self.sim.h) == 0 else None
        del ic
        self.sim.upload_ic()
        self._attach()
        d = self.dev
        npass = 2 if self.sim.initial_h_provided() else 3
        for _ in range(npass):
            d.build_tree()
            d.update_density()
            self.x.exchange(d.X_DENSITY)
            d.update_hmax()
        d.zero_accelerations()
        d.update_forces()
        self.x.exchange(d.X_FORCES)
        d.set_time(0.0, 0.0)
        d.compute_global_timestep()
        d.kdk_end(0, 0.0, 0.0)

    def steps(self, n):
        if n <= 0:
            return
        if self.world == 1:
            self.sim.main_loop(n)
            return
        d = self.dev
        for _ in range(n):
            d.step_begin()
            self.x.exchange(d.X_DENSITY)
            d.step_forces()
            self.x.exchange(d.X_FORCES)
            if self.simulate:
                try:                 # the other ranks' slices are stale: recoverable warnings are expected
                    d.step_end()
                except Exception as e:      # noqa: BLE001
                    if getattr(e, "code", -1) <= 0:
                        raise
            else:
                d.step_end()

    def count_density(self):
        """counters of one density pass over this rank's slice, on the current state (instrumented build)"""
        dev = self.dev
        h = dev.download("h")
        st = dev.update_density(stats=True)
        dev.upload_field("h", h)       # leave h as it was; the next step rebuilds everything else
        return st

    def count_forces(self):
        dev = self.dev
        saved = {k: dev.download(k) for k in ("a", "atree", "gpot", "gpot_hydro", "dudt", "div_v")}
        dev.zero_accelerations()
        st = dev.update_forces(stats=True)
        for k, v in saved.items():
            dev.upload_field(k, v)
        return st
