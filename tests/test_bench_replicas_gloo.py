"""
The N > 1 bookkeeping of bench.py on CPU (gloo, two processes): the launch plans rank 0 tuned are broadcast and pinned on every rank,
every rank reports its verification step, and the bit checksums tell identical replicas from a rank that drifted. The GPU work itself is
replaced by stand-ins (no GPU here); what runs is the code the driver's first multi-GPU bench run will execute around it.
"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Ctx:
    """ stand-in for _capi.Context: rank 0 'tunes' rank-dependent plans, the others would pick something else on their own """

    def __init__(self, rank):
        self.rank, self.autotune, self.pinned = rank, True, {}

    def query_plan(self, grid, flags, fam):
        return {"rows": 1 + fam + 10 * self.rank, "tpr": 16 << (fam % 3), "chunk": 32 + fam + self.rank}

    def set_autotune(self, on):
        self.autotune = bool(on)

    def set_tuning_kernel(self, fam, rows, tpr, chunk):
        self.pinned[fam] = (rows, tpr, chunk)


class _Sim:
    grid = None

    def __init__(self):
        self.steps, self.resets = 0, 0

    def step(self, allreduce):
        self.steps += 1

    def reset(self):
        self.resets += 1


def _worker(rank, world, port, drift):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    dev = torch.device("cpu")
    ctx, sim = _Ctx(rank), _Sim()
    plans = bench.sync_launch_plans(ctx, sim, dist, rank, dev)
    assert ctx.autotune is False and set(ctx.pinned) == {0, 1, 2, 3}
    assert all(ctx.pinned[f] == (1 + f, 16 << (f % 3), 32 + f) for f in range(4)), ctx.pinned          # rank 0's plans on every rank
    assert plans == {f"family{f}": [1 + f, 16 << (f % 3), 32 + f] for f in range(4)}
    assert (sim.steps, sim.resets) == ((1, 1) if rank == 0 else (0, 0))                                  # only rank 0 spent a step on tuning
    g = torch.Generator().manual_seed(7)
    p = torch.randn(1, 6, 5, 4, generator=g)
    v = [torch.randn(1, 6, 5, 4, generator=g) for _ in range(3)]
    if drift and rank == 1:
        v[2].view(torch.int32)[0, 0, 0, 0] += 1                                                           # one ulp in one sample
    its, rep = bench.gather_replicas(dist, world, [100], True, [p] + v, dev, plans)
    assert its == [[100]] * world and rep["verified_ok"] == [True] * world
    assert rep["all_bit_identical"] is (not drift) and rep["bit_identical_to_rank0"] == [True, not drift]
    assert rep["pressure_norm_rel_diff_vs_rank0"] == [0.0, 0.0] and rep["pinned_launch_plans"] == plans
    its, rep = bench.gather_replicas(dist, world, [100 - rank], rank == 0, [p] + v, dev, plans)       # a rank that stopped early is reported
    assert its == [[100], [99]] and rep["verified_ok"] == [True, False]
    # the sharded batch (`--workload config4`): 5 entries over 2 ranks = 3 + 2; entry 0 recomputed on every rank must agree bit for bit
    total = 5
    own = [torch.full((3 - rank, 4, 4), float(rank + 1))]
    ref = [p.clone()] + [t.clone() for t in v]                                                             # (v[2] of rank 1 carries the drift)
    sh = bench.gather_shards(dist, world, total, [50] * (3 - rank), True, own, ref, dev)
    assert sh["iterations_per_rank"] == [[50, 50, 50], [50, 50]] and sh["entries_per_rank"] == [3, 2] and sh["verified_ok"] == [True, True]
    assert sh["entry0_all_bit_identical"] is (not drift) and sh["entry0_bit_identical_to_rank0"] == [True, not drift]
    assert sh["owned_checksums"][0] != sh["owned_checksums"][1]
    sh = bench.gather_shards(dist, world, total, [50] * (3 - rank - (rank == 1)), rank == 0, own, ref, dev)     # rank 1 lost an entry / stopped early
    assert sh["iterations_per_rank"] == [[50, 50, 50], [50]] and sh["verified_ok"] == [True, False]
    assert bench.gather_shards(None, 1, 2, [7, 7], True, own, ref, dev)["iterations_per_rank"] == [[7, 7]]      # one process, no collective
    dist.barrier()
    dist.destroy_process_group()


def test_bench_replica_validation_world2_gloo():
    for drift in (False, True):
        mp.spawn(_worker, args=(2, _free_port(), drift), nprocs=2, join=True)
