"""
N > 1 path on CPU: two processes (gloo) shard a batch of independent simulations, run the projection on their shard through
the C ABI (kernel sources under the fiber emulation -- test infrastructure) and perform the step's single all-reduce of the
residual norm. The sharded results must equal the unsharded run bit for bit.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_fields(backend, batch):
    from phiflow_amd.flow import Box, StaggeredGrid, ZERO
    rng = np.random.default_rng(0)
    bounds = Box['x,y', 0:100, 0:100]
    vx = rng.standard_normal((batch, 15, 20)).astype(np.float32) * 0.1
    vy = rng.standard_normal((batch, 16, 19)).astype(np.float32) * 0.1
    return StaggeredGrid([vx, vy], ZERO, bounds, x=16, y=20, backend=backend)


def _worker(rank, world, port, emu_path, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phiflow_amd import _capi
    from phiflow_amd.backend import HipBackend
    from phiflow_amd.flow import Solve, fluid
    from phiflow_amd.parallel import global_relative_residual, local_batch_range, shard_field
    backend = HipBackend(library=_capi.Library(emu_path), device="cpu")
    total = 5                                        # uneven split: 3 + 2
    full = _make_fields(backend, total)
    mine = shard_field(full, rank, world)
    b0, b1 = local_batch_range(total, rank, world)
    assert mine.batch_size == b1 - b0
    v, p = fluid.make_incompressible(mine, (), Solve('CG', 1e-5, 0))
    rel = global_relative_residual(backend, mine.batch_size)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), vx=v.values[0].numpy(), vy=v.values[1].numpy(), p=p.values.numpy(),
             rel=rel.numpy(), b0=b0, b1=b1)
    dist.barrier()
    dist.destroy_process_group()


def test_batch_sharding_world2_gloo(emu_library, emu_backend, tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, emu_library.path, str(tmp_path)), nprocs=world, join=True)
    from phiflow_amd.flow import Solve, fluid
    from phiflow_amd.parallel import global_relative_residual
    full = _make_fields(emu_backend, 5)
    v, p = fluid.make_incompressible(full, (), Solve('CG', 1e-5, 0))
    rel_full = float(global_relative_residual(emu_backend, 5)[0])
    rels = []
    for rank in range(world):
        d = np.load(tmp_path / f"rank{rank}.npz")
        b0, b1 = int(d["b0"]), int(d["b1"])
        assert np.array_equal(d["vx"], v.values[0].numpy()[b0:b1])
        assert np.array_equal(d["vy"], v.values[1].numpy()[b0:b1])
        assert np.array_equal(d["p"], p.values.numpy()[b0:b1])
        rels.append(float(d["rel"][0]))
    assert rels[0] == rels[1] == pytest.approx(rel_full, rel=1e-12)      # the all-reduce delivered the global maximum
    assert 0 < rel_full <= 1e-5
