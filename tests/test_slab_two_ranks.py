"""
SURVEY §8 f4 with world > 1 ON A GPU. The boxes of this project have ONE GPU and RCCL refuses two ranks on one device
("duplicate GPU"), so until r5 the slab classes had met real asynchronous HIP streams with one rank only. Here TWO processes share cuda:0,
talk over a gloo process group and hand `SlabSolver` / `SlabFluid` DEVICE tensors (phiflow_amd/slab.py post_p2p stages the point-to-point
messages of that backend through host copies; the all-reduces are gloo's own device-tensor path). Asserted:
  * the slab CG (halo planes from BOTH neighbours on a periodic axis, a wall below / an open end above otherwise) reproduces the
    single-process solve of the same GPU (same iteration counts +- 1, values to rounding: the dot products are summed in another order);
  * the whole step (`SlabFluid.step`) reproduces the single-process step;
  * `SlabFluid(overlap=True)` -- the whole slab advected on empty ghosts while the exchange is in flight, the planes next to a cut redone on
    windows -- returns the SAME BITS as the plain order on every rank, on kernels that really run asynchronously to the host.
The same workers run on the CPU emulation (world 2, `-m "not gpu"`) so that the test logic itself is exercised without a GPU.
Reference: no counterpart (phi has no domain decomposition); the operators are phi/physics/fluid.py:94-162, advect.py:156-179.
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

CASES = {
    # (res, bc) -- 0 periodic, 1 closed, 2 open (phihip_grid.bc); x is the decomposed axis
    "periodic": dict(res=(32, 24, 64), bc=((0, 0), (0, 0), (0, 0))),
    "closed_open": dict(res=(30, 20, 72), bc=((1, 2), (1, 1), (0, 0))),
}
CASES_EMU = {
    "periodic": dict(res=(30, 6, 12), bc=((0, 0), (0, 0), (0, 0))),
    "closed_open": dict(res=(30, 6, 8), bc=((1, 2), (1, 1), (0, 0))),
}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _backend(lib_path, device):
    from phiflow_amd import _capi
    from phiflow_amd.backend import HipBackend
    if device == "cpu":
        return HipBackend(library=_capi.Library(lib_path), device="cpu")
    torch.cuda.set_device(0)
    return HipBackend(device="cuda:0")


def _smooth_velocity(ctx, grid, rng, batch):
    comps = []
    for c in range(3):
        shape = ctx.component_shape(grid, c)
        idx = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
        ph = rng.uniform(0, 2 * np.pi, size=(batch, 3))
        field = np.stack([0.2 * (np.sin(2 * np.pi * idx[0] / shape[0] + ph[b, 0]) + np.cos(2 * np.pi * idx[1] / shape[1] + ph[b, 1])
                                 + np.sin(4 * np.pi * idx[2] / shape[2] + ph[b, 2])) for b in range(batch)])
        comps.append(np.ascontiguousarray(field.astype(np.float32)))
    return comps


def _problem(cases, case, batch=2):
    from phiflow_amd import _capi as C
    res, bc = cases[case]["res"], cases[case]["bc"]
    grid = C.make_grid(3, C.PHIHIP_F32, batch, res, (0, 0, 0), tuple(float(r) for r in res), bc)
    rhs = np.random.default_rng(3).standard_normal((batch,) + res).astype(np.float32)
    if all(c != 2 for pair in bc for c in pair):
        rhs -= rhs.mean(axis=(1, 2, 3), keepdims=True)
    return res, bc, grid, rhs


def _worker(rank, world, port, lib_path, device, out_dir, case, emu):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PHIHIP_AUTOTUNE"] = "0"            # the same launch geometry in every process (summation order)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phiflow_amd import _capi
    from phiflow_amd.slab import SlabFluid, SlabSolver
    be = _backend(lib_path, device)
    dev = be.device
    cases = CASES_EMU if emu else CASES
    res, bc, grid, rhs = _problem(cases, case)
    B = grid.batch
    box = ((0.0, 0.0, 0.0), tuple(float(r) for r in res))
    # (1) the slab CG
    solver = SlabSolver(be, res, box[0], box[1], bc, torch.float32, batch=B)
    b0, b1 = solver.begin, solver.end
    x = torch.zeros((B, b1 - b0) + res[1:], dtype=torch.float32, device=dev)
    infos = solver.solve(torch.from_numpy(np.ascontiguousarray(rhs[:, b0:b1])).to(dev), x, rel_tol=1e-5, max_iterations=80, refresh_every=7, check_every=5)
    # (2) the step, plain order and overlapped: same bits
    v = _smooth_velocity(be.ctx, grid, np.random.default_rng(11), B)
    off = 0 if bc[0][0] != _capi.BC_CLOSED else 1
    adv, steps = [], []
    for overlap in (False, True):
        fl = SlabFluid(be, res, box[0], box[1], bc, torch.float32, batch=B, overlap=overlap)
        own = [torch.from_numpy(np.ascontiguousarray(v[0][:, fl.face_begin - off: fl.face_end - off])).to(dev),
               torch.from_numpy(np.ascontiguousarray(v[1][:, fl.begin: fl.end])).to(dev), torch.from_numpy(np.ascontiguousarray(v[2][:, fl.begin: fl.end])).to(dev)]
        for _ in range(3):           # (repeated: an ordering fault between the exchange and the kernels need not show on the first call)
            a = fl.advect(own, 0.9)
        adv.append(a)
        p = torch.zeros(fl.cell_shape, dtype=torch.float32, device=dev)
        out, sinfo = fl.step(own, p, 0.9, rel_tol=1e-5, max_iterations=200)
        steps.append((out, p, sinfo))
        n_windows = len(fl._overlap_windows() or []) if overlap else 0
        expected = int(fl.lo_rank is not None) + int(fl.hi_rank is not None)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), b0=b0, b1=b1, x=x.cpu().numpy(), it=[i.iterations for i in infos], conv=[i.converged for i in infos],
             f0=fl.face_begin - off, f1=fl.face_end - off, windows=n_windows, expected=expected,
             same_adv=[bool(torch.equal(a, b)) for a, b in zip(*adv)], same_out=[bool(torch.equal(a, b)) for a, b in zip(steps[0][0], steps[1][0])],
             same_p=bool(torch.equal(steps[0][1], steps[1][1])),
             **{f"adv{c}": adv[1][c].cpu().numpy() for c in range(3)}, **{f"out{c}": steps[1][0][c].cpu().numpy() for c in range(3)},
             p=steps[1][1].cpu().numpy(), step_it=[i.iterations for i in steps[1][2]])
    dist.barrier()
    dist.destroy_process_group()


def _check(ctx, to_dev, to_host, tmp_path, cases, case, world):
    """ the single-process run on the same library / device and the comparison """
    from phiflow_amd import _capi as C
    res, bc, grid, rhs = _problem(cases, case)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r, q in enumerate(parts):
        assert int(q["windows"]) == int(q["expected"]) > 0, f"rank {r}: {int(q['windows'])} windows for {int(q['expected'])} cut sides"
        assert all(bool(s) for s in q["same_adv"]), f"rank {r}: overlapped advection differs from the plain order: {list(q['same_adv'])}"
        assert all(bool(s) for s in q["same_out"]) and bool(q["same_p"]), f"rank {r}: the step with the overlapped exchange differs from the plain order"
    cat = lambda key: np.concatenate([q[key] for q in parts], axis=1)
    P = lambda ts: [t.data_ptr() if hasattr(t, "data_ptr") else t.ctypes.data for t in ts]
    # CG
    drhs, dx = to_dev(rhs), to_dev(np.zeros_like(rhs))
    info = ctx.cg_solve(grid, 0, 1, P([drhs])[0], P([dx])[0], C.Solve(1e-5, 0.0, 80, 7, 5, 0))
    x_ref = to_host(dx)
    assert int(parts[0]["b0"]) == 0 and int(parts[-1]["b1"]) == res[0]
    assert all(list(parts[0]["it"]) == list(q["it"]) for q in parts)
    assert all(abs(int(a) - i.iterations) <= 1 for a, i in zip(parts[0]["it"], info)), (list(parts[0]["it"]), [i.iterations for i in info])
    assert np.abs(cat("x") - x_ref).max() <= 2e-4 * np.abs(x_ref).max()
    # step
    v = _smooth_velocity(ctx, grid, np.random.default_rng(11), grid.batch)
    singular = all(c != 2 for pair in bc for c in pair)
    dv = [to_dev(a) for a in v]
    dadv = [to_dev(np.empty_like(a)) for a in v]
    ctx.advect_staggered(grid, P(dv), P(dv), P(dadv), 0.9)
    ddiv = to_dev(np.empty((grid.batch,) + res, np.float32))
    ctx.divergence(grid, P(dadv), 0, 1, singular, P([ddiv])[0])
    dp = to_dev(np.zeros((grid.batch,) + res, np.float32))
    sinfo = ctx.cg_solve(grid, 0, 1, P([ddiv])[0], P([dp])[0], C.Solve(1e-5, 0.0, 200, 50, 10, 0))
    adv = [to_host(a) for a in dadv]
    ctx.grad_subtract(grid, 0, 1, P([dp])[0], P(dadv))
    out, p = [to_host(a) for a in dadv], to_host(dp)
    for c in range(3):
        assert cat(f"adv{c}").shape == adv[c].shape and np.abs(cat(f"adv{c}") - adv[c]).max() <= 2e-5, (c, np.abs(cat(f"adv{c}") - adv[c]).max())
    assert all(abs(int(a) - i.iterations) <= 1 for a, i in zip(parts[0]["step_it"], sinfo))
    pr, pm = cat("p"), p
    if singular:
        pr, pm = pr - pr.mean(axis=(1, 2, 3), keepdims=True), pm - pm.mean(axis=(1, 2, 3), keepdims=True)
    assert np.abs(pr - pm).max() <= 5e-4 * np.abs(pm).max(), np.abs(pr - pm).max() / np.abs(pm).max()
    for c in range(3):
        assert np.abs(cat(f"out{c}") - out[c]).max() <= 2e-4, (c, np.abs(cat(f"out{c}") - out[c]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_two_ranks_on_one_gpu_slab_solver_and_overlapped_step(gpu_backend, tmp_path, case):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), "", "cuda", str(tmp_path), case, False), nprocs=world, join=True)
    dev = gpu_backend.device
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def to_host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()
    os.environ["PHIHIP_AUTOTUNE"] = "0"
    from phiflow_amd import _capi
    ctx = _capi.Context(gpu_backend.library, 0)          # a context of its own: the analytic launch plans, like the workers
    ctx.set_autotune(False)
    _check(ctx, to_dev, to_host, tmp_path, CASES, case, world)


@pytest.mark.parametrize("case", list(CASES_EMU))
def test_two_ranks_slab_workers_on_the_emulation(emu_library, emu_ctx, tmp_path, case):
    """ the same workers and checks on the CPU emulation (host tensors travel through gloo directly): the test logic runs in CI """
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), emu_library.path, "cpu", str(tmp_path), case, True), nprocs=world, join=True)
    _check(emu_ctx, lambda a: np.ascontiguousarray(a).copy(), lambda a: a.copy(), tmp_path, CASES_EMU, case, world)
