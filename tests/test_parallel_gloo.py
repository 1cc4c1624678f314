"""
N > 1 paths on CPU. (1) Two processes (gloo) shard a batch of independent simulations, run the projection on their shard through
the C ABI (kernel sources under the fiber emulation -- test infrastructure) and perform the step's single all-reduce of the
residual norm. The sharded results must equal the unsharded run bit for bit. (2) Two processes split ONE simulation into
x-slabs: halo-plane exchange + two scalar all-reduces per CG iteration reproduce the single-process solve.
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


# ---------------------------------------------------------------------------------------------------------------------
# ONE simulation decomposed into x-slabs over the ranks (halo-plane exchange + 2 scalar all-reduces per iteration)
# ---------------------------------------------------------------------------------------------------------------------
SLAB_CASES = {
    "periodic": dict(res=(12, 8, 16), bc=((0, 0), (0, 0), (0, 0))),              # both slab sides are halos on every rank (wrap)
    "closed_open": dict(res=(11, 8, 16), bc=((1, 2), (1, 1), (0, 0))),            # uneven split (6 + 5), wall / open ends on x
}


def _slab_problem(case, dtype=np.float32, batch=2):
    rng = np.random.default_rng(3)
    res, bc = SLAB_CASES[case]["res"], SLAB_CASES[case]["bc"]
    rhs = rng.standard_normal((batch,) + res).astype(dtype)
    if all(c != 2 for pair in bc for c in pair):          # singular (no open side): consistent right-hand side
        rhs -= rhs.mean(axis=(1, 2, 3), keepdims=True)
    return res, bc, rhs


def _slab_worker(rank, world, port, emu_path, out_dir, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phiflow_amd import _capi
    from phiflow_amd.backend import HipBackend
    from phiflow_amd.slab import SlabSolver
    backend = HipBackend(library=_capi.Library(emu_path), device="cpu")
    res, bc, rhs = _slab_problem(case)
    solver = SlabSolver(backend, res, (0.0, 0.0, 0.0), tuple(float(r) for r in res), bc, torch.float32, batch=rhs.shape[0])
    b0, b1 = solver.begin, solver.end
    x = torch.zeros((rhs.shape[0], b1 - b0) + res[1:], dtype=torch.float32)
    infos = solver.solve(torch.from_numpy(np.ascontiguousarray(rhs[:, b0:b1])), x, rel_tol=1e-5, max_iterations=60, refresh_every=7, check_every=5)
    np.savez(os.path.join(out_dir, f"slab{rank}.npz"), x=x.numpy(), b0=b0, b1=b1, it=[i.iterations for i in infos],
             conv=[i.converged for i in infos], rsq=[i.residual_sq for i in infos])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [(c, 2) for c in SLAB_CASES] + [("closed_open", 3), ("periodic", 3)])
def test_slab_decomposed_cg_world2_gloo(emu_library, emu_ctx, tmp_path, case, world):
    """ two (three) ranks, each with a share of the x planes, must reproduce the single-process solve (same iteration count; the dot
    products are summed in a different order, so values agree to rounding) """
    from phiflow_amd import _capi as C
    mp.spawn(_slab_worker, args=(world, _free_port(), emu_library.path, str(tmp_path), case), nprocs=world, join=True)
    res, bc, rhs = _slab_problem(case)
    grid = C.make_grid(3, C.PHIHIP_F32, rhs.shape[0], res, (0, 0, 0), tuple(float(r) for r in res), bc)
    x_ref = np.zeros_like(rhs)
    info = emu_ctx.cg_solve(grid, 0, 1, rhs.ctypes.data, x_ref.ctypes.data, C.Solve(1e-5, 0.0, 60, 7, 5, 0))
    parts = [np.load(tmp_path / f"slab{r}.npz") for r in range(world)]
    assert int(parts[0]["b0"]) == 0 and int(parts[-1]["b1"]) == res[0] and all(int(parts[r]["b1"]) == int(parts[r + 1]["b0"]) for r in range(world - 1))
    x = np.concatenate([p["x"] for p in parts], axis=1)
    assert all(list(parts[0]["it"]) == list(q["it"]) for q in parts)       # every rank took the same (global) decisions
    # the dot products are summed in a different order: an entry may cross the tolerance one iteration earlier or later
    assert all(abs(int(a) - i.iterations) <= 1 for a, i in zip(parts[0]["it"], info))
    assert list(parts[0]["conv"]) == [i.converged for i in info]
    scale = np.abs(x_ref).max()
    assert np.abs(x - x_ref).max() <= 2e-4 * scale, np.abs(x - x_ref).max() / scale


def test_slab_solver_single_rank_equals_cg(emu_backend, emu_ctx):
    """ world size 1: the slab phases are the ordinary CG (no halos) """
    from phiflow_amd import _capi as C
    from phiflow_amd.slab import SlabSolver
    res, bc, rhs = _slab_problem("closed_open")
    solver = SlabSolver(emu_backend, res, (0.0, 0.0, 0.0), tuple(float(r) for r in res), bc, torch.float32, batch=rhs.shape[0])
    x = torch.zeros(rhs.shape, dtype=torch.float32)
    infos = solver.solve(torch.from_numpy(rhs.copy()), x, rel_tol=1e-5, max_iterations=200)
    grid = C.make_grid(3, C.PHIHIP_F32, rhs.shape[0], res, (0, 0, 0), tuple(float(r) for r in res), bc)
    x_ref = np.zeros_like(rhs)
    info = emu_ctx.cg_solve(grid, 0, 1, rhs.ctypes.data, x_ref.ctypes.data, C.Solve(1e-5, 0.0, 200, 50, 10, 0))
    assert [i.iterations for i in infos] == [i.iterations for i in info] and all(i.converged for i in infos)
    assert np.abs(x.numpy() - x_ref).max() <= 1e-6 * np.abs(x_ref).max()


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY §8 f4, the whole step on slabs: advection, divergence and gradient run on the slab extended by ghost planes of the neighbours
# ---------------------------------------------------------------------------------------------------------------------
def _fluid_problem(case, dtype=np.float32, batch=2):
    from phiflow_amd import _capi as C
    res, bc = SLAB_CASES[case]["res"], SLAB_CASES[case]["bc"]
    grid = C.make_grid(3, C.PHIHIP_F32, batch, res, (0, 0, 0), tuple(float(r) for r in res), bc)
    rng = np.random.default_rng(11)
    return res, bc, grid, rng


def _smooth_velocity(ctx, grid, rng, batch):
    """ a smooth velocity with |u| <= 0.6 cells per unit time on the staggered layout of `grid` """
    comps = []
    for c in range(3):
        shape = ctx.component_shape(grid, c)
        idx = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
        ph = rng.uniform(0, 2 * np.pi, size=(batch, 3))
        field = np.stack([0.2 * (np.sin(2 * np.pi * idx[0] / shape[0] + ph[b, 0]) + np.cos(2 * np.pi * idx[1] / shape[1] + ph[b, 1])
                                 + np.sin(4 * np.pi * idx[2] / shape[2] + ph[b, 2])) for b in range(batch)])
        comps.append(np.ascontiguousarray(field.astype(np.float32)))
    return comps


def _reference_step(ctx, grid, v, dt, singular):
    from phiflow_amd import _capi as C
    P = lambda ts: [t.ctypes.data for t in ts]
    adv = [np.empty_like(t) for t in v]
    ctx.advect_staggered(grid, P(v), P(v), P(adv), dt)
    div = np.empty((grid.batch,) + tuple(grid.res[d] for d in range(3)), np.float32)
    ctx.divergence(grid, P(adv), 0, 1, singular, div.ctypes.data)
    p = np.zeros_like(div)
    info = ctx.cg_solve(grid, 0, 1, div.ctypes.data, p.ctypes.data, C.Solve(1e-5, 0.0, 200, 50, 10, 0))
    out = [t.copy() for t in adv]
    ctx.grad_subtract(grid, 0, 1, p.ctypes.data, P(out))
    return adv, div, p, out, info


def _fluid_worker(rank, world, port, emu_path, out_dir, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PHIHIP_AUTOTUNE"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phiflow_amd import _capi
    from phiflow_amd.backend import HipBackend
    from phiflow_amd.slab import SlabFluid
    backend = HipBackend(library=_capi.Library(emu_path), device="cpu")
    res, bc, grid, rng = _fluid_problem(case)
    v = _smooth_velocity(backend.ctx, grid, rng, grid.batch)
    fluid = SlabFluid(backend, res, (0.0, 0.0, 0.0), tuple(float(r) for r in res), bc, torch.float32, batch=grid.batch)
    off = 0 if bc[0][0] != _capi.BC_CLOSED else 1          # global x face number of stored index 0
    own = [torch.from_numpy(np.ascontiguousarray(v[0][:, fluid.face_begin - off: fluid.face_end - off])),
           torch.from_numpy(np.ascontiguousarray(v[1][:, fluid.begin: fluid.end])), torch.from_numpy(np.ascontiguousarray(v[2][:, fluid.begin: fluid.end]))]
    adv = fluid.advect(own, 0.9)
    div = fluid.divergence(adv, balance=all(c != 2 for pair in bc for c in pair))
    p = torch.zeros(fluid.cell_shape, dtype=torch.float32)
    out, infos = fluid.step(own, p, 0.9, rel_tol=1e-5, max_iterations=200)
    try:                                               # |u_x| dt / dx > ghost - 1: refused instead of rank-dependent values (the exchange itself still completes)
        fluid.advect(own, 6.0)
        refused = False
    except ValueError:
        refused = True
    np.savez(os.path.join(out_dir, f"fluid{rank}.npz"), refused=refused, f0=fluid.face_begin - off, f1=fluid.face_end - off, b0=fluid.begin, b1=fluid.end,
             adv0=adv[0].numpy(), adv1=adv[1].numpy(), adv2=adv[2].numpy(), div=div.numpy(), p=p.numpy(),
             out0=out[0].numpy(), out1=out[1].numpy(), out2=out[2].numpy(), it=[i.iterations for i in infos])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [(c, 2) for c in SLAB_CASES] + [("closed_open", 3)])
def test_slab_decomposed_fluid_step_world2_gloo(emu_library, emu_ctx, tmp_path, case, world):
    """ two ranks, each owning half of the x planes (and three ranks with 4 + 4 + 3 planes: the middle one has ghost planes on both
    sides), reproduce advection, divergence, pressure and the projected velocity of the single-process step (ghost-plane exchange;
    sample coordinates are rounded at local instead of global index magnitudes) """
    mp.spawn(_fluid_worker, args=(world, _free_port(), emu_library.path, str(tmp_path), case), nprocs=world, join=True)
    res, bc, grid, rng = _fluid_problem(case)
    v = _smooth_velocity(emu_ctx, grid, rng, grid.batch)
    singular = all(c != 2 for pair in bc for c in pair)
    adv, div, p, out, info = _reference_step(emu_ctx, grid, v, 0.9, singular)
    parts = [np.load(tmp_path / f"fluid{r}.npz") for r in range(world)]
    assert any(bool(q["refused"]) for q in parts), "a back-trace of > ghost - 1 cells along x must be refused"
    assert int(parts[0]["f0"]) == 0 and int(parts[-1]["f1"]) == v[0].shape[1]
    assert all(int(parts[r]["f1"]) == int(parts[r + 1]["f0"]) for r in range(world - 1))
    cat = lambda key: np.concatenate([q[key] for q in parts], axis=1)
    for c in range(3):
        assert cat(f"adv{c}").shape == adv[c].shape
        assert np.abs(cat(f"adv{c}") - adv[c]).max() <= 2e-5, (c, np.abs(cat(f"adv{c}") - adv[c]).max())
    assert np.abs(cat("div") - div).max() <= 2e-5 * max(1.0, np.abs(div).max())
    assert all(abs(int(a) - i.iterations) <= 1 for a, i in zip(parts[0]["it"], info))
    pr, pm = cat("p"), p
    if singular:
        pr, pm = pr - pr.mean(axis=(1, 2, 3), keepdims=True), pm - pm.mean(axis=(1, 2, 3), keepdims=True)
    assert np.abs(pr - pm).max() <= 5e-4 * np.abs(pm).max(), np.abs(pr - pm).max() / np.abs(pm).max()
    for c in range(3):
        assert np.abs(cat(f"out{c}") - out[c]).max() <= 2e-4, (c, np.abs(cat(f"out{c}") - out[c]).max())


OVERLAP_CASES = {
    "periodic": dict(res=(30, 8, 12), bc=((0, 0), (0, 0), (0, 0))),               # 15 + 15 planes: both cut sides of every rank get a window
    "closed_open": dict(res=(45, 6, 10), bc=((1, 2), (1, 1), (0, 0))),            # three ranks 15 + 15 + 15: wall below rank 0, open end above rank 2
}


def _overlap_worker(rank, world, port, emu_path, out_dir, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PHIHIP_AUTOTUNE"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phiflow_amd import _capi
    from phiflow_amd.backend import HipBackend
    from phiflow_amd.slab import SlabFluid
    backend = HipBackend(library=_capi.Library(emu_path), device="cpu")
    res, bc = OVERLAP_CASES[case]["res"], OVERLAP_CASES[case]["bc"]
    grid = _capi.make_grid(3, _capi.PHIHIP_F32, 2, res, (0, 0, 0), tuple(float(r) for r in res), bc)
    v = _smooth_velocity(backend.ctx, grid, np.random.default_rng(11), grid.batch)
    results = []
    for overlap in (False, True):
        fluid = SlabFluid(backend, res, (0.0, 0.0, 0.0), tuple(float(r) for r in res), bc, torch.float32, batch=grid.batch, overlap=overlap)
        off = 0 if bc[0][0] != _capi.BC_CLOSED else 1
        own = [torch.from_numpy(np.ascontiguousarray(v[0][:, fluid.face_begin - off: fluid.face_end - off])),
               torch.from_numpy(np.ascontiguousarray(v[1][:, fluid.begin: fluid.end])), torch.from_numpy(np.ascontiguousarray(v[2][:, fluid.begin: fluid.end]))]
        results.append(fluid.advect(own, 0.9))
        n_windows = len(fluid._overlap_windows() or []) if overlap else 0
    np.savez(os.path.join(out_dir, f"ovl{rank}.npz"), windows=n_windows, expected=int(fluid.lo_rank is not None) + int(fluid.hi_rank is not None),
             same=[bool(torch.equal(a, b)) for a, b in zip(*results)], adv0=results[1][0].numpy(), adv1=results[1][1].numpy(), adv2=results[1][2].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("periodic", 2), ("closed_open", 3)])
def test_slab_advection_overlapped_exchange_gloo(emu_library, emu_ctx, tmp_path, case, world):
    """ SlabFluid(overlap=True): the ghost exchange is posted, the whole slab is advected on empty ghosts while it is in flight, the planes
    within reach of a cut are redone on windows of the completed arrays -- the SAME bits as the plain order on every rank, and the
    single-process advection within rounding """
    from phiflow_amd import _capi as C
    mp.spawn(_overlap_worker, args=(world, _free_port(), emu_library.path, str(tmp_path), case), nprocs=world, join=True)
    res, bc = OVERLAP_CASES[case]["res"], OVERLAP_CASES[case]["bc"]
    grid = C.make_grid(3, C.PHIHIP_F32, 2, res, (0, 0, 0), tuple(float(r) for r in res), bc)
    v = _smooth_velocity(emu_ctx, grid, np.random.default_rng(11), grid.batch)
    adv = [np.empty_like(t) for t in v]
    emu_ctx.advect_staggered(grid, [t.ctypes.data for t in v], [t.ctypes.data for t in v], [t.ctypes.data for t in adv], 0.9)
    parts = [np.load(tmp_path / f"ovl{r}.npz") for r in range(world)]
    for r, q in enumerate(parts):
        assert int(q["windows"]) == int(q["expected"]) > 0, f"rank {r}: {int(q['windows'])} windows for {int(q['expected'])} cut sides"
        assert all(bool(x) for x in q["same"]), f"rank {r}: overlapped advection differs from the plain order: {list(q['same'])}"
    for c in range(3):
        got = np.concatenate([q[f"adv{c}"] for q in parts], axis=1)
        assert got.shape == adv[c].shape and np.abs(got - adv[c]).max() <= 2e-5, (c, np.abs(got - adv[c]).max())


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY §8 f4 with obstacles (fluid.py:130-137,212-240 on slabs): masks rasterised per rank, ghost-cell masks from the owner, flags on the
# extended grid, apply_boundary_conditions after the advection, balance over the active cells of the WHOLE domain
# ---------------------------------------------------------------------------------------------------------------------
def _slab_obstacles(res):
    from phiflow_amd import _capi as C
    # a box that straddles every cut of a 2- and 3-rank split and touches neither wall; a moving, rotating sphere near the lower x end
    return [dict(kind=C.OBSTACLE_BOX, center=(0.5 * res[0] + 0.3, 0.5 * res[1], 0.45 * res[2]), half_size=(2.2, 1.6, 3.1)),
            dict(kind=C.OBSTACLE_SPHERE, center=(2.4, 0.4 * res[1], 0.7 * res[2]), half_size=(1.3, 0, 0), velocity=(0.2, -0.1, 0.0),
                 angular_velocity=(0.0, 0.0, 0.3))]


def _obstacle_worker(rank, world, port, emu_path, out_dir, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PHIHIP_AUTOTUNE"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phiflow_amd import _capi
    from phiflow_amd.backend import HipBackend
    from phiflow_amd.slab import SlabFluid
    backend = HipBackend(library=_capi.Library(emu_path), device="cpu")
    res, bc, grid, rng = _fluid_problem(case)
    v = _smooth_velocity(backend.ctx, grid, rng, grid.batch)
    fluid = SlabFluid(backend, res, (0.0, 0.0, 0.0), tuple(float(r) for r in res), bc, torch.float32, batch=grid.batch, obstacles=_slab_obstacles(res))
    off = 0 if bc[0][0] != _capi.BC_CLOSED else 1
    own = [torch.from_numpy(np.ascontiguousarray(v[0][:, fluid.face_begin - off: fluid.face_end - off])),
           torch.from_numpy(np.ascontiguousarray(v[1][:, fluid.begin: fluid.end])), torch.from_numpy(np.ascontiguousarray(v[2][:, fluid.begin: fluid.end]))]
    p = torch.zeros(fluid.cell_shape, dtype=torch.float32)
    out, infos = fluid.step(own, p, 0.9, rel_tol=1e-5, max_iterations=300)
    np.savez(os.path.join(out_dir, f"obst{rank}.npz"), b0=fluid.begin, b1=fluid.end, flags=fluid.flags_own.numpy(), p=p.numpy(),
             out0=out[0].numpy(), out1=out[1].numpy(), out2=out[2].numpy(), it=[i.iterations for i in infos], conv=[i.converged for i in infos])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("closed_open", 2), ("periodic", 2), ("closed_open", 3)])
def test_slab_decomposed_step_with_obstacles_gloo(emu_library, emu_ctx, tmp_path, case, world):
    from phiflow_amd import _capi as C
    mp.spawn(_obstacle_worker, args=(world, _free_port(), emu_library.path, str(tmp_path), case), nprocs=world, join=True)
    res, bc, grid, rng = _fluid_problem(case)
    v = _smooth_velocity(emu_ctx, grid, rng, grid.batch)
    singular = all(c != 2 for pair in bc for c in pair)
    # single-process reference: the same C ABI calls on the whole grid
    items = _slab_obstacles(res)
    obs = C.make_obstacles(items)
    g1 = C.make_grid(3, C.PHIHIP_F32, 1, res, (0, 0, 0), tuple(float(r) for r in res), bc)
    acc, flags = np.empty(res, np.uint8), np.empty(res, np.uint8)
    emu_ctx.obstacle_accessible(g1, obs, len(items), acc.ctypes.data)
    emu_ctx.build_cellflags(g1, acc.ctypes.data, 0, 1, flags.ctypes.data)
    P = lambda ts: [t.ctypes.data for t in ts]
    adv = [np.empty_like(t) for t in v]
    emu_ctx.advect_staggered(grid, P(v), P(v), P(adv), 0.9)
    emu_ctx.apply_obstacles(grid, obs, len(items), P(adv))
    div = np.empty((grid.batch,) + tuple(res), np.float32)
    emu_ctx.divergence(grid, P(adv), flags.ctypes.data, 1, singular, div.ctypes.data)
    p = np.zeros_like(div)
    info = emu_ctx.cg_solve(grid, flags.ctypes.data, 1, div.ctypes.data, p.ctypes.data, C.Solve(1e-5, 0.0, 300, 50, 10, 0))
    out = [t.copy() for t in adv]
    emu_ctx.grad_subtract(grid, flags.ctypes.data, 1, p.ctypes.data, P(out))
    parts = [np.load(tmp_path / f"obst{r}.npz") for r in range(world)]
    assert 0 < int((acc == 0).sum()) < acc.size
    assert np.array_equal(np.concatenate([q["flags"] for q in parts], axis=0), flags)          # bit for bit: ghost masks came from their owners
    assert all(bool(c) for q in parts for c in q["conv"]) and all(i.converged for i in info)
    assert all(abs(int(a) - i.iterations) <= 2 for a, i in zip(parts[0]["it"], info))
    cat = lambda key: np.concatenate([q[key] for q in parts], axis=1)
    act = (flags & 64) != 0
    pr, pm = cat("p"), p
    if singular:                                        # the free constant lives on the active cells
        pr = pr - (pr * act).sum(axis=(1, 2, 3), keepdims=True) / act.sum() * act
        pm = pm - (pm * act).sum(axis=(1, 2, 3), keepdims=True) / act.sum() * act
    assert np.abs(pr - pm).max() <= 1e-3 * np.abs(pm).max(), np.abs(pr - pm).max() / np.abs(pm).max()
    for c in range(3):
        assert cat(f"out{c}").shape == out[c].shape
        assert np.abs(cat(f"out{c}") - out[c]).max() <= 3e-4, (c, np.abs(cat(f"out{c}") - out[c]).max())
