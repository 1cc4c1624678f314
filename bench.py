#!/usr/bin/env python3
"""
bench.py -- cell-updates/sec of PhiFlow's incompressible-fluid step on MI355X (BASELINE.json metric).

One "step" = semi-Lagrangian self-advection of the staggered velocity + pressure projection with EXACTLY `--cg-iters`
(100) CG iterations (tolerances 0, true-residual refresh every 50 like PhiML) + gradient subtraction, fp32, on the 3-D
periodic Taylor-Green configuration 256^3 (BASELINE.json configs[1]). Inputs are resident in HBM before the timed region.
N > 1: batch-parallel replicas, one simulation per GPU (weak scaling), one RCCL all-reduce per step of the relative
residual norm; every rank reports its verified iteration count and a bit checksum of its fields (`replicas`: identical
inputs and launch plans => bit-identical results). Prints ONE JSON line on rank 0.

`roofline` names the dominant kernel of the HBM-RESIDENT configuration (512^3 fp32 pressure solve, BASELINE configs[2], run inside this
invocation); the benchmark configuration's own dominant kernel is in `roofline_256`, marked `infinity_cache_assisted` (its three 67 MB
operands fit the 256 MiB Infinity Cache). Fractions count the bytes a kernel MOVES by construction (BASELINE.md "Byte model").

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from phiflow_amd import _capi as C   # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_CELL = {           # algorithmic words per cell (fp32 word = 4 B)
    "cg_matvec_dot": 4 * 4,      # pass A: read r, d; write d, q
    "cg_update": 6 * 4,          # pass B: read x, d, r, q; write x, r
    "cg_iteration": 10 * 4,
    "step_non_cg": 21 * 4,
}
ALG_BYTES_PER_CELL["cg_update_r"] = 6 * 4
MOVED_BYTES_PER_CELL = {         # words the kernels move by construction: q = A d is recomputed instead of stored, and
    "cg_matvec_dot": 3 * 4,      # x is updated every other iteration: UPDATE alternates r-only (3 words) and x + r (5 words)
    "cg_update": 5 * 4,          # UPDATE_X2: read x, r, d; write x, r
    "cg_update_r": 3 * 4,        # UPDATE_R: read r, d; write r
    "cg_iteration": 7 * 4,
}


def taylor_green_velocity(n, device, dtype, batch):
    """ 2-D Taylor-Green vortex of the reference (Taylor_Green.ipynb cell 7: u = cos x sin y, v = -sin x cos y) sampled at
    the staggered face centres and extruded along z, w = 0; domain [0, 2 pi]^3, periodic. """
    h = 2 * math.pi / n
    idx = torch.arange(n, device=device, dtype=torch.float64)
    face, cent = idx * h, (idx + 0.5) * h
    u = (torch.cos(face)[:, None] * torch.sin(cent)[None, :])[:, :, None].expand(n, n, n)
    v = (-torch.sin(cent)[:, None] * torch.cos(face)[None, :])[:, :, None].expand(n, n, n)
    w = torch.zeros(n, n, n, device=device, dtype=torch.float64)
    return [t.to(dtype).unsqueeze(0).repeat(batch, 1, 1, 1).contiguous() for t in (u, v, w)]


class FluidStep:
    """ the benchmarked step, driven through the C ABI with preallocated device buffers """

    def __init__(self, ctx, n, batch, cg_iters, device, dtype=torch.float32, refresh=50):
        self.ctx, self.n, self.batch = ctx, n, batch
        self.device = device
        code = C.PHIHIP_F64 if dtype == torch.float64 else C.PHIHIP_F32
        L = 2 * math.pi
        per = ((C.BC_PERIODIC, C.BC_PERIODIC),) * 3
        self.grid = C.make_grid(3, code, batch, (n, n, n), (0, 0, 0), (L, L, L), per)
        self.v = taylor_green_velocity(n, device, dtype, batch)
        self.v2 = [torch.empty_like(t) for t in self.v]
        self.p = torch.zeros(batch, n, n, n, device=device, dtype=dtype)
        self.div = torch.empty_like(self.p)
        self.res = torch.zeros(batch, 2, device=device, dtype=torch.float64)
        self.rel = torch.zeros(1, device=device, dtype=torch.float64)
        self.solve = C.Solve(0.0, 0.0, cg_iters, refresh, 0, 0)
        self.dt = 0.5 * L / n                         # CFL ~ 0.5
        self.stream = int(torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else 0

    def reset(self):
        """ back to the initial state (N > 1: rank 0 spends one step on the launch-plan autotune before the replicas start together) """
        self.v = taylor_green_velocity(self.n, self.device, self.p.dtype, self.batch)
        self.p.zero_()

    def step(self, allreduce=None):
        ctx, s = self.ctx, self.stream
        pv, pv2 = [t.data_ptr() for t in self.v], [t.data_ptr() for t in self.v2]
        ctx.advect_staggered(self.grid, pv, pv, pv2, self.dt, s)
        ctx.make_incompressible(self.grid, pv2, None, 0, 1, True, self.p.data_ptr(), self.div.data_ptr(), self.solve,
                                want_info=False, stream=s)
        self.v, self.v2 = self.v2, self.v
        ctx.solve_relative_residual(self.batch, self.rel.data_ptr(), s)     # max ||r|| / ||rhs|| over the entries: the all-reduce's operand
        if allreduce is not None:
            allreduce(self.rel)
            return self.rel
        return None


class SmokeBatchStep:
    """ BASELINE.json configs[3] (`--workload config4`): `total` independent 2-D smoke plumes of n^2 cells (closed box 100 x 100, inflow
    sphere at x in linspace(30, 70, total), buoyancy (0, 0.1), dt = 1 -- Batched_Smoke / Smoke_Plume.ipynb cell 5), block-distributed over
    the ranks (phiflow_amd.parallel.local_batch_range): this rank owns entries [b0, b1). One step = mac_cormack(smoke) + inflow,
    semi_lagrangian(v) + buoyancy resample, projection from the previous pressure with exactly `cg_iters` CG iterations. """

    def __init__(self, ctx, n, total, rank, world, cg_iters, device):
        from phiflow_amd.parallel import local_batch_range
        self.ctx, self.n, self.device = ctx, n, device
        self.b0, self.b1 = local_batch_range(total, rank, world)
        B = self.batch = self.b1 - self.b0
        clo = ((C.BC_CLOSED, C.BC_CLOSED),) * 2
        self.grid = C.make_grid(2, C.PHIHIP_F32, max(B, 1), (n, n), (0, 0), (100.0, 100.0), clo)
        h = 100.0 / n
        c = (torch.arange(n, device=device, dtype=torch.float64) + 0.5) * h
        xs = torch.linspace(30.0, 70.0, total, dtype=torch.float64, device=device)[self.b0:self.b1]
        inside = ((c[None, :, None] - xs[:, None, None]) ** 2 + (c[None, None, :] - 9.5) ** 2) <= 25.0
        self.inflow = (0.2 * inside).to(torch.float32).contiguous()
        z = lambda *shape: torch.zeros(max(B, 1), *shape, device=device, dtype=torch.float32)
        self.smoke, self.smoke2 = z(n, n), z(n, n)
        self.v, self.v2 = [z(n - 1, n), z(n, n - 1)], [z(n - 1, n), z(n, n - 1)]
        self.p = z(n, n)
        self.res = torch.zeros(max(B, 1), 2, device=device, dtype=torch.float64)
        self.rel = torch.zeros(1, device=device, dtype=torch.float64)
        self.solve = C.Solve(0.0, 0.0, cg_iters, 50, 0, 0)
        self.s_bc = ((C.BC_OPEN, C.BC_OPEN),) * 2            # ZERO_GRADIENT smoke
        self.stream = int(torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else 0

    def step(self, allreduce=None):
        ctx, s, g = self.ctx, self.stream, self.grid
        rel = self.rel
        rel.zero_()
        if self.batch > 0:
            P = lambda ts: [t.data_ptr() for t in ts]
            ctx.mac_cormack_centered(g, self.smoke.data_ptr(), self.s_bc, None, P(self.v), self.smoke2.data_ptr(), 1.0, 1.0, s)
            self.smoke2 += self.inflow
            ctx.advect_staggered(g, P(self.v), P(self.v), P(self.v2), 1.0, s)
            ctx.centered_to_staggered(g, self.smoke2.data_ptr(), self.s_bc, None, (0.0, 0.1), True, P(self.v2), s)
            ctx.make_incompressible(g, P(self.v2), None, 0, 1, True, self.p.data_ptr(), 0, self.solve, want_info=False, stream=s)
            ctx.solve_relative_residual(self.batch, rel.data_ptr(), s)
            self.smoke, self.smoke2 = self.smoke2, self.smoke
            self.v, self.v2 = self.v2, self.v
        if allreduce is not None:
            allreduce(rel)
        return rel


class Smoke3DStep:
    """ `--workload smoke256`: ONE 3-D smoke plume of n^3 cells fp32 in a closed box (the 3-D form of Smoke_Plume.ipynb cell 5 with the
    diffusion of Taylor_Green.ipynb cell 12) -- the step in which the NON-CG kernels matter:
        smoke = mac_cormack(smoke, v, dt) + inflow                   (advect.py:182-215)
        v     = semi_lagrangian(v, v, dt) + resample(smoke * (0, 0, 0.1), to=v)   (advect.py:156-179; _resample.py:156-157,272-276)
        v     = diffuse.explicit(v, 0.01, dt)                        (diffuse.py:13-60)
        v, p  = make_incompressible(v, (), Solve('CG', x0=p))        with exactly `cg_iters` (default 20) iterations from the previous pressure
    `op_ms` times every operation of one extra step with an event pair around the C call (several launches each). """

    def __init__(self, ctx, n, cg_iters, device):
        self.ctx, self.n, self.device = ctx, n, device
        clo = ((C.BC_CLOSED, C.BC_CLOSED),) * 3
        self.grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (100.0, 100.0, 100.0), clo)
        h = 100.0 / n
        c = (torch.arange(n, device=device, dtype=torch.float64) + 0.5) * h
        inside = ((c[:, None, None] - 50.0) ** 2 + (c[None, :, None] - 50.0) ** 2 + (c[None, None, :] - 9.5) ** 2) <= 25.0
        self.inflow = (0.2 * inside).to(torch.float32).unsqueeze(0).contiguous()
        z = lambda *shape: torch.zeros(1, *shape, device=device, dtype=torch.float32)
        self.smoke, self.smoke2 = z(n, n, n), z(n, n, n)
        shapes = [tuple(ctx.component_shape(self.grid, d)) for d in range(3)]
        self.v, self.v2 = [z(*sh) for sh in shapes], [z(*sh) for sh in shapes]
        self.p = z(n, n, n)
        self.rel = torch.zeros(1, device=device, dtype=torch.float64)
        self.solve = C.Solve(0.0, 0.0, cg_iters, 50, 0, 0)
        self.s_bc = ((C.BC_OPEN, C.BC_OPEN),) * 3            # ZERO_GRADIENT smoke
        self.dt = 1.0
        self.kdt = 0.01 * self.dt
        self.stream = int(torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else 0

    def ops(self):
        """ the step as (name, callable) pairs; buffers rotate v -> v2 -> v -> v2 (advect out of place, diffuse out of place) """
        ctx, s, g = self.ctx, self.stream, self.grid
        P = lambda ts: [t.data_ptr() for t in ts]

        def inflow():
            self.smoke2 += self.inflow
        return [
            ("mac_cormack_smoke", lambda: ctx.mac_cormack_centered(g, self.smoke.data_ptr(), self.s_bc, None, P(self.v), self.smoke2.data_ptr(), self.dt, 1.0, s)),
            ("inflow_add", inflow),
            ("semi_lagrangian_v", lambda: ctx.advect_staggered(g, P(self.v), P(self.v), P(self.v2), self.dt, s)),
            ("buoyancy_resample", lambda: ctx.centered_to_staggered(g, self.smoke2.data_ptr(), self.s_bc, None, (0.0, 0.0, 0.1), True, P(self.v2), s)),
            ("diffuse_explicit", lambda: ctx.diffuse_explicit(g, P(self.v2), P(self.v), self.kdt, s)),
            ("make_incompressible", lambda: ctx.make_incompressible(g, P(self.v), None, 0, 1, True, self.p.data_ptr(), 0, self.solve, want_info=False, stream=s)),
            ("residual_export", lambda: ctx.solve_relative_residual(1, self.rel.data_ptr(), s)),
        ]

    def step(self, allreduce=None):
        for _, fn in self.ops():
            fn()
        self.smoke, self.smoke2 = self.smoke2, self.smoke       # (v ends in self.v again: advect v -> v2, diffuse v2 -> v, projection in place)
        if allreduce is not None:
            allreduce(self.rel)
        return self.rel


_RECORD_FD = None


# ---- what a launch needs of the machine. bench.py has NO CPU path: these raise / use RCCL as written. tests/bench_dryrun.py replaces the four names
# below to rehearse the driver's `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` command line on a host without GPUs (gloo, the kernel
# sources under the fiber emulation): bookkeeping only, the record of such a run says so and is never a measurement. ----
DIST_BACKEND = "nccl"          # = RCCL on ROCm


def device_for_rank(local_rank: int) -> torch.device:
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    return device


def load_library():
    lib = C.load_default_library()
    # (PHIHIP_LIBRARY = same-box A/B against another BUILD of the library, e.g. the previous round's: the record carries that build's id)
    assert lib.built_from_tree() or os.environ.get("PHIHIP_LIBRARY"), f"stale libphihip.so ({lib.build_id()}) -- sources are src:{C.source_hash()}; run __graft_entry__.build()"
    return lib


def device_sync(device):
    torch.cuda.synchronize(device)


def emit_record(record: dict):
    """ the one JSON line of the contract, on the process's ORIGINAL stdout (see main) """
    line = (json.dumps(record) + "\n").encode()
    if _RECORD_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RECORD_FD, line)


def bench_config4(args, ctx, device, rank, world, dist, barrier, allreduce):
    """ strong scaling of the batched smoke workload: the SAME 8 simulations on 1 / 2 / 4 / 8 GPUs """
    total, n = args.batch_total, args.size if args.size != 256 else 512
    sim = SmokeBatchStep(ctx, n, total, rank, world, args.cg_iters, device)
    for _ in range(args.warmup):
        sim.step(allreduce)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rel = sim.step(allreduce)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- self-validation on EVERY rank (the first multi-GPU run of this workload will be the driver's): one reporting step on the owned
    # entries, then batch entry 0 recomputed alone on this rank with the analytic launch plans -- its bits must agree across ranks ----
    its_local, ok_local = [], True
    if sim.batch > 0:
        P = lambda ts: [t.data_ptr() for t in ts]
        g, s_ = sim.grid, sim.stream
        ctx.mac_cormack_centered(g, sim.smoke.data_ptr(), sim.s_bc, None, P(sim.v), sim.smoke2.data_ptr(), 1.0, 1.0, s_)
        sim.smoke2 += sim.inflow
        ctx.advect_staggered(g, P(sim.v), P(sim.v), P(sim.v2), 1.0, s_)
        ctx.centered_to_staggered(g, sim.smoke2.data_ptr(), sim.s_bc, None, (0.0, 0.1), True, P(sim.v2), s_)
        info = ctx.make_incompressible(g, P(sim.v2), None, 0, 1, True, sim.p.data_ptr(), 0, sim.solve, want_info=True, stream=s_)
        sim.smoke, sim.smoke2 = sim.smoke2, sim.smoke
        sim.v, sim.v2 = sim.v2, sim.v
        its_local = [int(i.iterations) for i in info]
        ok_local = all(i.iterations == args.cg_iters and not i.diverged for i in info)
    ctx.set_autotune(False)
    ref = SmokeBatchStep(ctx, n, total, 0, total, args.cg_iters, device)          # rank 0 of a world of `total`: entry 0 alone
    for _ in range(args.warmup + args.steps + 1):
        ref.step(None)
    device_sync(device)
    ctx.set_autotune(True)
    shards = gather_shards(dist, world, total, its_local, ok_local, [sim.p, sim.smoke] + sim.v, [ref.p, ref.smoke] + ref.v, device)
    assert all(shards["verified_ok"]), shards["iterations_per_rank"]
    if rank == 0:
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        sim.step(None)
        device_sync(device)
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        cells = total * n * n
        it_us = (prof["cg_matvec_dot"][1] + prof["cg_update"][1] + prof["cg_update_r"][1]) / max(1, prof["cg_matvec_dot"][0]) * 1e3
        if prof["cg_matvec_dot"][0] == 0:          # the resident solver: one launch for the whole solve
            it_us = prof["cg_update"][1] / max(1, args.cg_iters) * 1e3
        emit_record({
            "metric": f"cell-updates/sec (mac_cormack + advect + {args.cg_iters} CG iters), {total} x {n}^2 fp32 batched smoke", "value": cells * args.steps / elapsed,
            "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batched 2D smoke {total} x {n}^2 (BASELINE.json configs[3]), batch block-distributed over {world} GPU(s), "
                                   f"{args.cg_iters} CG iterations per projection", "sims_total": total, "sims_rank0": sim.batch, "cg_iterations": args.cg_iters,
                       "parallelism": f"batch-parallel x{world}, no data-path collective, 1 all-reduce(max residual)/step"},
            "final_relative_residual": float(rel.item()), "us_per_cg_iteration_rank0": round(it_us, 3), "resident_cg": int(args.resident_cg),
            "iterations_verified": shards["iterations_per_rank"], "shards": shards, "build_id": ctx.lib.build_id(),
            "scaling_measured": "one point of a strong-scaling curve; no multi-GPU curve has been measured by the builder (single-GPU boxes only)",
            "kernel_ms_per_step_rank0": {k: round(v[1], 5) for k, v in prof.items()},
            "plan": {name: ctx.query_plan(sim.grid, False, fam) for name, fam in (("matvec", 1), ("update_x2", 2), ("update_r", 3))}})
    if dist is not None:
        dist.destroy_process_group()


def bench_smoke3d(args, ctx, lib, device, rank, world, dist, barrier, allreduce):
    """ `--workload smoke256`: replicas of the 3-D smoke step (weak scaling like the headline workload); the record carries the share of every
    operation of the step -- with 20 warm-started CG iterations the advection / resample / diffusion kernels are ~half of it. """
    n = args.size
    iters = args.cg_iters if args.cg_iters != 100 else 20
    sim = Smoke3DStep(ctx, n, iters, device)
    for _ in range(max(args.warmup, 30)):          # the plume has to exist before the step is representative (smoke rises ~1 cell per step at first)
        sim.step(allreduce)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rel = sim.step(allreduce)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        # one more step with an event pair around every operation (C call) + the library's per-launch events for the projection's parts
        evs = []
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        e_begin = torch.cuda.Event(enable_timing=True)
        e_begin.record()
        for name, fn in sim.ops():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            evs.append((name, a, b))
        e_end = torch.cuda.Event(enable_timing=True)
        e_end.record()
        torch.cuda.synchronize(device)
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        sim.smoke, sim.smoke2 = sim.smoke2, sim.smoke
        op_ms = {name: round(a.elapsed_time(b), 5) for name, a, b in evs}
        wall = e_begin.elapsed_time(e_end)
        fb = list(ctx.advect_fallback_stats())
        cells = n ** 3
        ms_step = elapsed / args.steps * 1e3
        non_cg = sum(v for k, v in op_ms.items() if k != "make_incompressible") + prof["divergence"][1] + prof["grad_subtract"][1] + prof["cg_residual"][1]
        emit_record({
            "metric": f"cell-updates/sec (mac_cormack smoke + advect + buoyancy + diffuse + {iters} CG iters), {n}^3 fp32 closed box", "value": cells * world * args.steps / elapsed,
            "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 30), "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"3D smoke plume {n}^3 fp32, closed box: mac_cormack(smoke) + inflow, semi_lagrangian(v), buoyancy resample, diffuse.explicit, "
                                   f"projection with {iters} CG iterations from the previous pressure (Smoke_Plume.ipynb cell 5 in 3-D; NOT a BASELINE.json config: "
                                   f"the step in which the non-CG kernels count)", "cells_per_gpu": cells, "cg_iterations": iters,
                       "parallelism": f"batch-parallel replicas x{world}, 1 all-reduce(max residual)/step"},
            "final_relative_residual": float(rel.item()),
            "op_ms_profiled_step": op_ms, "profiled_step_wall_ms": round(wall, 5),
            "projection_parts_ms": {k: round(v[1], 5) for k, v in prof.items() if v[0] and k not in ("advect", "other")},
            "non_cg_share_of_profiled_step": round(non_cg / wall, 4),
            "advect_fallback_last_call": fb, "build_id": lib.build_id(),
            "note": "op_ms: event pairs around each C call of ONE extra step (the per-launch events of the profiling mode add ~2 us per launch: the "
                    "profiled step is slower than ms_per_step); smoke_max %.4f" % float(sim.smoke.max().item())})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_slab(args, lib, device, rank, world, dist, barrier):
    """ `--workload slab` (no reference counterpart): ONE n^3 fp32 periodic Taylor-Green simulation decomposed into x-slabs, one per
    GPU; step = ghost-plane exchange + advection + divergence + slab CG with exactly --cg-iters iterations (halo planes + 2 scalar
    all-reduces per iteration) + gradient subtraction. Fixed total size => strong scaling. """
    from phiflow_amd.backend import HipBackend
    from phiflow_amd.slab import SlabFluid
    be = HipBackend(library=lib, device=str(device))
    n = args.size if args.size != 256 else 512
    L = 2 * math.pi
    per = ((C.BC_PERIODIC, C.BC_PERIODIC),) * 3
    fluid = SlabFluid(be, (n, n, n), (0.0, 0.0, 0.0), (L, L, L), per, torch.float32, batch=1, ghost=2, overlap=bool(args.overlap))
    h = L / n
    idx = torch.arange(n, device=device, dtype=torch.float64)
    face, cent = idx * h, (idx + 0.5) * h
    xs_f, xs_c = face[fluid.face_begin:fluid.face_end], cent[fluid.begin:fluid.end]
    u = (torch.cos(xs_f)[:, None] * torch.sin(cent)[None, :])[:, :, None].expand(len(xs_f), n, n)
    v_ = (-torch.sin(xs_c)[:, None] * torch.cos(face)[None, :])[:, :, None].expand(len(xs_c), n, n)
    vel = [u.to(torch.float32).unsqueeze(0).contiguous(), v_.to(torch.float32).unsqueeze(0).contiguous(),
           torch.zeros(1, len(xs_c), n, n, device=device)]
    p = torch.zeros(fluid.cell_shape, device=device)
    dt = 0.5 * h

    def step(v):
        return fluid.step(v, p, dt, rel_tol=0.0, abs_tol=0.0, max_iterations=args.cg_iters, refresh_every=50, check_every=0)
    for _ in range(args.warmup):
        vel, infos = step(vel)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vel, infos = step(vel)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    its = torch.tensor([infos[0].iterations, int(infos[0].diverged)], device=device, dtype=torch.int64)
    every = [its.clone() for _ in range(world)]
    if dist is not None:
        dist.all_gather(every, its)
    if rank == 0:
        cells = n ** 3
        emit_record({
            "metric": f"cell-updates/sec (advect + {args.cg_iters} CG iters), ONE {n}^3 fp32 simulation on x-slabs", "value": cells * args.steps / elapsed,
            "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"3D periodic Taylor-Green {n}^3 fp32 as ONE simulation on {world} x-slab(s) (SURVEY §8 f4), {args.cg_iters} CG iterations/step",
                       "planes_rank0": fluid.end - fluid.begin, "ghost_planes": fluid.ghost, "cg_iterations": args.cg_iters,
                       "advect_exchange_overlap": bool(fluid.overlap and world > 1 and fluid._overlap_windows() is not None),
                       "parallelism": f"slab decomposition x{world}: per CG iteration 2 boundary-plane exchanges (point-to-point) + 2 all-reduces of 8 B"},
            "iterations_verified": [[int(x) for x in e.tolist()] for e in every], "final_relative_residual": math.sqrt(infos[0].residual_sq / infos[0].rhs_sq),
            "build_id": lib.build_id(),
            "scaling_measured": "one point of a strong-scaling curve; no multi-GPU curve has been measured by the builder (single-GPU boxes only)"})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(n, cg_iters):
    """ the NumPy oracle (restatement of the reference's CPU path) timed on the same step at a bounded size; the oracle's fields are
    kept for the parity block (the GPU repeats exactly this step on the same inputs) """
    from oracle import phi_oracle as O
    L = 2 * math.pi
    dom = O.Domain((n, n, n), (0, 0, 0), (L, L, L), ((O.PERIODIC, O.PERIODIC),) * 3)
    h = L / n
    idx = np.arange(n)
    face, cent = idx * h, (idx + 0.5) * h
    u = np.broadcast_to((np.cos(face)[:, None] * np.sin(cent)[None, :])[:, :, None], (n, n, n))
    v = np.broadcast_to((-np.sin(cent)[:, None] * np.cos(face)[None, :])[:, :, None], (n, n, n))
    vel = [np.ascontiguousarray(a, dtype=np.float32)[None] for a in (u, v, np.zeros((n, n, n)))]
    t0 = time.perf_counter()
    vel = O.semi_lagrangian_staggered(vel, vel, 0.5 * h, dom)
    vel, p, info, _ = O.make_incompressible(vel, dom, rtol=0.0, atol=0.0, max_iter=cg_iters, refresh=50)
    dt = time.perf_counter() - t0
    cpu = {"value": n ** 3 / dt, "unit": "cell-updates/s", "cores": 1, "kind": "port",
           "sample": f"1 step of the same workload at {n}^3 fp32 ({cg_iters} CG iterations) with the NumPy oracle "
                     f"(oracle/phi_oracle.py, single-threaded NumPy), {dt:.1f} s"}
    return cpu, dict(v=vel, p=p, rel_residual_sq=float(info.residual_sq[0] / info.rhs_sq[0]))


def parity_block(ctx, n, cg_iters, device, ref):
    """ the GPU on the SAME inputs as the CPU leg (one benchmark step at n^3), compared with the oracle's fields: the north-star's
    'pressure within 1e-4 rel-L2 of NumPy' measured inside the benchmark run """
    sim = FluidStep(ctx, n, 1, cg_iters, device)
    pv, pv2 = [t.data_ptr() for t in sim.v], [t.data_ptr() for t in sim.v2]
    ctx.advect_staggered(sim.grid, pv, pv, pv2, sim.dt, sim.stream)
    info = ctx.make_incompressible(sim.grid, pv2, None, 0, 1, True, sim.p.data_ptr(), 0, sim.solve, want_info=True, stream=sim.stream)
    torch.cuda.synchronize(device)
    pg = sim.p.cpu().numpy().astype(np.float64)
    po = ref["p"].astype(np.float64)
    pg -= pg.mean(); po -= po.mean()
    p_err = float(np.linalg.norm(pg - po) / np.linalg.norm(po))
    v_err = max(float(np.abs(a.cpu().numpy() - b).max()) for a, b in zip(sim.v2, ref["v"]))
    return {"size": n, "cg_iterations": int(info[0].iterations), "pressure_rel_l2": p_err, "velocity_max_abs": v_err,
            "rel_residual_sq": info[0].residual_sq / info[0].rhs_sq, "rel_residual_sq_oracle": ref["rel_residual_sq"],
            "tolerance": {"pressure_rel_l2": 1e-4, "velocity_max_abs": 2e-5}, "ok": bool(p_err <= 1e-4 and v_err <= 2e-5),
            "reference": "NumPy oracle (oracle/phi_oracle.py) on identical inputs: the cpu_baseline sample"}


def config3_block(ctx, device, n=512, iters=100):
    """ BASELINE configs[2] (the north-star's roofline target): 512^3 fp32 periodic pressure solve, `iters` fixed CG iterations on a
    seeded mean-free random rhs; wall time per iteration + per-kernel hipEvent times, bytes MOVED by construction vs 8 TB/s """
    L = 2 * math.pi
    per = ((C.BC_PERIODIC, C.BC_PERIODIC),) * 3
    grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), per)
    g = torch.Generator(device=device).manual_seed(0)
    rhs = torch.randn(1, n, n, n, generator=g, device=device)
    rhs -= rhs.mean()
    x = torch.zeros_like(rhs)
    solve = C.Solve(0.0, 0.0, iters, 50, 0, 0)
    stream = int(torch.cuda.current_stream(device).cuda_stream)
    ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 4, 0, 0, 0), want_info=False, stream=stream)   # warm-up (workspace, plans)
    x.zero_()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False, stream=stream)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    x.zero_()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    info = ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=True, stream=stream)
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    assert info[0].iterations == iters and not info[0].diverged, (info[0].iterations, info[0].diverged)
    cells = n ** 3
    per_k = {k: (v[1] / v[0] if v[0] else None) for k, v in prof.items()}
    ms_it = wall / iters * 1e3
    moved = MOVED_BYTES_PER_CELL["cg_iteration"] * cells
    out = {"size": n, "launches": {k: v[0] for k, v in prof.items() if v[0]},
           "workload": f"{n}^3 fp32 periodic pressure solve, {iters} CG iterations, seeded random rhs (BASELINE.json configs[2])",
           "ms_per_solve": round(wall * 1e3, 3), "ms_per_iteration": round(ms_it, 5),
           "moved_bytes_per_iteration": moved, "moved_GBs": round(moved / (ms_it * 1e-3) / 1e9, 1),
           "moved_frac": round(moved / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "algorithmic_equiv_frac": round(ALG_BYTES_PER_CELL["cg_iteration"] * cells / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "kernel_ms_per_launch": {k: (round(v, 5) if v else None) for k, v in per_k.items() if v},
           "plan": {name: ctx.query_plan(grid, False, fam) for name, fam in (("matvec", 1), ("update_x2", 2), ("update_r", 3))},
           "plan_all": {str(fam): ctx.query_plan(grid, False, fam) for fam in (0, 1, 2, 3)},
           "final_relative_residual": math.sqrt(info[0].residual_sq / info[0].rhs_sq),
           # r6: the first solve on the freshly grown workspace chose between candidate (r, d0, d1) allocations (include/phihip.h phihip_workspace_placement): how many,
           # and the iteration loop's microseconds on the first allocation and on the one kept
           "workspace_placement": ctx.workspace_placement() if hasattr(ctx, "workspace_placement") else None,
           "note": "wall time of the whole solve (initial residual, 100 x (MATVEC + UPDATE), refresh at 50, final state) / iterations; "
                   "moved = 7 words per cell and iteration by construction (3 MATVEC + mean of 3 / 5 UPDATE_R / UPDATE_X2)"}
    del rhs, x
    return out


def phi_level_block(ctx, lib, device, cg_iters, steps=20, sizes=(256, 128), plume_n=128, plume_steps=100):
    """ what a PhiFlow user calls -- `advect.semi_lagrangian(Field, Field, dt)` (phi/physics/advect.py:156) and
    `fluid.make_incompressible(Field, (), Solve)` (phi/physics/fluid.py:94) through `phiflow_amd.flow` -- timed NEXT TO the raw C-ABI loop of the
    same step on the same box: the Taylor-Green step at 256^3 and 128^3 and the BASELINE configs[0] smoke-plume step at 128^2 (Smoke_Plume.ipynb
    cell 5). `overhead` = phi-level wall / C-ABI wall - 1: Python objects, ctypes marshalling, the functional API's copies (a Field is immutable:
    the projection works on a clone) and the per-step host synchronisation that raising NotConverged / Diverged requires. """
    from phiflow_amd import flow as F
    from phiflow_amd.backend import HipBackend
    be = HipBackend(library=lib, device=str(device))
    be.ctx = ctx                                    # one context: the launch plans tuned by the C-ABI loop serve both
    out = {}

    def wall(fn, n_steps):
        for _ in range(3):
            fn()
        device_sync(device)
        # r6: CPython's generation-2 garbage collection walks torch's whole object graph (33 ms on the GPU box, profiles/r06_backward_step.txt); in a 20-100-step
        # window it either falls or not and moves an eager small-grid figure by up to 0.3 ms per step. Collect now and freeze what is alive: the loop's own garbage is
        # still collected (generations 0 / 1), the long-lived graph is not walked again. A user's loop gains the same from `gc.freeze()` after set-up.
        import gc
        gc.collect()
        gc.freeze()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            fn()
        device_sync(device)
        return (time.perf_counter() - t0) / n_steps * 1e3

    with be:
        for n in sizes:
            sim = FluidStep(ctx, n, 1, cg_iters, device)
            ms_c = wall(lambda: sim.step(None), steps)
            L = 2 * math.pi
            state = {"v": F.StaggeredGrid([t.clone() for t in taylor_green_velocity(n, device, torch.float32, 1)], F.PERIODIC, F.Box(x=L, y=L, z=L), x=n, y=n, z=n), "p": None}
            solve = lambda x0: F.Solve('CG', 0, 0, x0=x0, max_iterations=cg_iters, suppress=[F.NotConverged])

            def phi_step():
                v = F.advect.semi_lagrangian(state["v"], state["v"], sim.dt)
                state["v"], state["p"] = F.fluid.make_incompressible(v, (), solve(state["p"]))
            ms_phi = wall(phi_step, steps)
            its = list(state["p"].solve_info.iterations)
            assert its == [cg_iters], its
            # the same step as a captured hipGraph (phiflow_amd.jit.jit_compile, the role of math.jit_compile in the reference's examples)
            jstep = F.jit_compile(lambda v, p: F.fluid.make_incompressible(F.advect.semi_lagrangian(v, v, sim.dt), (), solve(p)))

            def jit_step():
                state["v"], state["p"] = jstep(state["v"], state["p"])
            ms_jit = wall(jit_step, steps) if device.type == "cuda" else None
            # r6: the same captured step as a ping-pong of two captures (copy_outputs=False: a result is consumed by the next call, the loop every example writes):
            # no 268-MB clone of the results, the inputs copied in every other step only
            pstep = F.jit_compile(lambda v, p: F.fluid.make_incompressible(F.advect.semi_lagrangian(v, v, sim.dt), (), solve(p)), copy_outputs=False)

            def pp_step():
                state["v"], state["p"] = pstep(state["v"], state["p"])
            ms_pp = wall(pp_step, steps) if device.type == "cuda" else None
            out[f"taylor_green_{n}"] = {"ms_c_abi": round(ms_c, 4), "ms_phi_level": round(ms_phi, 4), "overhead": round(ms_phi / ms_c - 1, 4),
                                        "ms_phi_level_jit": round(ms_jit, 4) if ms_jit else None, "overhead_jit": round(ms_jit / ms_c - 1, 4) if ms_jit else None,
                                        "ms_phi_level_jit_ping_pong": round(ms_pp, 4) if ms_pp else None,
                                        "overhead_jit_ping_pong": round(ms_pp / ms_c - 1, 4) if ms_pp else None}
            del jstep, pstep
            del sim, state
            if device.type == "cuda":
                torch.cuda.empty_cache()
        # BASELINE configs[0]: the 128^2 smoke plume (closed box, MacCormack smoke + inflow, buoyancy, projection from the previous pressure)
        n, iters = plume_n, 50
        sim = SmokeBatchStep(ctx, n, 1, 0, 1, iters, device)
        ms_c = wall(lambda: sim.step(None), plume_steps)
        dom = F.Box(x=100, y=100)
        inflow = 0.2 * F.resample(F.Sphere(x=50, y=9.5, radius=5), to=F.CenteredGrid(0, F.ZERO_GRADIENT, dom, x=n, y=n), soft=True)
        st = {"v": F.StaggeredGrid(0, 0, dom, x=n, y=n), "s": F.CenteredGrid(0, F.ZERO_GRADIENT, dom, x=n, y=n), "p": None}

        def plume_step():
            s_ = F.advect.mac_cormack(st["s"], st["v"], 1.0) + inflow
            v = F.advect.semi_lagrangian(st["v"], st["v"], 1.0) + F.resample(s_ * (0, 0.1), to=st["v"])
            st["v"], st["p"] = F.fluid.make_incompressible(v, (), F.Solve('CG', 0, 0, x0=st["p"], max_iterations=iters, suppress=[F.NotConverged]))
            st["s"] = s_
        ms_phi = wall(plume_step, plume_steps)

        @F.jit_compile
        def plume_jit(v, s, p):
            s_ = F.advect.mac_cormack(s, v, 1.0) + inflow
            v = F.advect.semi_lagrangian(v, v, 1.0) + F.resample(s_ * (0, 0.1), to=v)
            v, p = F.fluid.make_incompressible(v, (), F.Solve('CG', 0, 0, x0=p, max_iterations=iters, suppress=[F.NotConverged]))
            return v, s_, p

        def plume_jit_step():
            st["v"], st["s"], st["p"] = plume_jit(st["v"], st["s"], st["p"])
        ms_jit = wall(plume_jit_step, plume_steps) if device.type == "cuda" else None
        out[f"smoke_plume_{n}x{n}"] = {"ms_c_abi": round(ms_c, 4), "ms_phi_level": round(ms_phi, 4), "overhead": round(ms_phi / ms_c - 1, 4), "cg_iterations": iters,
                                       "ms_phi_level_jit": round(ms_jit, 4) if ms_jit else None, "overhead_jit": round(ms_jit / ms_c - 1, 4) if ms_jit else None}
    out["note"] = ("same box, same context, back to back; C-ABI = preallocated buffers driven like the timed region of this line; phi-level = phiflow_amd.flow "
                   "(immutable Fields, a fresh result per operator, SolveInfo read back every step); phi-level jit = the same function behind "
                   "phiflow_amd.flow.jit_compile: captured once in a hipGraph, replayed per step (inputs copied in, results cloned out, no read-back); jit ping-pong = "
                   "jit_compile(copy_outputs=False): two captures alternate, the second reads the first one's outputs in place (no clones, inputs copied every other step)")
    return out


def live_pmc_traffic(n, kernel_key, plans=None):
    """ HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes run NOW (separate FETCH_SIZE / WRITE_SIZE passes of
    tools/pmc_workload.py at this size, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x 2048 B -- gfx950 counts half of a
    wide streaming read --, WRITE_SIZE x 1024 B). Returns (bytes or None, source string). """
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or any(k.startswith("ROCPROF") for k in os.environ) or os.environ.get("PHIHIP_BENCH_PMC", "1") == "0":
        return None, None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_summary
        base = tempfile.mkdtemp(prefix="phihip_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        if plans:      # the traced child runs exactly the launch plans of this invocation's solve (no autotune candidates under the kernel's name)
            env["PHIHIP_PMC_PLANS"] = json.dumps({str(n): plans})
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(base, ctr), "-o", "pmc", "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py"), str(n)]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                return None, None
            if proc.returncode != 0:
                return None, None
        res = pmc_summary.summarise(base)
        kk = {"cg_update": "cg_update_x2"}.get(kernel_key, kernel_key)      # (pmc_summary names mode 7 cg_update_x2)
        cands = [(e.get("launches", 0), key, e) for key, e in res["kernels"].items()
                 if key.startswith(kk + "<") and "read_bytes_prescribed" in e and "write_bytes_prescribed" in e]
        for _, key, e in sorted(cands, key=lambda c: -c[0])[:1]:      # the plan the solve ran: the variant with the most launches
            if True:
                src = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run inside this bench invocation on tools/pmc_workload.py {n} "
                       f"({key}; {'launch plans of this invocation pinned, autotune off' if plans else 'self-tuned child'}; calibration copy: fetch unit "
                       f"{res['units']['fetch_bytes_per_unit_calibrated']}, write unit {res['units']['write_bytes_per_unit_calibrated']} B)")
                return int(round(e["read_bytes_prescribed"] + e["write_bytes_prescribed"])), src
    except Exception:
        pass
    return None, None


def cpu_cg_variants(n, iters, dtype=np.float32):
    """ part of the cpu_baseline leg: the two host forms of the CG -- the matrix-free NumPy restatement and the assembled
    SciPy-CSR operator, which is what PhiML's NumPy backend iterates on after tracing `masked_laplace` (phi/physics/fluid.py:165) """
    from oracle import phi_oracle as O
    L = 2 * np.pi
    dom = O.Domain((n, n, n), (0, 0, 0), (L, L, L), ((O.PERIODIC, O.PERIODIC),) * 3)
    rng = np.random.default_rng(0)
    rhs = rng.standard_normal((1, n, n, n)).astype(dtype)
    rhs -= rhs.mean()
    t0 = time.perf_counter()
    x_mf, _ = O.cg(lambda p: O.masked_laplace(p, dom, None, None), rhs, np.zeros_like(rhs), 0.0, 0.0, iters, 50)
    t_mf = time.perf_counter() - t0
    t0 = time.perf_counter()
    A = O.laplace_csr(dom, dtype)
    t_asm = time.perf_counter() - t0
    t0 = time.perf_counter()
    x_sp, _ = O.cg(lambda p: (A @ p.reshape(-1)).reshape(p.shape), rhs, np.zeros_like(rhs), 0.0, 0.0, iters, 50)
    t_sp = time.perf_counter() - t0
    rel = float(np.linalg.norm(x_mf - x_sp) / np.linalg.norm(x_mf))
    cells = n ** 3
    return {"size": n, "iterations": iters, "matrix_free_numpy_Mcell_it_per_s": round(cells * iters / t_mf / 1e6, 1),
            "scipy_csr_Mcell_it_per_s": round(cells * iters / t_sp / 1e6, 1), "csr_assembly_s": round(t_asm, 2),
            "solutions_rel_l2": rel, "cores": 1}



CG_KERNELS = {"cg_matvec_dot": "march_kernel<MODE_MATVEC> (d = r + beta d, sum d.Ad; one launch per CG iteration)",
              "cg_update": "march_kernel<MODE_UPDATE_X2> (x += two steps, r -= alpha A d, sum r^2; every other iteration)",
              "cg_update_r": "march_kernel<MODE_UPDATE_R> (r -= alpha A d, sum r^2; every other iteration)"}


def committed_traffic_ratio(n, kernel_name):
    """ PMC bytes / moved bytes of the same kernel in the committed per-kernel table (tools/kernel_roofline.sh -> profiles/r06_kernel_roofline.json (or the r05 table),
    pinned plans, tools/path_workload.py): the second, independent measurement the bench line's traffic is checked against """
    for name in ("r06_kernel_roofline.json", "r05_kernel_roofline.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                table = json.load(f)
            for grp in table["groups"]:
                if grp.get("size") != n or not str(grp.get("group", "")).startswith("f32"):
                    continue
                for k in grp["kernels"]:
                    if kernel_name in k.get("label", "") and k.get("pmc_over_moved"):
                        return float(k["pmc_over_moved"]), f"profiles/{name} group {grp['group']}", grp.get("pinned_plans")
        except Exception:
            continue
    return None, None, None


def roofline_block(n, per, pmc, world, cache_assisted, where, plans=None):
    """ per: {kernel family: (avg ms per launch, launches, total ms)} from hipEvent pairs on the solve stream. Dominant kernel = largest share
    of the GPU time among the three CG kernels; achieved = bytes it MOVES by construction / its average launch time. """
    cells = n ** 3
    dom_key = max(CG_KERNELS, key=lambda k: per[k][2])
    t_dom = per[dom_key][0]
    if not t_dom:
        return None, {}
    total_ms = sum(v[2] for v in per.values())
    moved = MOVED_BYTES_PER_CELL[dom_key] * cells
    alg = ALG_BYTES_PER_CELL[dom_key] * cells
    gbs = moved / (t_dom * 1e-3) / 1e9
    traffic, source = (None, None)
    if pmc and world == 1:
        traffic, source = live_pmc_traffic(n, dom_key, plans)
    # (r6: no fallback to a committed PMC file -- until r5 `roofline_256.traffic` came from profiles/pmc_traffic.json, a ROUND-2 measurement of other kernels on
    # another launch plan; a figure that cannot be measured in this invocation is null)
    block = {"bound": "hbm", "kernel": CG_KERNELS[dom_key], "size": n, "measured_on": where, "infinity_cache_assisted": bool(cache_assisted),
             "share_of_gpu_time": round(per[dom_key][2] / total_ms, 3), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source,
             "traffic_frac": round(traffic / (t_dom * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
             "traffic_over_moved": round(traffic / moved, 3) if traffic else None,
             "avg_launch_ms": round(t_dom, 5), "launches": per[dom_key][1], "bytes_per_launch": moved,
             "bytes_basis": f"bytes the kernel moves by construction (BASELINE.md 'Byte model', DESIGN.md 3.1): {MOVED_BYTES_PER_CELL[dom_key] // 4} words x 4 B x "
                            f"{cells} cells; `traffic` (rocprofv3 PMC) cross-checks it",
             "algorithmic_bytes_per_launch": alg, "algorithmic_equiv_frac": round(alg / (t_dom * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "achievable_copy_GBs": 6290.0}
    if traffic and plans:
        label = {"cg_matvec_dot": "CG MATVEC", "cg_update": "CG UPDATE_X2", "cg_update_r": "CG UPDATE_R"}[dom_key]
        ref_ratio, ref_src, ref_plans = committed_traffic_ratio(n, label)
        if ref_ratio:
            # the halo columns / planes a tile re-reads belong to the launch plan: the two figures are the same measurement only where the first-call
            # autotune of this invocation chose the plan the table was traced with
            fam = {"cg_matvec_dot": "1", "cg_update": "2", "cg_update_r": "3"}[dom_key]
            here = [int(v) for v in plans.get(fam, [])][:3]
            ref = ref_plans.get(fam) if isinstance(ref_plans, dict) else None
            ref = [int(ref[k]) for k in ("rows", "tpr", "chunk")] if ref else None
            same = bool(here) and here == ref
            block["traffic_cross_check"] = {"pmc_over_moved_here": block["traffic_over_moved"], "pmc_over_moved_committed_table": ref_ratio, "table": ref_src,
                                            "plan_here": here, "plan_table": ref, "same_launch_plan": same,
                                            "agree_within_3_percent": bool(abs(block["traffic_over_moved"] / ref_ratio - 1) <= 0.03) if same else None}
    it = {}
    t_mv, t_x2, t_ur = per["cg_matvec_dot"][0], per["cg_update"][0], per["cg_update_r"][0]
    if t_mv and (t_x2 or t_ur):
        t_up = 0.5 * ((t_x2 or t_ur) + (t_ur or t_x2))          # the two update forms alternate
        moved_it = MOVED_BYTES_PER_CELL["cg_iteration"] * cells
        it = {"size": n, "infinity_cache_assisted": bool(cache_assisted), "ms_matvec_dot": round(t_mv, 5), "ms_update_x2": round(t_x2, 5) if t_x2 else None,
              "ms_update_r": round(t_ur, 5) if t_ur else None, "ms_iteration": round(t_mv + t_up, 5), "moved_bytes": moved_it,
              "moved_GBs": round(moved_it / ((t_mv + t_up) * 1e-3) / 1e9, 1), "moved_frac": round(moved_it / ((t_mv + t_up) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
              "algorithmic_bytes": ALG_BYTES_PER_CELL["cg_iteration"] * cells,
              "algorithmic_equiv_frac": round(ALG_BYTES_PER_CELL["cg_iteration"] * cells / ((t_mv + t_up) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return block, it


def sync_launch_plans(ctx, sim, dist, rank, device):
    """ N > 1: every replica must run the SAME launch geometry (the summation order of the dot products follows it), or the replicas are
    equal to rounding only. Rank 0 runs one step with the first-call autotune; its plans are broadcast and pinned on every rank. """
    fams = (0, 1, 2, 3)
    plans = torch.zeros(len(fams), 3, dtype=torch.int64, device=device)
    if rank == 0:
        sim.step(None)
        if device.type == "cuda":
            device_sync(device)
        for i, f in enumerate(fams):
            q = ctx.query_plan(sim.grid, False, f)
            plans[i] = torch.tensor([q["rows"], q["tpr"], q["chunk"]], dtype=torch.int64)
        sim.reset()
    dist.broadcast(plans, src=0)
    ctx.set_autotune(False)
    for i, f in enumerate(fams):
        rows, tpr, chunk = (int(x) for x in plans[i].tolist())
        ctx.set_tuning_kernel(f, rows, tpr, chunk)
    return {f"family{f}": [int(x) for x in plans[i].tolist()] for i, f in enumerate(fams)}


def bit_checksum(tensors):
    """ order-independent checksum of the BIT patterns (int64 sum of the words + of the words weighted by position mod 65521) """
    out = []
    for t in tensors:
        w = t.contiguous().view(torch.int32).reshape(-1).to(torch.int64)
        pos = (torch.arange(w.numel(), device=w.device) % 65521) + 1
        out += [int(w.sum().item()), int((w * pos).sum().item())]
    return out


def gather_replicas(dist, world, its_local, ok_local, fields, device, pinned_plans):
    """ N > 1: every rank contributes the iteration counts of its verification step, whether they are the ones the line claims, and the bit
    checksums + pressure norm of its fields; returns (iterations of all ranks, `replicas` record). `fields[0]` is the pressure. """
    mine = torch.tensor(list(its_local) + [int(ok_local)] + bit_checksum(fields), dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    rows = [g.tolist() for g in gathered]
    iterations_all = [r[:len(its_local)] for r in rows]
    sums = [r[len(its_local) + 1:] for r in rows]
    p_norm = torch.tensor([float(fields[0].double().norm())], dtype=torch.float64, device=device)
    norms = [torch.zeros_like(p_norm) for _ in range(world)]
    dist.all_gather(norms, p_norm)
    replicas = {"bit_identical_to_rank0": [s_ == sums[0] for s_ in sums], "all_bit_identical": all(s_ == sums[0] for s_ in sums),
                "pressure_norm_rel_diff_vs_rank0": [abs(float(x) - float(norms[0])) / max(float(norms[0]), 1e-300) for x in norms],
                "verified_ok": [bool(r[len(its_local)]) for r in rows], "pinned_launch_plans": pinned_plans,
                "note": "every rank advanced the same initial state with the launch plans rank 0 tuned: checksums of the bit patterns of p and v"}
    return iterations_all, replicas


def gather_shards(dist, world, total, its_local, ok_local, own_fields, ref_fields, device):
    """ sharded batch (config4), N >= 1: every rank reports the iteration counts of its verification step (one per OWNED entry), whether they
    are the ones the line claims, a bit checksum of its owned entries, and a bit checksum of batch entry 0 recomputed on THIS rank as a
    batch of one with the analytic launch plans (same inputs, same launch geometry on every rank => the same bits on every rank: a GPU or
    rank that computes something else shows up without any cross-rank data exchange of fields). Returns the `shards` record. """
    pad = [-1] * (total - len(its_local))
    mine = torch.tensor(list(its_local) + pad + [len(its_local), int(ok_local)] + bit_checksum(own_fields) + bit_checksum(ref_fields),
                        dtype=torch.int64, device=device)
    if dist is not None:
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        rows = [g.tolist() for g in gathered]
    else:
        rows = [mine.tolist()]
    n_own = 2 * len(own_fields)
    its = [r[:r[total]] for r in rows]
    ref_sums = [r[total + 2 + n_own:] for r in rows]
    return {"iterations_per_rank": its, "entries_per_rank": [r[total] for r in rows], "verified_ok": [bool(r[total + 1]) for r in rows],
            "owned_checksums": [r[total + 2:total + 2 + n_own] for r in rows],
            "entry0_bit_identical_to_rank0": [x == ref_sums[0] for x in ref_sums], "entry0_all_bit_identical": all(x == ref_sums[0] for x in ref_sums),
            "note": "entry0_*: every rank re-ran batch entry 0 alone (analytic launch plans, same number of steps) and the bit checksums of "
                    "(p, smoke, v) were compared; owned_checksums identify the state each rank ended with"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=256, help="cells per axis (BASELINE: 256)")
    ap.add_argument("--cg-iters", type=int, default=100)
    ap.add_argument("--cpu-size", type=int, default=256, help="grid size of the CPU baseline sample and of the parity block (0 = skip); 256 = the metric's own configuration, ~30 s of NumPy")
    ap.add_argument("--profile-steps", type=int, default=1, help="extra steps with per-launch hipEvent timing for the roofline")
    ap.add_argument("--workload", default="config2", choices=["config2", "config4", "slab", "smoke256"],
                    help="config2 (default, the BASELINE metric): 256^3 Taylor-Green replicas, weak scaling. config4: 8 x 512^2 batched smoke, the batch "
                         "sharded over the GPUs, strong scaling. slab: ONE --size^3 simulation decomposed into x-slabs over the GPUs (SURVEY §8 f4). smoke256: a 3-D smoke "
                         "plume step (MacCormack smoke, advection, buoyancy, diffusion, 20 warm-started CG iterations): the share of the non-CG kernels")
    ap.add_argument("--overlap", type=int, default=0, help="slab: 1 = SlabFluid(overlap=True): the ghost-plane exchange of the advection is in flight while the "
                    "whole slab is advected, the planes within reach of a cut are redone on windows afterwards (same bits; tests/test_parallel_gloo.py)")
    ap.add_argument("--resident-cg", type=int, default=-1, choices=[-1, 0, 1, 2], help="-1: the library's default (r6: mode 1); resident solver for 2-D fp32 projections (phihip_set_resident_cg; config4: "
                    "the projection's CG iterations become ONE launch): 0 off, 1 up to the built-in cell limit, 2 whenever applicable")
    ap.add_argument("--batch-total", type=int, default=8, help="config4: simulations in the batch (all ranks together)")
    ap.add_argument("--config3-size", type=int, default=512, help="grid size of the BASELINE configs[2] block = the `roofline` kernel (pressure solve only; 0 = skip)")
    ap.add_argument("--pmc", type=int, default=1, help="1: run the rocprofv3 FETCH_SIZE / WRITE_SIZE passes for roofline.traffic inside this invocation")
    ap.add_argument("--tuning", type=str, default="", help="rows,threads_per_row,chunk override of the CG tile")
    ap.add_argument("--advect-halo", type=int, default=-1, help="phihip_set_advect_halo: -1 adaptive reach (default), 0 gather kernels, 1 / 2 fixed reach of the LDS-staged advection passes (A/B of the policy)")
    ap.add_argument("--phi-level", type=int, default=1, help="1: time the same steps through phiflow_amd.flow next to the C-ABI loop (`phi_level`; rank 0, N = 1, ~3 s)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: libraries that write to file descriptor 1 behind Python's back (RCCL prints a
    # version banner through C stdio, flushed at exit, i.e. AFTER the record) are sent to stderr instead
    global _RECORD_FD
    sys.stdout.flush()
    _RECORD_FD = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    device = device_for_rank(local_rank)
    dist = None
    force_dist = os.environ.get("PHIHIP_BENCH_FORCE_DIST") == "1"          # lets a 1-GPU box exercise the RCCL path incl. the replica validation
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(DIST_BACKEND, rank=rank, world_size=world, **({"device_id": device} if device.type == "cuda" else {}))

    lib = load_library()
    ctx = C.Context(lib, local_rank if device.type == "cuda" else 0)
    if args.resident_cg >= 0:
        ctx.set_resident_cg(args.resident_cg)
    if args.advect_halo != -1:
        ctx.set_advect_halo(args.advect_halo)
    if args.tuning:
        ctx.set_tuning(*[int(x) for x in args.tuning.split(",")])
    n, B = args.size, 1

    def allreduce(t):
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)

    def barrier():
        if dist is not None:
            dist.barrier()
        device_sync(device)

    if args.workload == "config4":
        return bench_config4(args, ctx, device, rank, world, dist, barrier, allreduce)
    if args.workload == "slab":
        return bench_slab(args, lib, device, rank, world, dist, barrier)
    if args.workload == "smoke256":
        return bench_smoke3d(args, ctx, lib, device, rank, world, dist, barrier, allreduce)
    sim = FluidStep(ctx, n, B, args.cg_iters, device)
    pinned_plans = sync_launch_plans(ctx, sim, dist, rank, device) if dist is not None and (world > 1 or force_dist) else None

    for _ in range(args.warmup):
        sim.step(allreduce)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rel = sim.step(allreduce)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cells = n ** 3 * B
    value = cells * world * args.steps / elapsed

    # ---- the timed steps must have run the iterations they claim (an entry that freezes -- diverged / residual 0 -- makes the remaining
    # launches return at once and would overstate the throughput): one more step that reports, ON EVERY RANK; with it the bit checksum of
    # the rank's fields (replicas start from the same state and run the same launch plans: they must agree bit for bit) ----
    pv2 = [t.data_ptr() for t in sim.v2]
    ctx.advect_staggered(sim.grid, [t.data_ptr() for t in sim.v], [t.data_ptr() for t in sim.v], pv2, sim.dt, sim.stream)
    info = ctx.make_incompressible(sim.grid, pv2, None, 0, 1, True, sim.p.data_ptr(), sim.div.data_ptr(), sim.solve, want_info=True, stream=sim.stream)
    sim.v, sim.v2 = sim.v2, sim.v
    its_local = [int(i.iterations) for i in info]
    ok_local = all(i.iterations == args.cg_iters and not i.diverged for i in info)
    replicas = None
    iterations_all = [its_local]
    if dist is not None and (world > 1 or force_dist):
        iterations_all, replicas = gather_replicas(dist, world, its_local, ok_local, [sim.p] + sim.v, device, pinned_plans)
        assert all(replicas["verified_ok"]), iterations_all
    else:
        assert ok_local, [(i.iterations, i.diverged, i.residual_sq) for i in info]

    # ---- benchmark configuration (256^3): hipEvent pairs around every launch on the solve stream, extra profiled steps ----
    extra = {}
    roofline_bench = None
    if rank == 0 and args.profile_steps > 0:
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.profile_steps):
            sim.step(None)
        e1.record()
        torch.cuda.synchronize(device)
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        per = {k: (v[1] / v[0] if v[0] else None, v[0], v[1]) for k, v in prof.items()}
        # The per-launch figures come from EXTRA steps with a hipEvent pair around every launch: the pairs serialise what the timed region
        # overlaps (the next launch's dispatch behind the running kernel) and add ~2 us each, so their sum exceeds `ms_per_step` -- it must not
        # exceed the wall time of the profiled steps themselves, which is reported next to it
        profiled_ms = e0.elapsed_time(e1) / args.profile_steps
        ksum = sum(v[2] for v in per.values()) / args.profile_steps
        assert ksum <= 1.02 * profiled_ms, (ksum, profiled_ms)
        extra["profiled_step"] = {"ms_wall": round(profiled_ms, 5), "sum_kernel_ms": round(ksum, 5), "launches": int(sum(v[1] for v in per.values()) / args.profile_steps),
                                  "note": "kernel_ms_per_step / kernel_ms_per_launch are hipEvent intervals of these extra steps (one event pair per launch: "
                                          "slower than the untouched step of ms_per_step by the events' own cost); rocprofv3 --kernel-trace figures of the "
                                          "untouched step: profiles/r06_bench256_kernel_stats.csv"}
        plans_b = {str(fam): [q["rows"], q["tpr"], q["chunk"]] for fam, q in ((fam, ctx.query_plan(sim.grid, False, fam)) for fam in (0, 1, 2, 3))}
        roofline_bench, it = roofline_block(n, per, bool(args.pmc) and rank == 0, world, cache_assisted=4 * 4 * n ** 3 <= 256 * 2 ** 20,
                                            where="profiled steps of the timed benchmark configuration", plans=plans_b)
        if it:
            extra["roofline_cg_iteration_256" if n == 256 else f"roofline_cg_iteration_{n}"] = it
        extra["kernel_ms_per_launch"] = {k: (round(v[0], 5) if v[0] else None) for k, v in per.items()}
        extra["kernel_ms_per_step"] = {k: round(v[2] / args.profile_steps, 5) for k, v in per.items()}
        extra["plan"] = {name: ctx.query_plan(sim.grid, False, fam) for name, fam in (("matvec", 1), ("update_x2", 2), ("update_r", 3))}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_size > 0:
        cpu, ref = cpu_baseline(args.cpu_size, args.cg_iters)
        extra["parity"] = parity_block(ctx, args.cpu_size, args.cg_iters, device, ref)
        del ref
        extra["cpu_cg_variants"] = cpu_cg_variants(96, 20)

    if rank == 0 and world == 1 and args.phi_level and n == 256:
        extra["phi_level"] = phi_level_block(ctx, lib, device, args.cg_iters)

    # ---- the HBM-resident configuration: 512^3 pressure solve (BASELINE configs[2]) -> `roofline` ----
    roofline = None
    if rank == 0 and args.config3_size > 0:
        del sim
        torch.cuda.empty_cache()
        if pinned_plans is not None:
            # N > 1 (or the forced-dist rehearsal): the replicas ran rank 0's 256^3 plans PINNED (sync_launch_plans). The 512^3 block below is another grid: with those
            # plans (16-plane chunks) its iteration took 0.80 ms instead of 0.70 in the r6 rehearsal -- the `roofline` of an N > 1 line would have priced a mis-planned
            # kernel. Unpin and let the first call tune for this grid, as at N = 1.
            for fam in (0, 1, 2, 3):
                ctx.set_tuning_kernel(fam, 0, 0, 0)
            ctx.set_autotune(True)
        c3 = config3_block(ctx, device, args.config3_size, args.cg_iters)
        # r6: at identical launch plans and virtual addresses a 512^3 iteration draws one of two levels (+- 3 %) with the PHYSICAL pages of the three workspace vectors
        # (profiles/r06_autotune_stability.txt) -- one context is one draw. Two more fresh contexts (their own workspaces, their own first-call autotune); the line reports
        # all three and prices the `roofline` on the MEDIAN one. (Last session of r6: every context now CHOOSES its workspace among candidate allocations by timing the
        # iteration loop -- `workspace_placement` in each draw -- which takes most of that spread away; the estimator stays.)
        draws = [c3]
        if device.type == "cuda" and args.config3_size >= 384:
            for _ in range(2):
                extra_ctx = C.Context(lib, device.index or 0)
                draws.append(config3_block(extra_ctx, device, args.config3_size, args.cg_iters))
                del extra_ctx
                torch.cuda.empty_cache()
            order = sorted(range(3), key=lambda i: draws[i]["ms_per_iteration"])
            c3 = draws[order[1]]
            c3["contexts"] = {"estimator": "median of three fresh contexts (each with its own workspace allocation and first-call autotune); all three listed in run order",
                              "ms_per_iteration": [d_["ms_per_iteration"] for d_ in draws],
                              "ms_matvec_per_launch": [d_["kernel_ms_per_launch"].get("cg_matvec_dot") for d_ in draws],
                              "matvec_plan": [[d_["plan"]["matvec"][k] for k in ("rows", "tpr", "chunk")] for d_ in draws],
                              "workspace_placement": [d_.get("workspace_placement") for d_ in draws], "chosen": order[1]}
        extra["config3"] = c3
        per3 = {k: (c3["kernel_ms_per_launch"].get(k), c3["launches"].get(k, 0), (c3["kernel_ms_per_launch"].get(k) or 0.0) * c3["launches"].get(k, 0))
                for k in C.K_NAMES}
        plans3 = {str(fam): [q["rows"], q["tpr"], q["chunk"]] for fam, q in ((fam, c3["plan_all"][str(fam)]) for fam in (0, 1, 2, 3))}
        roofline, it3 = roofline_block(args.config3_size, per3, bool(args.pmc), world, cache_assisted=False,
                                       where=f"{args.config3_size}^3 fp32 pressure solve run inside this invocation (`config3`)", plans=plans3)
        if it3:
            it3["ms_iteration_wall"] = c3["ms_per_iteration"]
            it3["moved_frac_wall"] = c3["moved_frac"]
            extra[f"roofline_cg_iteration_{args.config3_size}"] = it3
    if roofline is None:
        roofline = roofline_bench
    elif roofline_bench is not None:
        extra["roofline_256" if n == 256 else f"roofline_{n}"] = roofline_bench

    if rank == 0:
        out = {
            "metric": "cell-updates/sec (advect+100 CG iters), 256^3 fp32", "value": value, "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"3D periodic Taylor-Green {n}^3 fp32, semi-Lagrangian advect + projection with {args.cg_iters} CG "
                                   f"iterations/step (BASELINE.json configs[1])", "cells_per_gpu": cells, "batch_per_gpu": B,
                       "cg_iterations": args.cg_iters, "parallelism": f"batch-parallel replicas x{world}, 1 all-reduce(max residual)/step"},
            "final_relative_residual": float(rel.item()) if rel is not None else None,
            "iterations_verified": iterations_all[0] if world == 1 else iterations_all,
            "replicas": replicas, "build_id": lib.build_id(),
            "scaling_measured": ("this line is ONE point of the curve (n_gpus above); the driver computes efficiency from the per-N lines -- no "
                                 "multi-GPU curve has been measured by the builder (single-GPU boxes only)"),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        out.update(extra)
        emit_record(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
