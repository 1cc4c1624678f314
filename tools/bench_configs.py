#!/usr/bin/env python3
"""
Timing of the BASELINE.json configurations that are not the headline bench line (one JSON line each):
  config 3: 512^3 fp32 periodic pressure solve only, 100 fixed CG iterations  (HBM-roofline microbenchmark)
  config 4: batched 2-D smoke 8 x 512^2 on ONE GPU (B = 8; the 8-GPU run places one simulation per GPU)
  config 5: 3-D lid-driven cavity 384^3 fp64 with a solid box obstacle, 100 fixed CG iterations
    python tools/bench_configs.py [3] [4] [5]
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C   # noqa: E402

dev = torch.device("cuda:0")
lib = C.load_default_library()
ctx = C.Context(lib, 0)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def kernel_profile(fn):
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    fn()
    torch.cuda.synchronize()
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    return {k: (round(v[1] / v[0], 5) if v[0] else None) for k, v in prof.items()}


def config3(n=512, iters=100, label="3"):
    L = 2 * math.pi
    grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
    if n <= 512:
        g = torch.Generator(device="cpu").manual_seed(0)
        rhs = torch.randn(1, n, n, n, generator=g)
        rhs -= rhs.mean()
        rhs = rhs.to(dev)
    else:                                         # (1024^3: 4.3 GB per array -- generated on the device)
        rhs = torch.randn(1, n, n, n, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
        rhs -= rhs.mean()
    x = torch.zeros_like(rhs)
    solve = C.Solve(0.0, 0.0, iters, 50, 0, 0)

    def run():
        x.zero_()
        ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False)
    t = timed(run, 3)
    prof = kernel_profile(run)
    cells = n ** 3
    moved = 28.0 * cells * iters          # 7 fp32 words per cell and iteration move by construction; SURVEY §8d's textbook count is 40 B
    print(json.dumps({"config": f"{label}: {n}^3 fp32 periodic pressure solve, {iters} CG iterations",
                      "plans": {f: ctx.query_plan(grid, False, k) for f, k in (("matvec", 1), ("update_x2", 2), ("update_r", 3))},
                      "ms_per_solve": t * 1e3, "ms_per_iteration": t * 1e3 / iters,
                      "moved_GBs": moved / t / 1e9, "moved_frac_of_8TBs": moved / t / 8e12, "textbook_40B_equiv_frac": 40.0 * cells * iters / t / 8e12,
                      "kernel_ms": prof}), flush=True)


def config4():
    n, B, iters = 512, 8, 100
    grid = C.make_grid(2, C.PHIHIP_F32, B, (n, n), (0, 0), (100, 100), ((1, 1), (1, 1)))
    rng = np.random.default_rng(0)
    v = [torch.from_numpy((rng.standard_normal((B, n - 1, n)) * 0.1).astype(np.float32)).to(dev),
         torch.from_numpy((rng.standard_normal((B, n, n - 1)) * 0.1).astype(np.float32)).to(dev)]
    v2 = [torch.empty_like(t) for t in v]
    p = torch.zeros(B, n, n, device=dev)
    solve = C.Solve(0.0, 0.0, iters, 50, 0, 0)

    def step():
        ctx.advect_staggered(grid, [t.data_ptr() for t in v], [t.data_ptr() for t in v], [t.data_ptr() for t in v2], 0.5)
        ctx.make_incompressible(grid, [t.data_ptr() for t in v2], None, 0, 1, True, p.data_ptr(), 0, solve, want_info=False)
    t = timed(step, 5)
    prof = kernel_profile(step)
    rec = {"config": "4: batched 2-D smoke 8 x 512^2 fp32 closed box, advect + 100 CG iterations, one GPU", "ms_per_step": t * 1e3,
           "cell_updates_per_s": B * n * n / t, "us_per_cg_iteration": t * 1e6 / iters, "kernel_ms": prof}
    # the same step with the opt-in resident solver (cg_resident.hip: the projection's 100 iterations are ONE launch), and ONE simulation
    # alone = what a GPU of the 8-GPU sharded run does
    p_launch = p.clone()
    ctx.set_resident_cg(2)
    p.zero_()
    tr = timed(step, 5)
    rec["resident_cg"] = {"ms_per_step": tr * 1e3, "cell_updates_per_s": B * n * n / tr, "us_per_cg_iteration": tr * 1e6 / iters, "kernel_ms": kernel_profile(step)}
    ctx.set_resident_cg(0)
    grid1 = C.make_grid(2, C.PHIHIP_F32, 1, (n, n), (0, 0), (100, 100), ((1, 1), (1, 1)))
    v1, v21, p1 = [t[:1].contiguous() for t in v], [torch.empty_like(t[:1]) for t in v], torch.zeros(1, n, n, device=dev)

    def step1():
        ctx.advect_staggered(grid1, [t.data_ptr() for t in v1], [t.data_ptr() for t in v1], [t.data_ptr() for t in v21], 0.5)
        ctx.make_incompressible(grid1, [t.data_ptr() for t in v21], None, 0, 1, True, p1.data_ptr(), 0, solve, want_info=False)
    rec["one_entry"] = {}
    for label, mode in (("launches", 0), ("resident_cg", 2)):
        ctx.set_resident_cg(mode)
        t1 = timed(step1, 5)
        rec["one_entry"][label] = {"ms_per_step": t1 * 1e3, "us_per_cg_iteration": t1 * 1e6 / iters}
    ctx.set_resident_cg(0)
    print(json.dumps(rec), flush=True)


def config5():
    n, iters = 384, 100
    bcv = np.zeros((3, 2, 3)); bcv[2, 1, 0] = 1.0
    grid = C.make_grid(3, C.PHIHIP_F64, 1, (n, n, n), (0, 0, 0), (1, 1, 1), ((1, 1),) * 3, bcv)
    grid1 = C.make_grid(3, C.PHIHIP_F64, 1, (n, n, n), (0, 0, 0), (1, 1, 1), ((1, 1),) * 3, bcv)
    c = (np.arange(n) + 0.5) / n
    inside = (np.abs(c - 0.5) <= 0.125)
    acc = ~(inside[:, None, None] & inside[None, :, None] & inside[None, None, :])
    acc_t = torch.from_numpy(acc.astype(np.uint8)).to(dev)
    flags = torch.empty(n, n, n, dtype=torch.uint8, device=dev)
    ctx.build_cellflags(grid1, acc_t.data_ptr(), 0, 1, flags.data_ptr())
    shapes = [ctx.component_shape(grid, d) for d in range(3)]
    g = torch.Generator(device="cpu").manual_seed(0)
    v = [(torch.randn(1, *s, generator=g, dtype=torch.float64) * 0.01).to(dev) for s in shapes]
    v2 = [torch.empty_like(t) for t in v]
    p = torch.zeros(1, n, n, n, dtype=torch.float64, device=dev)
    solve = C.Solve(0.0, 0.0, iters, 50, 0, 0)
    dt = 0.5 / n

    def step():
        ctx.advect_staggered(grid, [t.data_ptr() for t in v], [t.data_ptr() for t in v], [t.data_ptr() for t in v2], dt)
        ctx.make_incompressible(grid, [t.data_ptr() for t in v2], None, flags.data_ptr(), 1, True, p.data_ptr(), 0, solve, want_info=False)
    t = timed(step, 3)
    prof = kernel_profile(step)
    cells = n ** 3
    moved_iter = 58.0 * cells    # 7 fp64 words + 1 flag byte per kernel (2 kernels) per cell and iteration; textbook: 81 B
    upd = [prof[k] for k in ("cg_update", "cg_update_r") if prof.get(k)]
    it_ms = (prof["cg_matvec_dot"] or 0) + (sum(upd) / len(upd) if upd else 0)      # the two update forms alternate
    plans = {name: ctx.query_plan(grid, True, fam) for name, fam in (("matvec", 1), ("update_x2", 2), ("update_r", 3), ("residual", 0))}
    print(json.dumps({"config": "5: lid-driven cavity 384^3 fp64, closed box + solid box obstacle, advect + 100 CG iterations", "ms_per_step": t * 1e3, "plan": plans,
                      "cell_updates_per_s": cells / t, "cg_iteration_ms": it_ms, "cg_moved_GBs": moved_iter / (it_ms * 1e-3) / 1e9 if it_ms else None,
                      "cg_moved_frac_of_8TBs": moved_iter / (it_ms * 1e-3) / 8e12 if it_ms else None,
                      "kernel_ms": prof}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4", "5"]
    for w in which:
        {"3": config3, "4": config4, "5": config5, "6": lambda: config3(1024, 40, "6 (beyond BASELINE: the largest grid of the test suite)")}[w]()
