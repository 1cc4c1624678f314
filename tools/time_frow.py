#!/usr/bin/env python3
"""
hipEvent timings of the NON-CG kernels of a fluid step at one size, one JSON line per run:
the kernels the round-3 verdict lists below 0.55 of the HBM peak. `--lib other.so` times another build of the library (same-box A/B:
boxes differ by +-5 %, only figures of one gpurun call compare).

    python tools/time_frow.py --size 256 --dtype f32 --bc periodic [--lib phiflow_amd/lib/libphihip_r3.so] [--reps 30]
    python tools/time_frow.py --size 384 --dtype f64 --bc closed

Every figure is the mean of `reps` back-to-back calls (the launch gap of a dependent launch is part of it, as in a real step); `GBs` is
bytes moved by construction (one read per input word, one write per output word) / time.
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phiflow_amd import _capi as C   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--bc", default="periodic", choices=["periodic", "closed", "open"])
    ap.add_argument("--lib", default="")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--only", default="")
    ap.add_argument("--halo", type=int, default=-1, help="phihip_set_advect_halo: -1 adaptive reach (the library's default), 0 gather kernels, 1 / 2 fixed reach of the LDS windows")
    ap.add_argument("--cfl", type=float, default=0.5, help="max |u| dt / dx of the smooth test field (> 1: the fastest regions leave the LDS windows: fix-up pass)")
    ap.add_argument("--device", default="cuda:0", help="cpu + --lib tests/hipemu/libphihip_emu.so = dry run of the call sequence")
    a = ap.parse_args()
    lib = C.Library(a.lib, strict=False) if a.lib else C.load_default_library()
    ctx = C.Context(lib, 0)
    if a.halo != -1:
        ctx.set_advect_halo(a.halo)
    dev = torch.device(a.device)
    gpu = dev.type == "cuda"
    n, D, B = a.size, a.rank, a.batch
    f64 = a.dtype == "f64"
    tdt = torch.float64 if f64 else torch.float32
    w = 8 if f64 else 4
    code = {"periodic": C.BC_PERIODIC, "closed": C.BC_CLOSED, "open": C.BC_OPEN}[a.bc]
    L = 2 * math.pi
    res = (n,) * D
    grid = C.make_grid(D, C.PHIHIP_F64 if f64 else C.PHIHIP_F32, B, res, (0,) * D, (L,) * D, ((code, code),) * D)
    shapes = [tuple(ctx.component_shape(grid, d)) for d in range(D)]
    h = L / n
    g = torch.Generator(device=dev).manual_seed(1)
    # smooth field, CFL ~ 0.5 (the tiled advection stays in its LDS window, as in the benchmark step)
    vel = []
    for d, sh in enumerate(shapes):
        idx = [torch.arange(s, device=dev, dtype=torch.float64) * h for s in sh]
        f = torch.ones(sh, device=dev, dtype=torch.float64)
        for k, x in enumerate(idx):
            view = [1] * D
            view[k] = sh[k]
            f = f * (torch.cos(x + 0.3 * k) if (k + d) % 2 == 0 else torch.sin(x + 0.2 * d)).view(view)
        vel.append(f.to(tdt).unsqueeze(0).expand(B, *sh).contiguous())
    out = [torch.empty_like(t) for t in vel]
    s = torch.rand((B,) + res, device=dev, dtype=tdt, generator=g)
    s2 = torch.empty_like(s)
    p = torch.randn((B,) + res, device=dev, dtype=tdt, generator=g)
    div = torch.empty_like(p)
    dt = a.cfl * h
    s_bc = ((C.BC_PERIODIC, C.BC_PERIODIC),) * D if a.bc == "periodic" else ((C.BC_OPEN, C.BC_OPEN),) * D
    P = lambda ts: [t.data_ptr() for t in ts]
    N = B * n ** D
    NC = [B * math.prod(sh) for sh in shapes]
    NV = sum(NC)

    def timed(fn):
        fn()
        if not gpu:
            import time
            t0 = time.perf_counter()
            fn()
            return (time.perf_counter() - t0) * 1e3
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    vec_all = [0.3, -1.5, 0.1][:D]
    vec_one = [0.0] * (D - 1) + [0.1]
    cases = {
        # name: (callable, bytes moved)
        "advect_self": (lambda: ctx.advect_staggered(grid, P(vel), P(vel), P(out), dt), 2 * NV * w),
        "mac_cormack_self": (lambda: ctx.mac_cormack_staggered(grid, P(vel), P(vel), P(out), dt, 1.0), 2 * NV * w),
        "advect_centered": (lambda: ctx.advect_centered(grid, s.data_ptr(), s_bc, None, P(vel), s2.data_ptr(), dt), (2 * N + NV) * w),
        "mac_cormack_centered": (lambda: ctx.mac_cormack_centered(grid, s.data_ptr(), s_bc, None, P(vel), s2.data_ptr(), dt, 1.0), (2 * N + NV) * w),
        "diffuse_explicit": (lambda: ctx.diffuse_explicit(grid, P(vel), P(out), 0.1 * h * h), 2 * NV * w),
        "diffuse_explicit_centered": (lambda: ctx.diffuse_explicit_centered(grid, s.data_ptr(), s_bc, None, s2.data_ptr(), 0.1 * h * h), 2 * N * w),
        "resample_all": (lambda: ctx.centered_to_staggered(grid, s.data_ptr(), s_bc, None, vec_all, False, P(out)), (N + NV) * w),
        "buoyancy_accumulate_one": (lambda: ctx.centered_to_staggered(grid, s.data_ptr(), s_bc, None, vec_one, True, P(out)), (N + 2 * NC[-1]) * w),
        "divergence_balance": (lambda: ctx.divergence(grid, P(vel), 0, 1, 1, div.data_ptr()), (NV + 2 * N) * w + N * w),
        "divergence": (lambda: ctx.divergence(grid, P(vel), 0, 1, 0, div.data_ptr()), (NV + N) * w),
        "laplace_apply": (lambda: ctx.laplace_apply(grid, 0, 1, p.data_ptr(), div.data_ptr()), 2 * N * w),
        "grad_subtract": (lambda: ctx.grad_subtract(grid, 0, 1, p.data_ptr(), P(out)), (N + 2 * NV) * w),
    }
    # f3 / a7 (r5): the mask kernels of a MOVING-obstacle step -- rasterisation, stencil flags, apply_boundary_conditions -- with the two obstacles of
    # tools/path_workload.py (a moving box of L / 4 and a rotating sphere of radius L / 10)
    if D == 3:
        obs = C.make_obstacles([dict(kind=C.OBSTACLE_BOX, center=(L / 2, L / 2, L / 2), half_size=(L / 8, L / 8, L / 8), velocity=(0.1, 0, 0), angular_velocity=(0, 0, 0)),
                                dict(kind=C.OBSTACLE_SPHERE, center=(L / 4, L / 4, L / 4), half_size=(L / 10, L / 10, L / 10), velocity=(0, 0, 0), angular_velocity=(0, 0, 0.2))])
        acc = torch.empty(res, device=dev, dtype=torch.uint8)
        fl = torch.empty(res, device=dev, dtype=torch.uint8)
        g1 = C.make_grid(D, C.PHIHIP_F64 if f64 else C.PHIHIP_F32, 1, res, (0,) * D, (L,) * D, ((code, code),) * D)
        tmp = [t.clone() for t in vel]
        cases["obstacle_accessible"] = (lambda: ctx.obstacle_accessible(g1, obs, 2, acc.data_ptr()), n ** D)
        cases["build_cellflags"] = (lambda: ctx.build_cellflags(g1, acc.data_ptr(), 0, 1, fl.data_ptr()), 2 * n ** D)
        cases["apply_obstacles"] = (lambda: ctx.apply_obstacles(grid, obs, 2, P(tmp)), 2 * NV * w)
    only = [x for x in a.only.split(",") if x]
    rec = {"lib": os.path.basename(a.lib) if a.lib else "default", "build_id": lib.build_id() if hasattr(lib, "build_id") else None,
           "size": n, "rank": D, "batch": B, "dtype": a.dtype, "bc": a.bc, "cfl": a.cfl, "halo": a.halo, "reps": a.reps, "kernels": {}}
    for name, (fn, nbytes) in cases.items():
        if only and name not in only:
            continue
        try:
            ms = timed(fn)
        except Exception as exc:      # (an older library without the entry point)
            rec["kernels"][name] = {"error": str(exc)[:80]}
            continue
        rec["kernels"][name] = {"ms": round(ms, 5), "GBs": round(nbytes / ms / 1e6, 1), "frac": round(nbytes / ms / 1e6 / 8000.0, 3)}
    if hasattr(ctx, "advect_fallback_stats"):
        try:
            ctx.advect_staggered(grid, P(vel), P(vel), P(out), dt)
            if gpu:
                torch.cuda.synchronize()
            rec["advect_fallback"] = list(ctx.advect_fallback_stats())
        except Exception:
            pass
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
