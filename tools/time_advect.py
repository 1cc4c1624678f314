#!/usr/bin/env python3
"""
Times the advection kernels (semi-Lagrangian staggered self-advection, MacCormack, centred scalar) at a cubic size with hipEvent
pairs:  python tools/time_advect.py --size 256 [--lib other/libphihip.so]
"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--lib", default="")
    ap.add_argument("--field", default="random", help="random | tg (the smooth Taylor-Green velocity of the benchmark, CFL 0.5)")
    ap.add_argument("--cfl", type=float, default=0.5, help="dt = cfl * dx (random field: |u| ~ N(0,1))")
    ap.add_argument("--bc", type=int, default=0, help="boundary code of every side: 0 periodic, 1 closed, 2 open")
    args = ap.parse_args()
    n = args.size
    dev = torch.device("cuda:0")
    ctx = C.Context(C.Library(args.lib, strict=False) if args.lib else C.load_default_library(), 0)
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    L = 2 * math.pi
    grid = C.make_grid(3, C.PHIHIP_F64 if args.dtype == "f64" else C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((args.bc, args.bc),) * 3)
    g = torch.Generator().manual_seed(0)
    shapes = [tuple(n + (0, -1, 1)[args.bc] if a == d else n for a in range(3)) for d in range(3)]
    v = [torch.randn(1, *sh, generator=g, dtype=tdt).to(dev) for sh in shapes]
    if args.field == "tg":
        h = L / n
        face = (torch.arange(n, dtype=tdt) * h)
        cent = ((torch.arange(n, dtype=tdt) + 0.5) * h)
        u = (torch.cos(face)[:, None, None] * torch.sin(cent)[None, :, None]).expand(n, n, n)
        w = (-torch.sin(cent)[:, None, None] * torch.cos(face)[None, :, None]).expand(n, n, n)
        full = [u, w, torch.zeros(n, n, n, dtype=tdt)]
        v = [full[d][tuple(slice(0, sh[a]) for a in range(3))].contiguous()[None].to(dev) for d, sh in enumerate(shapes)]
    out = [torch.empty_like(t) for t in v]
    s = torch.randn(1, n, n, n, generator=g, dtype=tdt).to(dev)
    so = torch.empty_like(s)
    P = lambda ts: [t.data_ptr() for t in ts]
    dt = args.cfl * L / n

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    res = {"lib": os.path.basename(args.lib) if args.lib else "default", "size": n, "dtype": args.dtype, "bc": args.bc, "field": args.field, "cfl": args.cfl}
    if hasattr(ctx.lib.dll, "phihip_set_advect_halo") and not args.lib:
        for halo in (0, 1, 2, 3):      # 0: gather kernels (one launch per component); 1 / 2: LDS-staged tiles (advect_tile.hip)
            ctx.set_advect_halo(halo)
            res[f"ms_semi_lagrangian_staggered_halo{halo}"] = round(timed(lambda: ctx.advect_staggered(grid, P(v), P(v), P(out), dt)), 5)
        ctx.set_advect_halo(1)
        for chunk in (8, 16, 32, 64):
            ctx.set_advect_chunk(chunk)
            res[f"ms_halo1_chunk{chunk}"] = round(timed(lambda: ctx.advect_staggered(grid, P(v), P(v), P(out), dt)), 5)
        ctx.set_advect_chunk(0)
    res.update({
           "ms_semi_lagrangian_staggered": round(timed(lambda: ctx.advect_staggered(grid, P(v), P(v), P(out), dt)), 5),
           "ms_mac_cormack_staggered": round(timed(lambda: ctx.mac_cormack_staggered(grid, P(v), P(v), P(out), dt, 1.0)), 5),
           "ms_semi_lagrangian_centered": round(timed(lambda: ctx.advect_centered(grid, s.data_ptr(), ((args.bc and 2, args.bc and 2),) * 3, None, P(v), so.data_ptr(), dt)), 5)})
    sbc = ((args.bc and 2, args.bc and 2),) * 3
    res["ms_mac_cormack_centered"] = round(timed(lambda: ctx.mac_cormack_centered(grid, s.data_ptr(), sbc, None, P(v), so.data_ptr(), dt, 1.0)), 5)
    res["GBs_semi_lagrangian_staggered"] = round(6 * n ** 3 * (8 if args.dtype == "f64" else 4) / res["ms_semi_lagrangian_staggered"] / 1e6, 1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
