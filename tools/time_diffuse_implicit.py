#!/usr/bin/env python3
"""
diffuse.implicit on the MI355X: time per CG iteration of the operator I - k dt L on the lattices of a staggered velocity and of a centred
scalar (fixed iteration count, tolerances 0), next to the pressure solve of the same grid.
    python tools/time_diffuse_implicit.py --size 256 --bc 0
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--bc", type=int, default=0, help="0 periodic, 1 closed, 2 open (every side)")
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ctx = C.Context(C.load_default_library(), 0)
    n = args.size
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    esize = 8 if args.dtype == "f64" else 4
    grid = C.make_grid(3, C.PHIHIP_F64 if args.dtype == "f64" else C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (1, 1, 1), ((args.bc, args.bc),) * 3)
    shapes = [ctx.component_shape(grid, d) for d in range(3)]
    g = torch.Generator(device=dev).manual_seed(0)
    v = [torch.randn(1, *s, generator=g, device=dev, dtype=tdt) for s in shapes]
    out = [torch.empty_like(t) for t in v]
    sc = torch.randn(1, n, n, n, generator=g, device=dev, dtype=tdt)
    sco = torch.empty_like(sc)
    solve = C.Solve(0.0, 0.0, args.iters, 0, 0, 0)
    kdt = 2.0 / n ** 2
    s_codes = ((args.bc, args.bc),) * 3

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    t_v = timed(lambda: ctx.diffuse_implicit(grid, [t.data_ptr() for t in v], [t.data_ptr() for t in out], kdt, solve))
    t_s = timed(lambda: ctx.diffuse_implicit_centered(grid, sc.data_ptr(), s_codes, [(0.0, 0.0)] * 3, sco.data_ptr(), kdt, solve))
    rhs = sc - sc.mean()
    x = torch.zeros_like(rhs)
    t_p = timed(lambda: (x.zero_(), ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False)))
    cells = n ** 3
    it = args.iters
    rec = {"size": n, "dtype": args.dtype, "bc": args.bc, "iterations": it,
           "ms_velocity_3_components": t_v * 1e3, "us_per_iteration_component": t_v / 3 / it * 1e6,
           "ms_centred_scalar": t_s * 1e3, "us_per_iteration_scalar": t_s / it * 1e6, "moved_GBs_scalar": 7 * esize * cells / (t_s / it) / 1e9,
           "us_per_iteration_pressure": t_p / it * 1e6, "rows": [s[-1] for s in shapes]}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
