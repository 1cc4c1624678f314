#!/usr/bin/env python3
""" Tile sweep of the CG kernels on a batched 2-D grid (BASELINE configs[3] shape: 8 x 512^2, closed box):
    python tools/sweep_cg2d.py --size 512 --batch 8 """
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--method", type=int, default=0, help="phihip_method: 0 = CG, 1 = CG-adaptive")
    ap.add_argument("--lib", default="")
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--small-limit", type=int, default=1, help="cell limit of the single-kernel solver (1 = built-in rule)")
    args = ap.parse_args()
    n, B = args.size, args.batch
    dev = torch.device("cuda:0")
    ctx = C.Context(C.Library(args.lib, strict=False) if args.lib else C.load_default_library(), 0)
    D = args.dim
    grid = C.make_grid(D, C.PHIHIP_F32, B, (n,) * D, (0,) * D, (100.0,) * D, ((1, 1),) * D)
    rhs = torch.randn((B,) + (n,) * D, generator=torch.Generator().manual_seed(0))
    rhs -= rhs.mean(dim=tuple(range(1, D + 1)), keepdim=True)
    rhs = rhs.to(dev)
    x = torch.zeros_like(rhs)
    solve = C.Solve(0.0, 0.0, args.iters, 0, 0, args.method)
    ap_small = n ** D <= max(8192, args.small_limit)
    for rows, tpr in ([(-1, -1)] if ap_small else []) + [(0, 0)] + ([(1, 16), (2, 16), (2, 32), (4, 32), (4, 64), (1, 64), (2, 64), (1, 32)] if D == 2 else []):
        ctx.lib.check(ctx.lib.dll.phihip_set_small_grid_solver(ctx.handle, args.small_limit if rows < 0 else 0))   # rows = -1: cg_small.hip
        ctx.set_tuning(max(rows, 0), max(tpr, 0), 0)
        x.zero_()
        ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 3, 0, 0, args.method), want_info=False)
        torch.cuda.synchronize()
        x.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.iters * 1e3
        plans = {f: list(ctx.query_plan(grid, False, f).values()) for f in (1, 2)}
        print(json.dumps({"size": n, "batch": B, "rows": rows, "tpr": tpr, "us_per_iteration": round(us, 2),
                          "alg_GBs": round(40 * B * n ** D / us / 1e3, 1), "plan_mv": plans[1], "plan_up": plans[2]}), flush=True)


if __name__ == "__main__":
    main()
