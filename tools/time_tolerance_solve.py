#!/usr/bin/env python3
""" wall time of tolerance-mode CG solves (host polls the device-side continue flags every `check_every` iterations):
    python tools/time_tolerance_solve.py [--lib other/libphihip.so] """
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    args = ap.parse_args()
    ctx = C.Context(C.Library(args.lib, strict=False) if args.lib else C.load_default_library(), 0)
    dev = torch.device("cuda:0")
    for shape, B, rtol in (((512, 512), 8, 1e-3), ((256, 256), 8, 1e-4), ((128, 128, 128), 1, 1e-5), ((256, 256, 256), 1, 1e-5)):
        rank = len(shape)
        grid = C.make_grid(rank, C.PHIHIP_F32, B, shape, (0,) * rank, tuple(float(n) for n in shape), ((1, 1),) * rank)
        rhs = torch.randn((B,) + shape, generator=torch.Generator().manual_seed(0))
        rhs -= rhs.mean(dim=tuple(range(1, rank + 1)), keepdim=True)
        rhs = rhs.to(dev)
        x = torch.zeros_like(rhs)
        solve = C.Solve(rtol, 0.0, 20000, 50, 10, 0)
        ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve)
        best = None
        for _ in range(3):
            x.zero_(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            info = ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        its = max(i.iterations for i in info)
        print(json.dumps({"lib": os.path.basename(args.lib) if args.lib else "default", "shape": shape, "batch": B, "rtol": rtol,
                          "iterations": its, "converged": all(i.converged for i in info), "ms": round(best * 1e3, 3),
                          "us_per_iteration": round(best / its * 1e6, 2)}), flush=True)


if __name__ == "__main__":
    main()
