#!/usr/bin/env python3
"""
Is the mid-size dip of the CG kernels (288^3 ... 448^3) a property of the kernels or of the memory system? Times, per
cubic size, plain streaming passes over arrays of the same size (torch copy: 2 words, torch triad out = a + beta b: 3 words) next to the
marching kernels (APPLY: 2 words, one stencil source; MATVEC: 3 words, two stencil sources; UPDATE_R: 3 words, one stencil source) and
prints one JSON line per size with moved GB/s.
    python tools/mid_size_probe.py --sizes 256,320,384,448,512
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


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps   # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="256,288,320,384,448,512")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ctx = C.Context(C.load_default_library(), 0)
    L = 2 * math.pi
    for n in [int(v) for v in args.sizes.split(",")]:
        grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
        a = torch.randn(1, n, n, n, device=dev)
        a -= a.mean()
        b = torch.randn_like(a)
        c = torch.empty_like(a)
        cells = n ** 3
        rec = {"size": n, "MB_per_array": round(4 * cells / 2 ** 20, 1)}
        t = timed(lambda: c.copy_(a), args.reps)
        rec["copy_GBs"] = round(2 * 4 * cells / t / 1e6, 1)
        t = timed(lambda: torch.add(a, b, alpha=0.5, out=c), args.reps)
        rec["triad_GBs"] = round(3 * 4 * cells / t / 1e6, 1)
        t = timed(lambda: ctx.laplace_apply(grid, 0, 1, a.data_ptr(), c.data_ptr()), args.reps)
        rec["apply_GBs"] = round(2 * 4 * cells / t / 1e6, 1)
        rec["apply_plan"] = list(ctx.query_plan(grid, False, 0).values())
        x = torch.zeros_like(a)
        ctx.cg_solve(grid, 0, 1, a.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 3, 0, 0, 0), want_info=False)   # autotune
        torch.cuda.synchronize()
        x.zero_()
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        ctx.cg_solve(grid, 0, 1, a.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, args.iters, 0, 0, 0), want_info=False)
        torch.cuda.synchronize()
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        per = {k: (v[1] / v[0] if v[0] else 0.0) for k, v in prof.items()}
        rec["matvec_GBs"] = round(3 * 4 * cells / per["cg_matvec_dot"] / 1e6, 1)
        rec["update_r_GBs"] = round(3 * 4 * cells / per["cg_update_r"] / 1e6, 1)
        rec["update_x2_GBs"] = round(5 * 4 * cells / per["cg_update"] / 1e6, 1)
        rec["plans"] = {f: list(ctx.query_plan(grid, False, f).values()) for f in (1, 2, 3)}
        print(json.dumps(rec), flush=True)
        del a, b, c, x


if __name__ == "__main__":
    main()
