#!/usr/bin/env python3
"""
Two launches per CG iteration (MATVEC + UPDATE, 7 words per cell) against the single-reduction form (one fused launch, 10 words per
cell; stencil_march.hpp MODE_CG1): wall time per iteration of a fixed-iteration solve, one JSON line per grid.
    python tools/sweep_cg1.py            # default list: batched 2-D and small / mid 3-D grids
"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C   # noqa: E402

CASES = [((512, 512), 1), ((512, 512), 8), ((1024, 1024), 1), ((1024, 1024), 8), ((2048, 2048), 1), ((256, 256), 8), ((256, 256), 64),
         ((64, 64, 64), 1), ((96, 96, 96), 1), ((128, 128, 128), 1), ((160, 160, 160), 1), ((192, 192, 192), 1), ((256, 256, 256), 1), ((64, 64, 64), 16)]


def main():
    dev = torch.device("cuda:0")
    ctx = C.Context(C.load_default_library(), 0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for dtype, code, esize in ((torch.float32, C.PHIHIP_F32, 4),):
        for res, batch in CASES:
            D = len(res)
            grid = C.make_grid(D, code, batch, res, (0.0,) * D, tuple(float(n) for n in res), ((C.BC_CLOSED, C.BC_CLOSED),) * D)
            rhs = torch.randn(batch, *res, generator=torch.Generator(device=dev).manual_seed(0), device=dev, dtype=dtype)
            rhs -= rhs.mean(dim=tuple(range(1, D + 1)), keepdim=True)
            x = torch.zeros_like(rhs)
            rec = {"res": list(res), "batch": batch, "cells_x_batch": batch * int(torch.tensor(res).prod())}
            for label, mode in (("two_launch", 0), ("single_reduction", 2)):
                ctx.set_single_reduction_cg(mode)
                ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 5, 50, 0, 0), want_info=False)   # plans, workspace
                best = 1e30
                for _ in range(3):
                    x.zero_()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    info = ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, iters, 50, 0, 0), want_info=True)
                    best = min(best, time.perf_counter() - t0)
                rec[label] = {"us_per_iteration": round(best / iters * 1e6, 3), "rel_residual": math.sqrt(info[0].residual_sq / info[0].rhs_sq)}
            rec["speedup_single_reduction"] = round(rec["two_launch"]["us_per_iteration"] / rec["single_reduction"]["us_per_iteration"], 3)
            print(json.dumps(rec), flush=True)
            del rhs, x
    ctx.set_single_reduction_cg(0)


if __name__ == "__main__":
    main()
