#!/usr/bin/env python3
"""
r6: does the first-call autotune of the CG marching kernels settle on the same launch plans in every fresh context? One JSON line per (size, trial): the plans
(rows, threads per row, planes per workgroup) of MATVEC / UPDATE_X2 / UPDATE_R and the wall time of a CG iteration with them.
    python tools/autotune_stability.py [trials] [sizes ...]
"""
import json
import math
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from phiflow_amd import _capi as C   # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    sizes = [int(a) for a in sys.argv[2:]] or [384, 512]
    dev = torch.device("cuda:0")
    lib = C.load_default_library()
    L = 2 * math.pi
    for n in sizes:
        grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
        rhs = torch.randn(1, n, n, n, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
        rhs -= rhs.mean()
        x = torch.zeros_like(rhs)
        for t in range(trials):
            ctx = C.Context(lib, 0)                       # a fresh context: nothing tuned
            ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 10, 0, 0, 0), want_info=False)
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(3):
                x.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 60, 0, 0, 0), want_info=False)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 60)
            plans = {name: [q[k] for k in ("rows", "tpr", "chunk")] for name, q in ((nm, ctx.query_plan(grid, False, f)) for nm, f in (("matvec", 1), ("update_x2", 2), ("update_r", 3)))}
            print(json.dumps({"size": n, "trial": t, "plans": plans, "ms_per_iteration": round(best, 5), "build": lib.build_id()}), flush=True)
            del ctx


if __name__ == "__main__":
    main()
