#!/usr/bin/env python3
"""
r6 (last session): is the +-3 % of a 512^3 CG iteration (profiles/r06_autotune_stability.txt: alternates from one fresh context to the next at identical launch
plans and virtual addresses) a property of the workspace ALLOCATION? K contexts are created one after the other and ALL KEPT ALIVE (every one owns its own r, d0, d1:
no page is handed from one to the next), the launch plans of the first are pinned in all of them, the caller's x / rhs are shared; every context is timed in three
round-robin rounds. Stable per-context times that differ between contexts = the placement decides, and a library can choose between candidates it holds.
    python tools/micro/ws_placement_probe.py [contexts] [sizes ...]
"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phiflow_amd import _capi as C   # noqa: E402


def time_solve(ctx, grid, rhs, x, iters):
    x.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, iters, 0, 0, 0), want_info=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    sizes = [int(a) for a in sys.argv[2:]] or [512, 256]
    dev = torch.device("cuda:0")
    lib = C.load_default_library()
    L = 2 * math.pi
    for n in sizes:
        grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
        rhs = torch.randn(1, n, n, n, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
        rhs -= rhs.mean()
        x = torch.zeros_like(rhs)
        first = C.Context(lib, 0)
        first.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 10, 0, 0, 0), want_info=False)     # tunes
        torch.cuda.synchronize()
        plans = {f: first.query_plan(grid, False, f) for f in (0, 1, 2, 3)}
        ctxs = [first]
        for _ in range(k - 1):
            c = C.Context(lib, 0)
            c.set_autotune(False)
            for f, q in plans.items():
                c.set_tuning_kernel(f, int(q["rows"]), int(q["tpr"]), int(q["chunk"]))
            c.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 4, 0, 0, 0), want_info=False)        # allocates its workspace
            ctxs.append(c)
        torch.cuda.synchronize()
        iters = 60 if n >= 384 else 200
        rounds = [[round(time_solve(c, grid, rhs, x, iters), 5) for c in ctxs] for _ in range(3)]
        print(json.dumps({"size": n, "contexts_alive": k, "plans": {str(f): [q["rows"], q["tpr"], q["chunk"]] for f, q in plans.items()},
                          "ms_per_iteration_rounds": rounds, "build": lib.build_id()}), flush=True)
        del ctxs, first


if __name__ == "__main__":
    main()
