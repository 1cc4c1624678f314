#!/usr/bin/env python3
"""
r6: a CG iteration with PINNED launch plans, candidates alternating in one process (the autotune's loop against the solve it tunes for).
    python tools/micro/plan_ab.py SIZE FAMILY "r,t,c" "r,t,c" ...      (FAMILY: 1 MATVEC, 2 UPDATE_X2, 3 UPDATE_R; the other families keep the autotuned plan)
"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phiflow_amd import _capi as C   # noqa: E402


def main():
    n, fam = int(sys.argv[1]), int(sys.argv[2])
    plans = [tuple(int(q) for q in a.split(",")) for a in sys.argv[3:]]
    dev = torch.device("cuda:0")
    lib = C.load_default_library()
    L = 2 * math.pi
    grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
    rhs = torch.randn(1, n, n, n, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
    rhs -= rhs.mean()
    x = torch.zeros_like(rhs)
    ctx = C.Context(lib, 0)
    ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 10, 0, 0, 0), want_info=False)      # autotune
    tuned = {f: ctx.query_plan(grid, False, f) for f in (1, 2, 3)}
    print(json.dumps({"size": n, "autotuned": {f: [q[k] for k in ("rows", "tpr", "chunk")] for f, q in tuned.items()}}), flush=True)
    for rnd in range(3):
        for pl in plans:
            for f in (1, 2, 3):
                q = tuned[f]
                ctx.set_tuning_kernel(f, *(pl if f == fam else (q["rows"], q["tpr"], q["chunk"])))
            best = 1e30
            for _ in range(3):
                x.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 60, 0, 0, 0), want_info=False)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 60)
            print(json.dumps({"size": n, "family": fam, "plan": pl, "round": rnd, "ms_per_iteration": round(best, 5)}), flush=True)


if __name__ == "__main__":
    main()
