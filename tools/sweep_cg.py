#!/usr/bin/env python3
"""
Tile-configuration sweep of the CG marching kernels on a MI355X: times MATVEC/UPDATE per launch (hipEvent pairs) for
every (rows_per_thread, threads_per_row, chunk) and prints one JSON line per configuration.
    python tools/sweep_cg.py --size 256 --iters 20 > gpurun_out/sweep_256.jsonl
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--method", type=int, default=0, help="phihip_method: 0 = CG, 1 = CG-adaptive")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--configs", default="")
    ap.add_argument("--lib", default="", help="alternative libphihip build to load (A/B comparisons)")
    ap.add_argument("--obstacle", type=int, default=0, help="1: closed box with a solid box obstacle in the middle (cell flags, BASELINE config 5)")
    ap.add_argument("--defer", type=int, default=1, help="0: update x in every iteration (phihip_set_deferred_x_update)")
    ap.add_argument("--family", type=int, default=-1, help="-1: all kernels share the configuration; 1 = MATVEC only, 2 = UPDATE only")
    args = ap.parse_args()
    n = args.size
    dev = torch.device("cuda:0")
    lib = C.Library(args.lib, strict=False) if args.lib else C.load_default_library()
    ctx = C.Context(lib, 0)
    ctx.set_deferred_x_update(bool(args.defer))
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    esize = 8 if args.dtype == "f64" else 4
    L = 2 * math.pi
    grid = C.make_grid(3, C.PHIHIP_F64 if args.dtype == "f64" else C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L),
                       ((1, 1),) * 3 if args.obstacle else ((0, 0),) * 3)
    flags_ptr = 0
    if args.obstacle:
        import numpy as np
        inside = (np.arange(n) >= 3 * n // 8) & (np.arange(n) < 5 * n // 8)
        acc = ~(inside[:, None, None] & inside[None, :, None] & inside[None, None, :])
        acc_t = torch.from_numpy(acc.astype(np.uint8)).to(dev)
        flags_t = torch.empty(n, n, n, dtype=torch.uint8, device=dev)
        ctx.build_cellflags(grid, acc_t.data_ptr(), 0, 1, flags_t.data_ptr())
        flags_ptr = flags_t.data_ptr()
    g = torch.Generator(device="cpu").manual_seed(0)
    rhs = torch.randn(1, n, n, n, generator=g, dtype=tdt)
    rhs -= rhs.mean()
    rhs = rhs.to(dev)
    x = torch.zeros_like(rhs)
    solve = C.Solve(0.0, 0.0, args.iters, 0, 0, args.method)
    configs = [(r, t, c) for (r, t) in [(1, 16), (2, 16), (2, 32), (4, 32), (4, 64), (1, 64), (2, 64), (1, 32)] for c in (8, 16, 32, 64, n)]
    if args.configs:
        configs = [tuple(int(v) for v in item.split(",")) for item in args.configs.split(";")]
    configs = [(0, 0, 0)] + configs
    cells = n ** 3
    for rows, tpr, chunk in configs:
        if args.family < 0:
            ctx.set_tuning(rows, tpr, chunk)
        else:
            ctx.set_tuning(0, 0, 0)
            ctx.set_tuning_kernel(args.family, rows, tpr, chunk)
        x.zero_()
        ctx.cg_solve(grid, flags_ptr, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 3, 0, 0, args.method), want_info=False)   # warm-up
        torch.cuda.synchronize()
        x.zero_()
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        ctx.cg_solve(grid, flags_ptr, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False)
        torch.cuda.synchronize()
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        # un-profiled wall time of the same solve
        x.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.cg_solve(grid, flags_ptr, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False)
        e1.record()
        torch.cuda.synchronize()
        wall_ms = e0.elapsed_time(e1)
        mv = prof["cg_matvec_dot"][1] / max(1, prof["cg_matvec_dot"][0])
        up = prof["cg_update"][1] / max(1, prof["cg_update"][0])
        sc = prof["cg_scalar"][1] / max(1, prof["cg_scalar"][0])
        ur = prof.get("cg_update_r", (0, 0.0))
        ur = ur[1] / max(1, ur[0])
        words = esize
        plans = {f: ctx.query_plan(grid, bool(args.obstacle), f) for f in (1, 2, 3)}
        out = {"lib": os.path.basename(args.lib) if args.lib else "default", "size": n, "dtype": args.dtype, "family": args.family, "defer": args.defer, "plan_mv": list(plans[1].values()), "plan_up": list(plans[2].values()), "plan_ur": list(plans[3].values()), "rows": rows, "tpr": tpr, "chunk": chunk,
               "ms_matvec": round(mv, 5), "ms_update": round(up, 5), "ms_update_r": round(ur, 5), "ms_scalar": round(sc, 5),
               "ms_iter_events": round(mv + up + 2 * sc, 5), "ms_iter_wall": round(wall_ms / args.iters, 5),
               "alg_GBs_iter_wall": round(10 * words * cells / (wall_ms / args.iters * 1e-3) / 1e9, 1),
               "alg_GBs_update": round(6 * words * cells / (up * 1e-3) / 1e9, 1),
               "alg_GBs_matvec": round(4 * words * cells / (mv * 1e-3) / 1e9, 1),
               "actual_GBs_update": round(5 * words * cells / (up * 1e-3) / 1e9, 1),
               "actual_GBs_matvec": round(3 * words * cells / (mv * 1e-3) / 1e9, 1)}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
