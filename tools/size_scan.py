#!/usr/bin/env python3
"""
CG throughput versus problem size (does the 256 MiB Infinity Cache carry the working set?): times the CG iteration at a range
of cubic sizes with the default tile plan and prints one JSON line per size.
    python tools/size_scan.py --sizes 128,160,192,224,256,288,320,384,448,512
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
    ap.add_argument("--sizes", default="128,160,192,224,256,288,320,384,448,512")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--lib", default="", help="alternative libphihip build to load (A/B comparisons)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ctx = C.Context(C.Library(args.lib, strict=False) if args.lib else C.load_default_library(), 0)
    L = 2 * math.pi
    for n in [int(v) for v in args.sizes.split(",")]:
        grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
        rhs = torch.randn(1, n, n, n, generator=torch.Generator(device=dev).manual_seed(0), device=dev)   # on the device: 1024^3 = 4 GiB
        rhs -= rhs.mean()
        x = torch.zeros_like(rhs)
        solve = C.Solve(0.0, 0.0, args.iters, 0, 0, 0)
        ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 3, 0, 0, 0), want_info=False)
        torch.cuda.synchronize()
        x.zero_()
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False)
        torch.cuda.synchronize()
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        mv = prof["cg_matvec_dot"][1] / max(1, prof["cg_matvec_dot"][0])
        up = prof["cg_update"][1] / max(1, prof["cg_update"][0])
        cells = n ** 3
        plans = {f: ctx.query_plan(grid, False, f) for f in (1, 2, 3)}
        print(json.dumps({"lib": os.path.basename(args.lib) if args.lib else "default", "size": n, "plan_mv": list(plans[1].values()), "plan_up": list(plans[2].values()), "plan_ur": list(plans[3].values()), "working_set_MB": round(4 * 4 * cells / 2 ** 20, 1), "ms_matvec": round(mv, 5), "ms_update": round(up, 5),
                          "actual_GBs_matvec": round(12 * cells / mv / 1e6, 1), "actual_GBs_update": round(16 * cells / up / 1e6, 1),   # mean of the r-only (3 words) and the paired (5 words) form
                          "alg_GBs_iter": round(40 * cells / (mv + up) / 1e6, 1)}), flush=True)
        del rhs, x


if __name__ == "__main__":
    main()
