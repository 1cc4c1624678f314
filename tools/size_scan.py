#!/usr/bin/env python3
"""
CG throughput versus problem size (does the 256 MiB Infinity Cache carry the working set?): times the CG iteration at a range
of cubic sizes with the default tile plan and prints one JSON line per size.
    python tools/size_scan.py --sizes 128,160,192,224,256,288,320,384,448,512
An entry may be a box `n0xn1xn2` (x slow ... z fast): the ROW-PITCH experiment of DESIGN.md §8 -- does the rate of the 288^3 ... 448^3 sizes
change when only the row length (the phase of the rows against the HBM channel interleave) or only the number of rows / planes changes?
    python tools/size_scan.py --sizes 288,288x288x320,288x320x288,320x288x288,320
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
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--lib", default="", help="alternative libphihip build to load (A/B comparisons)")
    ap.add_argument("--bc", type=int, default=0, help="boundary code of every side: 0 periodic, 1 closed")
    ap.add_argument("--flags", type=int, default=0, help="1: solve with a (fully accessible) cell-flag array -- prices the flag path of the kernels")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ctx = C.Context(C.Library(args.lib, strict=False) if args.lib else C.load_default_library(), 0)
    L = 2 * math.pi
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    esize = 8 if args.dtype == "f64" else 4
    for entry in args.sizes.split(","):
        shape = tuple(int(v) for v in entry.split("x")) if "x" in entry else (int(entry),) * 3
        n = entry if "x" in entry else int(entry)
        grid = C.make_grid(3, C.PHIHIP_F64 if args.dtype == "f64" else C.PHIHIP_F32, 1, shape, (0, 0, 0), tuple(L * k / shape[0] for k in shape), ((args.bc, args.bc),) * 3)
        rhs = torch.randn(1, *shape, generator=torch.Generator(device=dev).manual_seed(0), device=dev, dtype=tdt)   # on the device: 1024^3 = 4 GiB
        rhs -= rhs.mean()
        x = torch.zeros_like(rhs)
        fl = 0
        if args.flags:
            acc = torch.ones(*shape, dtype=torch.uint8, device=dev)
            flags = torch.empty(*shape, dtype=torch.uint8, device=dev)
            ctx.build_cellflags(grid, acc.data_ptr(), 0, 1, flags.data_ptr())
            fl = flags.data_ptr()
        solve = C.Solve(0.0, 0.0, args.iters, 0, 0, 0)
        rec = {"lib": os.path.basename(args.lib) if args.lib else "default", "size": n, "dtype": args.dtype, "bc": args.bc, "flags": args.flags, "working_set_MB": round(4 * esize * math.prod(shape) / 2 ** 20, 1)}
        for label, tune in (("model", False), ("tuned", True)):          # analytic plan, then the first-call autotune (cg.hip)
            if not hasattr(ctx.lib.dll, "phihip_set_autotune"):
                if tune:
                    continue
            else:
                ctx.set_autotune(tune)
            ctx.cg_solve(grid, fl, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 3, 0, 0, 0), want_info=False)
            torch.cuda.synchronize()
            x.zero_()
            ctx.profile_enable(True)
            ctx.profile_read(reset=True)
            ctx.cg_solve(grid, fl, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False)
            torch.cuda.synchronize()
            prof = ctx.profile_read(reset=True)
            ctx.profile_enable(False)
            per = {k: (v[1] / v[0] if v[0] else 0.0) for k, v in prof.items()}
            mv, x2, ur = per["cg_matvec_dot"], per["cg_update"], per.get("cg_update_r", 0.0)
            it = mv + 0.5 * (x2 + (ur or x2))
            cells = math.prod(shape)
            plans = {f: ctx.query_plan(grid, bool(args.flags), f) for f in (1, 2, 3)}
            rec[label] = {"plan_mv": list(plans[1].values()), "plan_x2": list(plans[2].values()), "plan_ur": list(plans[3].values()),
                          "us_matvec": round(mv * 1e3, 2), "us_update_x2": round(x2 * 1e3, 2), "us_update_r": round(ur * 1e3, 2), "us_iteration": round(it * 1e3, 2),
                          "moved_GBs_matvec": round(3 * esize * cells / mv / 1e6, 1), "moved_GBs_update_x2": round(5 * esize * cells / x2 / 1e6, 1),
                          "moved_GBs_update_r": round(3 * esize * cells / ur / 1e6, 1) if ur else None, "moved_GBs_iteration": round(7 * esize * cells / it / 1e6, 1)}
        print(json.dumps(rec), flush=True)
        del rhs, x


if __name__ == "__main__":
    main()
