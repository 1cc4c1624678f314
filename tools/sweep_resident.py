#!/usr/bin/env python3
"""
The launch-per-iteration CG forms (two launches, or the single-reduction MODE_CG1 launch -- whatever the library picks by default) against the
RESIDENT solver (cg_resident.hip: the whole solve in one launch, vectors in registers, one barrier among an entry's workgroups per
iteration): wall time per iteration of a fixed-iteration solve and the time / iterations of a tolerance-mode solve, one JSON line per grid.
    python tools/sweep_resident.py [iterations]
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

CASES = [((512, 512), 1), ((512, 512), 8), ((256, 256), 8), ((256, 256), 16), ((384, 384), 4), ((512, 256), 8), ((128, 128), 16), ((256, 512), 2), ((192, 192), 1)]


def main():
    dev = torch.device("cuda:0")
    alt = os.environ.get("PHIHIP_SWEEP_LIB", "")          # another build of the library (A/B of experiments)
    lib = C.Library(alt, strict=False) if alt else C.load_default_library()
    ctx = C.Context(lib, 0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    cases = CASES if not os.environ.get("PHIHIP_SWEEP_SHORT") else [((512, 512), 1), ((512, 512), 8), ((256, 256), 16), ((384, 384), 4)]
    if os.environ.get("PHIHIP_SWEEP_BATCHES"):      # r6: BASELINE configs[3]'s sharding story in numbers -- B entries of 512^2 on ONE GPU (B = 1 is what each GPU holds when
        cases = [((512, 512), b) for b in (1, 2, 4, 8, 16, 32, 64)]      # 8 x 512^2 are sharded over 8 GPUs; "resident" = mode 2: falls back to the launch forms where B x 32 workgroups > CUs)
    if os.environ.get("PHIHIP_SWEEP_CASES"):        # "128x128x1,256x256x2": explicit (rows x columns x batch) list
        cases = [((int(a), int(b)), int(c)) for a, b, c in (t.split("x") for t in os.environ["PHIHIP_SWEEP_CASES"].split(","))]
    for res, batch in cases:
        for bc_name, bc in (("closed", C.BC_CLOSED), ("periodic", C.BC_PERIODIC))[: (1 if os.environ.get("PHIHIP_SWEEP_SHORT") or os.environ.get("PHIHIP_SWEEP_BATCHES") else 2)]:
            D = len(res)
            grid = C.make_grid(D, C.PHIHIP_F32, batch, res, (0.0,) * D, tuple(float(n) for n in res), ((bc, bc),) * D)
            rhs = torch.randn(batch, *res, generator=torch.Generator(device=dev).manual_seed(0), device=dev, dtype=torch.float32)
            rhs -= rhs.mean(dim=tuple(range(1, D + 1)), keepdim=True)
            x = torch.zeros_like(rhs)
            fptr = 0
            if os.environ.get("PHIHIP_SWEEP_FLAGS"):      # r6: the same solves WITH cell flags -- a solid disc in the middle of every entry (the resident solver's FLAGS form)
                yy, xx = torch.meshgrid(torch.arange(res[0], device=dev) + 0.5, torch.arange(res[1], device=dev) + 0.5, indexing="ij")
                acc = (((yy - res[0] / 2) ** 2 + (xx - res[1] / 2) ** 2) > (min(res) / 6.0) ** 2).to(torch.uint8).contiguous()
                flags = torch.zeros(res, dtype=torch.uint8, device=dev)
                g1 = C.make_grid(D, C.PHIHIP_F32, 1, res, (0.0,) * D, tuple(float(n) for n in res), ((bc, bc),) * D)
                ctx.build_cellflags(g1, acc.data_ptr(), 0, 1, flags.data_ptr())
                rhs = rhs * acc
                rhs -= (rhs.sum(dim=tuple(range(1, D + 1)), keepdim=True) / acc.sum()) * acc
                fptr = flags.data_ptr()
            rec = {"res": list(res), "batch": batch, "bc": bc_name, "cell_flags": bool(fptr), "cells_x_batch": batch * int(torch.tensor(res).prod()), "build": lib.build_id()}
            sols = {}
            for label, mode in (("launches", 0), ("resident", 2)):
                ctx.set_resident_cg(mode)
                ctx.cg_solve(grid, fptr, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 5, 50, 0, 0), want_info=False)   # plans, workspace
                best = 1e30
                for _ in range(3):
                    x.zero_()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    info = ctx.cg_solve(grid, fptr, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, iters, 50, 0, 0), want_info=True)
                    best = min(best, time.perf_counter() - t0)
                rec[label] = {"us_per_iteration": round(best / iters * 1e6, 3), "rel_residual": math.sqrt(info[0].residual_sq / info[0].rhs_sq),
                              "iterations": info[0].iterations}
                # tolerance mode: 1e-5, host polling every 10 iterations for the launch forms, none for the resident kernel
                tb = 1e30
                for _ in range(3):
                    x.zero_()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    info = ctx.cg_solve(grid, fptr, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(1e-5, 0.0, 4000, 50, 10, 0), want_info=True)
                    tb = min(tb, time.perf_counter() - t0)
                rec[label]["tolerance_solve"] = {"ms": round(tb * 1e3, 4), "iterations": [i.iterations for i in info][:4], "converged": all(i.converged for i in info)}
                sols[label] = x.clone()
            d = (sols["resident"] - sols["launches"])
            d = d - d.mean(dim=tuple(range(1, D + 1)), keepdim=True)
            ref = sols["launches"] - sols["launches"].mean(dim=tuple(range(1, D + 1)), keepdim=True)
            rec["rel_l2_resident_vs_launches"] = float(d.norm() / ref.norm())
            rec["speedup_resident"] = round(rec["launches"]["us_per_iteration"] / rec["resident"]["us_per_iteration"], 3)
            print(json.dumps(rec), flush=True)
            del rhs, x
    ctx.set_resident_cg(1)      # (the library's default since r6)


if __name__ == "__main__":
    main()
