#!/usr/bin/env python3
"""
Wall time per step of the phi-level mirror API (Smoke_Plume.ipynb cell 5: mac_cormack smoke + buoyancy + semi-Lagrangian velocity +
make_incompressible) against the summed kernel time of the same step -- how much of a small simulation is Python / ctypes glue.
    python tools/time_host_api.py --size 128 --batch 1
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd.flow import *   # noqa: E402,F401,F403
from phiflow_amd.flow import default_backend, resample   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    n, B = args.size, args.batch
    be = default_backend()
    domain = Box(x=100, y=100)
    xs = [30 + 40 * b / max(1, B - 1) for b in range(B)] if B > 1 else 50
    inflow = 0.2 * resample(Sphere(x=xs, y=9.5, radius=5), to=CenteredGrid(0, ZERO_GRADIENT, domain, x=n, y=n), soft=True)
    v = StaggeredGrid(0, 0, domain, x=n, y=n, batch=B if B > 1 else None)
    s = CenteredGrid(0, ZERO_GRADIENT, domain, x=n, y=n, batch=B if B > 1 else None)
    p = None

    def step(v, s, p):
        s = advect.mac_cormack(s, v, 1.0) + inflow
        v = advect.semi_lagrangian(v, v, 1.0) + resample(s * (0, 0.1), to=v)
        v, p = fluid.make_incompressible(v, (), Solve('CG', 1e-3, x0=p))
        return v, s, p
    for _ in range(60):      # one-time costs (kernel code loading, allocator growth) amount to ~50 ms
        v, s, p = step(v, s, p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = 0
    for _ in range(args.steps):
        v, s, p = step(v, s, p)
        its += max(p.solve_info.iterations)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps
    be.ctx.profile_enable(True)
    be.ctx.profile_read(reset=True)
    for _ in range(5):
        v, s, p = step(v, s, p)
    torch.cuda.synchronize()
    prof = be.ctx.profile_read(reset=True)
    be.ctx.profile_enable(False)
    kernel_ms = sum(t for _, t in prof.values()) / 5
    print(json.dumps({"size": n, "batch": B, "ms_per_step_wall": round(wall * 1e3, 3), "ms_per_step_kernels": round(kernel_ms, 3),
                      "cg_iterations_per_step": round(its / args.steps, 1), "launches_per_step": sum(c for c, _ in prof.values()) / 5}))


if __name__ == "__main__":
    main()
