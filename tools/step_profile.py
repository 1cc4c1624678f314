#!/usr/bin/env python3
""" per-kernel time of the benchmark step (advect + projection, 100 fixed CG iterations, 256^3) with a given build of the library:
    python tools/step_profile.py [--lib other.so] [--size 256] """
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
    ap.add_argument("--lib", default="")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    n, dev, L = args.size, torch.device("cuda:0"), 2 * math.pi
    ctx = C.Context(C.Library(args.lib, strict=False) if args.lib else C.load_default_library(), 0)
    grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
    h = L / n
    face, cent = torch.arange(n) * h, (torch.arange(n) + 0.5) * h
    u = (torch.cos(face)[:, None, None] * torch.sin(cent)[None, :, None]).expand(n, n, n)
    w = (-torch.sin(cent)[:, None, None] * torch.cos(face)[None, :, None]).expand(n, n, n)
    v = [t.contiguous()[None].float().to(dev) for t in (u, w, torch.zeros(n, n, n))]
    v2 = [torch.empty_like(t) for t in v]
    p = torch.zeros(1, n, n, n, device=dev)
    solve = C.Solve(0.0, 0.0, 100, 50, 0, 0)
    P = lambda ts: [t.data_ptr() for t in ts]

    def step():
        nonlocal v, v2
        ctx.advect_staggered(grid, P(v), P(v), P(v2), 0.5 * h)
        ctx.make_incompressible(grid, P(v2), None, 0, 1, True, p.data_ptr(), 0, solve, want_info=False)
        v, v2 = v2, v

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    print(json.dumps({"lib": os.path.basename(args.lib) or "default", "size": n, "ms_per_step": round(ms, 4),
                      "kernel_ms_per_step": {k: round(val[1] / 3, 5) for k, val in prof.items()}}))


if __name__ == "__main__":
    main()
