#!/usr/bin/env python3
"""
The benchmark's physics through the phi-level API: a periodic 3-D Taylor-Green vortex advanced by advect.semi_lagrangian +
fluid.make_incompressible (docs/Taylor_Green.ipynb cell 12 in 3-D). Needs an MI355X.
    python examples/taylor_green_3d.py [--size 128] [--steps 20]
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phiflow_amd.flow import *   # noqa: E402,F401,F403
from phiflow_amd.flow import default_backend   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--fp64", action="store_true", help="double precision with Solve('CG', 1e-10) like the reference notebook (1e-12 there)")
    args = ap.parse_args()
    with precision(64 if args.fp64 else 32):
        run(args)


def run(args):
    n, L = args.size, 2 * math.pi
    be = default_backend()
    h = L / n
    tdt = torch.float64 if args.fp64 else torch.float32
    face = torch.arange(n, dtype=tdt) * h
    cent = (torch.arange(n, dtype=tdt) + 0.5) * h
    u = (torch.cos(face)[:, None, None] * torch.sin(cent)[None, :, None]).expand(n, n, n)
    w = (-torch.sin(cent)[:, None, None] * torch.cos(face)[None, :, None]).expand(n, n, n)
    comps = [t.contiguous()[None].to(be.device) for t in (u, w, torch.zeros(n, n, n, dtype=tdt))]
    v = StaggeredGrid(comps, PERIODIC, Box(x=L, y=L, z=L), x=n, y=n, z=n)
    p = None
    dt = 0.5 * h
    energy = lambda f: sum(float((c.astype('float64') ** 2).sum()) for c in f.numpy()) * h ** 3 / 2
    e0 = energy(v)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        v = advect.semi_lagrangian(v, v, dt)
        # (fp32 CG on a periodic 128^3+ box stagnates near a relative residual of 1e-4: condition number x machine epsilon)
        v, p = fluid.make_incompressible(v, (), Solve('CG', 1e-10 if args.fp64 else 1e-3, x0=p))
    e1 = energy(v)
    wall = time.perf_counter() - t0
    print(f"{args.steps} steps of {n}^3: {wall / args.steps * 1e3:.2f} ms per step, kinetic energy {e0:.4f} -> {e1:.4f}, "
          f"last solve: {int(p.solve_info.iterations[0])} CG iterations")


if __name__ == "__main__":
    main()
