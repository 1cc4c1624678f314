#!/usr/bin/env python3
"""
Decaying 2-D Taylor-Green vortex with viscosity: operator-split step  diffuse.implicit -> advect.semi_lagrangian -> make_incompressible
(the step of docs/Taylor_Green.ipynb cell 12 with the explicit diffusion replaced by the implicit one, examples/grids/Burgers.ipynb cell 2:
stable at any viscosity * dt / dx^2). The exact solution decays like exp(-2 nu t): the script prints the measured decay next to it.
Both linear solves -- the Poisson problem of the projection and (I - nu dt laplace) of the diffusion -- run on the same matrix-free CG
kernels. Needs an MI355X.
    python examples/viscous_taylor_green.py [--size 128] [--steps 50] [--viscosity 0.05]
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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--viscosity", type=float, default=0.05)
    args = ap.parse_args()
    n, L = args.size, 2 * math.pi
    be = default_backend()
    h = L / n
    face = torch.arange(n, dtype=torch.float32) * h
    cent = (torch.arange(n, dtype=torch.float32) + 0.5) * h
    u = (torch.cos(face)[:, None] * torch.sin(cent)[None, :])
    w = (-torch.sin(cent)[:, None] * torch.cos(face)[None, :])
    v = StaggeredGrid([t.contiguous()[None].to(be.device) for t in (u, w)], PERIODIC, Box(x=L, y=L), x=n, y=n)
    p = None
    dt = 0.5 * h
    energy = lambda f: sum(float((c.astype('float64') ** 2).sum()) for c in f.numpy()) * h ** 2 / 2
    e0 = energy(v)
    t0 = time.perf_counter()
    its = 0
    for _ in range(args.steps):
        v = diffuse.implicit(v, args.viscosity, dt, Solve('CG', 1e-5, 0))
        its += max(v.solve_info.iterations)
        v = advect.semi_lagrangian(v, v, dt)
        v, p = fluid.make_incompressible(v, (), Solve('CG', 1e-4, x0=p))
    wall = time.perf_counter() - t0
    e1 = energy(v)
    t_end = args.steps * dt
    print(f"{args.steps} steps of {n}^2: {wall / args.steps * 1e3:.2f} ms per step, kinetic energy {e0:.4f} -> {e1:.4f} "
          f"(ratio {e1 / e0:.4f}, exact viscous decay exp(-4 nu t) = {math.exp(-4 * args.viscosity * t_end):.4f}), "
          f"{its / args.steps:.1f} CG iterations per implicit diffusion step, nu dt / dx^2 = {args.viscosity * dt / h ** 2:.2f}")


if __name__ == "__main__":
    main()
