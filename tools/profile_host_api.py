#!/usr/bin/env python3
""" cProfile of the phi-level mirror API on a small plume step (where the host side dominates): python tools/profile_host_api.py --size 64 """
import argparse
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd.flow import *   # noqa: E402,F401,F403
from phiflow_amd.flow import resample   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    n = args.size
    domain = Box(x=100, y=100)
    inflow = 0.2 * resample(Sphere(x=50, y=9.5, radius=5), to=CenteredGrid(0, ZERO_GRADIENT, domain, x=n, y=n), soft=True)
    v = StaggeredGrid(0, 0, domain, x=n, y=n)
    s = CenteredGrid(0, ZERO_GRADIENT, domain, x=n, y=n)
    p = None

    def step(v, s, p):
        s = advect.mac_cormack(s, v, 1.0) + inflow
        v = advect.semi_lagrangian(v, v, 1.0) + resample(s * (0, 0.1), to=v)
        v, p = fluid.make_incompressible(v, (), Solve('CG', 1e-3, x0=p))
        return v, s, p
    for _ in range(10):
        v, s, p = step(v, s, p)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        v, s, p = step(v, s, p)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
