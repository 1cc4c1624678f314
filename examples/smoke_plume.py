#!/usr/bin/env python3
"""
The reference's Smoke_Plume notebook (docs/Smoke_Plume.ipynb cell 5: MacCormack smoke, buoyancy, semi-Lagrangian velocity, projection)
with the import line changed -- `from phiflow_amd.flow import *` instead of `from phi.torch.flow import *`. Needs an MI355X.
    python examples/smoke_plume.py [--size 128] [--steps 50] [--jit]
--jit: the step as the notebook writes it, `@jit_compile def step(v, s, p, dt)` -- captured in a hipGraph at the first call of a signature and replayed
afterwards (phiflow_amd/jit.py). Inside a captured function the host is not told how a solve went: give the tolerance solve a launch budget that fits.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phiflow_amd.flow import *   # noqa: E402,F401,F403


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--jit", action="store_true")
    args = ap.parse_args()
    n = args.size
    domain = Box(x=100, y=100)
    velocity = StaggeredGrid(0, 0, domain, x=n, y=n)                       # closed box
    smoke = CenteredGrid(0, ZERO_GRADIENT, domain, x=n, y=n)
    inflow = 0.2 * CenteredGrid(Sphere(x=50, y=9.5, radius=5), ZERO_GRADIENT, domain, x=n, y=n)
    pressure = None
    dt = 1.0

    def step(velocity, smoke, pressure, dt=1.0):
        smoke = advect.mac_cormack(smoke, velocity, dt) + inflow
        buoyancy = resample(smoke * (0, 0.1), to=velocity)
        velocity = advect.semi_lagrangian(velocity, velocity, dt) + buoyancy * dt
        velocity, pressure = fluid.make_incompressible(velocity, (), Solve('CG', 1e-3, x0=pressure, max_iterations=200 if args.jit else 1000))
        return velocity, smoke, pressure
    if args.jit:
        step = jit_compile(step)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        velocity, smoke, pressure = step(velocity, smoke, pressure, dt)
    s = smoke.numpy()                                                       # (synchronises)
    wall = time.perf_counter() - t0
    print(f"{args.steps} steps of {n} x {n}: {wall / args.steps * 1e3:.2f} ms per step, smoke mass {float(s.sum()):.3f}, "
          f"max |v_y| {float(abs(velocity.numpy()[1]).max()):.4f}, last solve: "
          + (f"{int(pressure.solve_info.iterations[0])} CG iterations" if pressure.solve_info is not None else f"not reported (captured: {step.traces} captures, {step.replays} replays)"))


if __name__ == "__main__":
    main()
