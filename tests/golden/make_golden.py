#!/usr/bin/env python3
"""
Generates the golden fixtures under tests/golden/ with the NumPy oracle (oracle/phi_oracle.py).

Why the oracle and not the reference: the reference's arithmetic lives in phiml (>= 1.14.0), which is neither vendored nor
installable in this environment, so PhiFlow itself cannot produce these vectors here; the oracle restates the reference path
and is pinned against the reference's own known-answer / property tests (tests/test_oracle_reference_pins.py).

    python tests/golden/make_golden.py        # rewrites smoke_plume_64.npz, smoke_plume_mc_128.npz, taylor_green_32.npz, cavity_obstacle_16.npz
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import phi_oracle as O   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def smoke_plume(n=64, steps=5):
    """ BASELINE configs[0] at reduced size: closed box, smoke inflow sphere, buoyancy (0, 0.1), semi-Lagrangian smoke and
    velocity, projection with Solve('CG', 1e-3, x0=p) (Smoke_Plume.ipynb cell 5 with semi_lagrangian instead of mac_cormack) """
    dom = O.Domain((n, n), (0, 0), (100, 100), ((O.CLOSED, O.CLOSED),) * 2)
    cp = O.cell_positions(dom, np.float64)
    inflow = (((cp[0] - 50) ** 2 + (cp[1] - 9.5) ** 2) <= 25).astype(np.float32)[None] * np.float32(0.2)
    s_codes = ((O.OPEN, O.OPEN),) * 2      # ZERO_GRADIENT smoke
    smoke = np.zeros((1, n, n), np.float32)
    v = [np.zeros((1,) + dom.comp_shape(d), np.float32) for d in range(2)]
    p = np.zeros((1, n, n), np.float32)
    for _ in range(steps):
        smoke = O.semi_lagrangian_centered(smoke, v, 1.0, dom, s_codes) + inflow
        sp = O.pad_scalar(smoke, [(0, 0), (1, 1)], s_codes)
        buoy = (np.float32(0.5) * (sp[:, :, 1:] + sp[:, :, :-1]))[:, :, 1:n] * np.float32(0.1)
        v = O.semi_lagrangian_staggered(v, v, 1.0, dom)
        v[1] = v[1] + buoy
        v, p, info, _ = O.make_incompressible(v, dom, x0=p, rtol=1e-3, atol=0.0)
    return dict(smoke=smoke[0], vx=v[0][0], vy=v[1][0], p=p[0], iterations=info.iterations, n=n, steps=steps)


def smoke_plume_mac_cormack(n=128, steps=50, keep=(1, 10)):
    """ BASELINE configs[0]: 2-D smoke plume 128 x 128, closed box, 50 steps of Smoke_Plume.ipynb cell 5 with dt = 1:
    s = mac_cormack(s, v, dt) + inflow ; v = semi_lagrangian(v, v, dt) + resample(s * (0, 0.1), to=v) * dt ;
    v, p = make_incompressible(v, (), Solve('CG', 1e-3, x0=p)).  Snapshots after the steps in `keep` and after the last. """
    dom = O.Domain((n, n), (0, 0), (100, 100), ((O.CLOSED, O.CLOSED),) * 2)
    cp = O.cell_positions(dom, np.float64)
    inflow = (((cp[0] - 50) ** 2 + (cp[1] - 9.5) ** 2) <= 25).astype(np.float32)[None] * np.float32(0.2)
    s_codes = ((O.OPEN, O.OPEN),) * 2      # ZERO_GRADIENT smoke
    smoke = np.zeros((1, n, n), np.float32)
    v = [np.zeros((1,) + dom.comp_shape(d), np.float32) for d in range(2)]
    p = np.zeros((1, n, n), np.float32)
    out = dict(n=n, steps=steps, keep=np.asarray(keep))
    its = []
    for step in range(1, steps + 1):
        smoke = O.mac_cormack_centered(smoke, v, 1.0, dom, s_codes) + inflow
        buoy = O.centered_to_staggered(smoke, dom, s_codes, None, (0.0, 0.1))
        v = O.semi_lagrangian_staggered(v, v, 1.0, dom)
        v = [a + b * np.float32(1.0) for a, b in zip(v, buoy)]
        v, p, info, _ = O.make_incompressible(v, dom, x0=p, rtol=1e-3, atol=0.0)
        its.append(int(info.iterations[0]))
        if step in keep or step == steps:
            out.update({f"smoke_{step}": smoke[0].copy(), f"vx_{step}": v[0][0].copy(), f"vy_{step}": v[1][0].copy(), f"p_{step}": p[0].copy()})
    out["iterations"] = np.asarray(its)
    return out


def taylor_green(n=32, steps=2, iters=100):
    """ BASELINE configs[1] at 32^3: periodic [0, 2 pi]^3, fixed 100 CG iterations per step (tolerances 0) """
    L = 2 * math.pi
    dom = O.Domain((n, n, n), (0, 0, 0), (L, L, L), ((O.PERIODIC, O.PERIODIC),) * 3)
    h = L / n
    idx = np.arange(n)
    face, cent = idx * h, (idx + 0.5) * h
    u = np.broadcast_to((np.cos(face)[:, None] * np.sin(cent)[None, :])[:, :, None], (n, n, n))
    w = np.broadcast_to((-np.sin(cent)[:, None] * np.cos(face)[None, :])[:, :, None], (n, n, n))
    v = [np.ascontiguousarray(a, dtype=np.float32)[None] for a in (u, w, np.zeros((n, n, n)))]
    v0 = [a.copy() for a in v]
    p = np.zeros((1, n, n, n), np.float32)
    for _ in range(steps):
        v = O.semi_lagrangian_staggered(v, v, 0.5 * h, dom)
        v, p, info, _ = O.make_incompressible(v, dom, x0=p, rtol=0.0, atol=0.0, max_iter=iters)
    return dict(v0x=v0[0][0], v0y=v0[1][0], v0z=v0[2][0], vx=v[0][0], vy=v[1][0], vz=v[2][0], p=p[0], n=n, steps=steps, iters=iters)


def cavity_obstacle(n=16):
    """ BASELINE configs[4] flavour at 16^3 fp64: closed box, lid velocity (1,0,0) on z+, solid box obstacle, one projection """
    bcv = np.zeros((3, 2, 3)); bcv[2, 1, 0] = 1.0
    dom = O.Domain((n, n, n), (0, 0, 0), (1, 1, 1), ((O.CLOSED, O.CLOSED),) * 3, bcv)
    rng = np.random.default_rng(0)
    v = [0.05 * rng.standard_normal((1,) + dom.comp_shape(d)) for d in range(3)]
    obstacles = [O.BoxObstacle((0.375, 0.375, 0.375), (0.625, 0.625, 0.625))]
    v0 = [a.copy() for a in v]
    v = O.semi_lagrangian_staggered(v, v, 0.02, dom)
    v, p, info, rhs = O.make_incompressible(v, dom, obstacles, rtol=1e-10, atol=0.0)
    return dict(v0x=v0[0][0], v0y=v0[1][0], v0z=v0[2][0], vx=v[0][0], vy=v[1][0], vz=v[2][0], p=p[0], iterations=info.iterations, n=n)


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "smoke_plume_64.npz"), **smoke_plume())
    np.savez_compressed(os.path.join(HERE, "smoke_plume_mc_128.npz"), **smoke_plume_mac_cormack())
    np.savez_compressed(os.path.join(HERE, "taylor_green_32.npz"), **taylor_green())
    np.savez_compressed(os.path.join(HERE, "cavity_obstacle_16.npz"), **cavity_obstacle())
    print("golden fixtures written to", HERE)
