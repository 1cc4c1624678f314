#!/usr/bin/env python3
"""
Does any operator of the phi-level plume step leave part of a result unwritten, or read a result buffer before writing it? (round 5: the captured and the eager
step differed in the last bits depending on the ALLOCATION pattern of the surrounding code.) Every `empty` buffer the Python layer hands to the library is filled
with NaN here; a NaN in any operator's result names the operator.     python tools/micro/nan_fill_check.py [halo] [n] [steps]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from phiflow_amd.backend import HipBackend            # noqa: E402
from phiflow_amd.flow import *                        # noqa: E402,F401,F403

halo = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
_empty_like, _empty = torch.empty_like, torch.empty


def nan_empty_like(t, **kw):
    out = _empty_like(t, **kw)
    return out.fill_(float("nan")) if out.is_floating_point() else out


def nan_empty(*a, **kw):
    out = _empty(*a, **kw)
    return out.fill_(float("nan")) if out.is_floating_point() else out


torch.empty_like, torch.empty = nan_empty_like, nan_empty
be = HipBackend()
be.ctx.set_advect_halo(halo)
dom = Box(x=100, y=100)
inflow = 0.2 * resample(Sphere(x=50, y=9.5, radius=5), to=CenteredGrid(0, ZERO_GRADIENT, dom, x=n, y=n, backend=be), soft=True)
v = StaggeredGrid(0, 0, dom, x=n, y=n, backend=be)
s = CenteredGrid(0, ZERO_GRADIENT, dom, x=n, y=n, backend=be)
p = None


def nans(f):
    a = f.numpy()
    return sum(int(np.isnan(x).sum()) for x in (a if isinstance(a, list) else [a]))


for k in range(steps):
    s_adv = advect.mac_cormack(s, v, 1.0)
    s1 = s_adv + inflow
    v_adv = advect.semi_lagrangian(v, v, 1.0)
    buoy = resample(s1 * (0, 0.1), to=v)
    v_pre = v_adv + buoy
    v, p = fluid.make_incompressible(v_pre, (), Solve('CG', 0, 0, x0=p, max_iterations=50, suppress=[NotConverged]))
    s = s1
    rep = {name: nans(f) for name, f in (("mac_cormack(s)", s_adv), ("semi_lagrangian(v)", v_adv), ("resample", buoy), ("v_out", v), ("p", p)) if nans(f)}
    print(f"step {k}: max |v| {max(float(np.abs(a).max()) for a in v.numpy()):.3f} fallback {be.ctx.advect_fallback_stats()} NaN counts {rep}", flush=True)
