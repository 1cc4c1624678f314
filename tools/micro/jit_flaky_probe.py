#!/usr/bin/env python3
"""
r6 probe: eager step vs hipGraph replay of the 2-D smoke plume (tests/test_jit.py _plume), many trials on ONE used context, with the intermediates of the step
returned as extra outputs and both sides REPEATED from the same inputs when they disagree -- which operator moves first, and which side (eager / replay) is the
one that is not a function of its inputs.       python tools/micro/jit_flaky_probe.py [trials]       (needs an MI355X)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc                                                                                                      # noqa: E402
from parity_cases import CLO, OPN, PER                                                                                        # noqa: E402
from phiflow_amd.backend import HipBackend                                                                                    # noqa: E402
from phiflow_amd.flow import (ZERO_GRADIENT, Box, CenteredGrid, NotConverged, Solve, Sphere, StaggeredGrid, advect, fluid, jit_compile, resample)   # noqa: E402


def arrays(fields):
    out = []
    for f in fields:
        a = f.numpy()
        out += a if isinstance(a, list) else [a]
    return out


def diff(a, b):
    msgs = []
    for i, (x, y) in enumerate(zip(arrays(a), arrays(b))):
        if x.tobytes() != y.tobytes():          # BIT comparison: -0.0 vs +0.0 counts
            d = np.abs(x.astype(np.float64) - y.astype(np.float64))
            msgs.append(f"[{i}] {int((x.view(np.uint32) != y.view(np.uint32)).sum())} words differ, max |d| {d.max():.2e}")
    return "; ".join(msgs)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    be = HipBackend()
    ctx = be.ctx
    mem = pc.TorchMem(str(be.device))
    rng = np.random.default_rng(1)
    # use the context like the parity suite does before the jit tests
    warm = (((64, 128), ((PER, PER), (OPN, OPN))), ((24, 40, 128), ((PER, PER),) * 3), ((40, 36, 256), ((CLO, CLO),) * 3))
    for res, bc in (() if os.environ.get('PROBE_NO_WARMUP') else warm):
        dom, grid = pc.make_case(res, bc, np.float32, batch=2)
        pc.check_advect_staggered(ctx, mem, dom, grid, np.float32, rng, dt=0.7)
        pc.check_cg(ctx, mem, dom, grid, np.float32, rng)
        torch.cuda.synchronize(); print('parity warm-up ok', res, flush=True)
    names = ["v3", "s_", "p3", "s1", "va", "bu", "v2"]
    bad = 0
    for trial in range(trials):
        sizes = [(int(a), 20) for a in os.environ['PROBE_SIZES'].split(',')] if os.environ.get('PROBE_SIZES') else ((32, 30), (128, 50), (192, 20))
        modes = [int(a) for a in os.environ.get('PROBE_RESIDENT', '1,0').split(',')]
        for n, iters in sizes:
            for extra in (False, True):
                for resident in modes:
                    ctx.set_resident_cg(resident)
                    dom = Box(x=100, y=100)
                    inflow = 0.2 * resample(Sphere(x=50, y=9.5, radius=5), to=CenteredGrid(0, ZERO_GRADIENT, dom, x=n, y=n, backend=be), soft=True)
                    v0 = StaggeredGrid(0, 0, dom, x=n, y=n, backend=be)
                    s0 = CenteredGrid(0, ZERO_GRADIENT, dom, x=n, y=n, backend=be)

                    def step(v, s, p):
                        s1 = advect.mac_cormack(s, v, 1.0)
                        s_ = s1 + inflow
                        va = advect.semi_lagrangian(v, v, 1.0)
                        bu = resample(s_ * (0, 0.1), to=v)
                        v2 = va + bu
                        v3, p3 = fluid.make_incompressible(v2, (), Solve('CG', 0, 0, x0=p, max_iterations=iters, suppress=[NotConverged]))
                        return (v3, s_, p3, s1, va, bu, v2) if extra else (v3, s_, p3)
                    jstep = jit_compile(step)
                    print(f'-- trial {trial} n={n} extra={extra} resident={resident}', flush=True)
                    se, sj = (v0, s0, None), (v0, s0, None)
                    for k in range(8):
                        oe = step(*se); torch.cuda.synchronize(); print('   eager ok', k, flush=True)
                        oj = jstep(*sj); torch.cuda.synchronize(); print('   replay ok', k, flush=True)
                        d = diff(oe, oj)
                        if d:
                            bad += 1
                            oe2, oj2 = step(*se), jstep(*sj)
                            print(f"trial {trial} n={n} extra={extra} resident={resident} step {k}: eager vs replay: {d} || eager again vs eager: {diff(oe2, oe) or 'same'} || "
                                  f"replay again vs replay: {diff(oj2, oj) or 'same'} || inputs: {diff(se, sj) or 'same bits'} || fallback {ctx.advect_fallback_stats()} "
                                  f"plan {[ctx.query_plan(be.grid_of(v0) if hasattr(be, 'grid_of') else None, False, f) for f in ()] }", flush=True)
                            break
                        se, sj = oe[:3], oj[:3]
    ctx.set_resident_cg(1)
    print(f"done: {bad} disagreements in {trials} trials", flush=True)


if __name__ == "__main__":
    main()
