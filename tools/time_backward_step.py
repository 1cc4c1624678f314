#!/usr/bin/env python3
"""
Forward and backward time of ONE differentiated fluid step through the phi-level API (the pattern of tests/commit/physics/test_fluid.py:55-73
and test_colab_fluids_tutorial.py:11-34 at benchmark size): loss = l2(make_incompressible(semi_lagrangian(v, v, dt))), gradient w.r.t. v.
The backward pass = the implicit-function adjoint of the projection (one more CG solve with the same iteration count) + the gather-form
adjoint of the self-advection.
    python tools/time_backward_step.py --size 256 --iters 100
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd.flow import *   # noqa: E402,F401,F403
from phiflow_amd.flow import default_backend, functional_gradient, l2_loss, NotConverged   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    n, L = args.size, 2 * math.pi
    be = default_backend()
    h = L / n
    face = torch.arange(n, dtype=torch.float32) * h
    cent = (torch.arange(n, dtype=torch.float32) + 0.5) * h
    u = (torch.cos(face)[:, None, None] * torch.sin(cent)[None, :, None]).expand(n, n, n)
    w = (-torch.sin(cent)[:, None, None] * torch.cos(face)[None, :, None]).expand(n, n, n)
    comps = [t.contiguous()[None].to(be.device) for t in (u, w, torch.zeros(n, n, n))]
    mk = lambda: StaggeredGrid([c.clone() for c in comps], PERIODIC, Box(x=L, y=L, z=L), x=n, y=n, z=n)
    solve = Solve('CG', 0, 0, max_iterations=args.iters, suppress=[NotConverged])       # benchmark mode: exactly `iters` iterations, forward and backward
    dt = 0.5 * h

    def simulate(v):
        v = advect.semi_lagrangian(v, v, dt)
        v, p = fluid.make_incompressible(v, (), solve)
        return l2_loss(v)

    grad = functional_gradient(simulate, wrt=[0], get_output=True)

    def sync():
        torch.cuda.synchronize()

    simulate(mk()); grad(mk()); grad(mk()); sync()
    # r6: per-repetition times (each bracketed by a synchronisation) and their median. The mean of three back-to-back calls that rounds 3-5 reported is at the mercy of
    # CPython's generation-2 garbage collection: with torch's object graph one collection takes 33-38 ms and falls wherever the allocation counter trips -- with the r6
    # library in the SECOND differentiated step of this script, which made the 3-repetition mean read 30 ms instead of 18 (found with faulthandler + gc.callbacks:
    # profiles/r06_backward_step.txt). The median of 7 does not see it; `gc.freeze()` after the warm-up keeps long-lived objects out of later collections.
    import gc
    gc.collect(); gc.freeze()
    def timed(fn):
        ts = []
        for _ in range(args.reps):
            v0 = mk(); sync()
            t0 = time.perf_counter()
            out = fn(v0)
            sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        return out, ts

    def fwd(v0):
        with torch.no_grad():
            return simulate(v0)
    _, t_f = timed(fwd)
    (loss, (g,)), t_b = timed(grad)
    med = lambda ts: sorted(ts)[len(ts) // 2]
    t_fwd, t_both = med(t_f) * 1e-3, med(t_b) * 1e-3
    gn = math.sqrt(sum(float((c.astype('float64') ** 2).sum()) for c in g.numpy()))
    print(json.dumps({"size": n, "cg_iterations": args.iters, "ms_forward_only": t_fwd * 1e3, "ms_forward_plus_backward": t_both * 1e3,
                      "ms_backward": (t_both - t_fwd) * 1e3, "reps": args.reps, "ms_forward_each": [round(t, 3) for t in t_f], "ms_forward_plus_backward_each": [round(t, 3) for t in t_b], "loss": float(loss.detach()), "gradient_norm": gn, "finite": math.isfinite(gn)}), flush=True)


if __name__ == "__main__":
    main()
