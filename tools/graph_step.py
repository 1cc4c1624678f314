#!/usr/bin/env python3
"""
The fluid step (semi-Lagrangian self-advection + make_incompressible with a fixed number of CG iterations, no host read-back) captured in a
hipGraph (torch.cuda.CUDAGraph on a side stream) and replayed, against the same step enqueued eagerly: results must be bit-identical,
and the replay removes the host's enqueue cost -- which matters where the loop is launch-bound (2-D grids, small 3-D grids).
    python tools/graph_step.py --res 512,512 --batch 8 --iters 100
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C   # noqa: E402


def run(ctx, res, batch, iters, bc, dtype, reps, device):
    rank = len(res)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    grid = C.make_grid(rank, C.PHIHIP_F64 if dtype == "f64" else C.PHIHIP_F32, batch, res, (0,) * rank, (1,) * rank, ((bc, bc),) * rank)
    shapes = [ctx.component_shape(grid, d) for d in range(rank)]
    g = torch.Generator(device="cpu").manual_seed(0)
    v0 = [(torch.randn(batch, *s, generator=g, dtype=tdt) * 0.05).to(device) for s in shapes]
    solve = C.Solve(0.0, 0.0, iters, 50, 0, 0)          # benchmark mode: exactly `iters` iterations, no host polling
    dt = 0.5 / max(res)

    class State:
        def __init__(self):
            self.v = [t.clone() for t in v0]
            self.v2 = [torch.empty_like(t) for t in v0]
            self.p = torch.zeros(batch, *res, dtype=tdt, device=device)

    def two_steps(st, stream):
        """ v -> v2 -> v: after two steps every buffer is back in its role, so one captured graph can be replayed indefinitely """
        for a, b in ((st.v, st.v2), (st.v2, st.v)):
            pa, pb = [t.data_ptr() for t in a], [t.data_ptr() for t in b]
            ctx.advect_staggered(grid, pa, pa, pb, dt, stream)
            ctx.make_incompressible(grid, pb, None, 0, 1, bc != 2, st.p.data_ptr(), 0, solve, want_info=False, stream=stream)

    side = torch.cuda.Stream(device)
    eager, graph = State(), State()
    with torch.cuda.stream(side):
        for st in (eager, graph):                        # warm-up on the stream that will capture: workspaces grown, launch plans tuned
            two_steps(st, side.cuda_stream)
            st.__init__()
    side.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, stream=side):
        two_steps(graph, torch.cuda.current_stream().cuda_stream)
    # (the state object keeps its buffers: the graph holds their addresses)
    for t, t0 in zip(graph.v, v0):
        t.copy_(t0)
    graph.p.zero_()
    torch.cuda.synchronize(device)

    def timed(fn):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / (2 * reps)

    with torch.cuda.stream(side):
        t_eager = timed(lambda: two_steps(eager, side.cuda_stream))
    t_graph = timed(cg.replay)
    same = all(torch.equal(a, b) for a, b in zip(eager.v + [eager.p], graph.v + [graph.p]))
    finite = bool(torch.isfinite(graph.p).all())
    return {"res": list(res), "batch": batch, "dtype": dtype, "bc": bc, "cg_iterations": iters, "ms_per_step_eager": t_eager * 1e3,
            "ms_per_step_graph_replay": t_graph * 1e3, "speedup": t_eager / t_graph, "bit_identical": same, "finite": finite,
            "steps_compared": 2 * reps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", default="512,512")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--bc", type=int, default=1)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    ctx = C.Context(C.load_default_library(), 0)
    print(json.dumps(run(ctx, tuple(int(x) for x in args.res.split(",")), args.batch, args.iters, args.bc, args.dtype, args.reps, device)), flush=True)


if __name__ == "__main__":
    main()
