#!/usr/bin/env python3
""" r6: where does a differentiated 256^3 step spend its wall time? Host enqueue time (call returns) vs completion (after the sync), forward and backward separately. """
import math, os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from phiflow_amd.flow import *   # noqa
from phiflow_amd.flow import default_backend, functional_gradient, l2_loss, NotConverged   # noqa

n, L = 256, 2 * math.pi
be = default_backend()
h = L / n
face = torch.arange(n, dtype=torch.float32) * h
cent = (torch.arange(n, dtype=torch.float32) + 0.5) * h
u = (torch.cos(face)[:, None, None] * torch.sin(cent)[None, :, None]).expand(n, n, n)
w = (-torch.sin(cent)[:, None, None] * torch.cos(face)[None, :, None]).expand(n, n, n)
comps = [t.contiguous()[None].to(be.device) for t in (u, w, torch.zeros(n, n, n))]
mk = lambda: StaggeredGrid([c.clone() for c in comps], PERIODIC, Box(x=L, y=L, z=L), x=n, y=n, z=n)
solve = Solve('CG', 0, 0, max_iterations=100, suppress=[NotConverged])
dt = 0.5 * h

def simulate(v):
    v = advect.semi_lagrangian(v, v, dt)
    v, p = fluid.make_incompressible(v, (), solve)
    return l2_loss(v)

grad = functional_gradient(simulate, wrt=[0], get_output=True)
# wall time of every C-ABI call (no extra synchronisation): which call of a slow step blocks?
from phiflow_amd import _capi as C   # noqa
CALLS = []
for name in ([m for m in dir(C.Context) if not m.startswith("_") and callable(getattr(C.Context, m))] if os.environ.get("PROBE_WRAP") else []):
    def wrap(fn, name=name):
        def inner(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                CALLS.append((name, (time.perf_counter() - t) * 1e3))
        return inner
    setattr(C.Context, name, wrap(getattr(C.Context, name)))
simulate(mk()); grad(mk()); torch.cuda.synchronize()
CALLS.clear()
import gc
if os.environ.get("PROBE_GC") == "0":
    gc.disable()
if os.environ.get("PROBE_GC") == "stats":
    gc.callbacks.append(lambda phase, info: print("gc", phase, info, round(time.perf_counter() * 1e3, 2), flush=True))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    v0 = mk(); torch.cuda.synchronize()
    import faulthandler
    faulthandler.dump_traceback_later(0.030, repeat=False)          # a call that blocks longer than 30 ms prints where (all threads)
    t0 = time.perf_counter()
    loss, (g,) = grad(v0)
    t1 = time.perf_counter()
    faulthandler.cancel_dump_traceback_later()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"rep": rep, "grad_call_returns_ms": round((t1 - t0) * 1e3, 2), "complete_ms": round((t2 - t0) * 1e3, 2),
                      "c_abi_calls_ms": [(n, round(ms, 2)) for n, ms in CALLS if ms > 0.3]}), flush=True)
    CALLS.clear()
for rep in range(3):
    v0 = mk(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        simulate(v0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"rep": rep, "forward_call_returns_ms": round((t1 - t0) * 1e3, 2), "complete_ms": round((t2 - t0) * 1e3, 2)}), flush=True)
