import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, parity_cases as pc
import fuzz_parity as F
from phiflow_amd import _capi as C
O=pc.O
lib=C.load_default_library(); ctx=C.Context(lib,0); mem=pc.TorchMem()
orig=pc.check_resident_with_flags
def probe(ctx_, mem_, res, bc, batch, obstacles, seed=11, projection=True):
    dtype=np.float32
    dom, grid = pc.make_case(res, bc, dtype, batch=batch)
    active, hard, soft = O.obstacle_masks(obstacles, dom, dtype)
    acc=(active[0]>0).astype(np.uint8)
    dacc, dflags = mem.to_dev(acc), mem.empty(dom.res, np.uint8)
    g1 = C.make_grid(dom.rank, grid.dtype, 1, dom.res, dom.lower, dom.upper, dom.bc, dom.bc_val)
    ctx.build_cellflags(g1, mem.ptr(dacc), 0, 1, mem.ptr(dflags)); mem.sync()
    flags=mem.to_host(dflags)
    print("case", res, bc, batch, "solid cells", int(acc.size-acc.sum()))
    rng=np.random.default_rng(seed+1)
    v=pc.random_velocity(dom, batch, dtype, rng)
    div=O.divergence(v, dom)*active
    rhs=div if dom.flexible() else O.balance_divergence(div, active)
    for mode in (0,2):
        ctx.set_resident_cg(mode)
        for small in ((True,False) if mode==0 else (True,)):
            ctx.set_small_grid_solver(small)
            s=pc.solve_params(dtype, 3000, None, 0.0, 50, 10, 0)
            drhs, dx = mem.to_dev(rhs.astype(dtype)), mem.to_dev(np.zeros_like(rhs))
            info=ctx.cg_solve(grid, mem.ptr(dflags), 1, mem.ptr(drhs), mem.ptr(dx), s); mem.sync()
            print("mode", mode, "small", small, [(i.iterations, i.converged) for i in info])
    ctx.set_small_grid_solver(True)
    A=lambda q: O.masked_laplace(q, dom, hard, active)
    xo, io = O.cg(A, rhs.astype(dtype), np.zeros_like(rhs), 1e-5, 0.0, 3000, 50)
    print("oracle fp32", list(io.iterations))
    xo, io = O.cg(A, rhs.astype(np.float64), np.zeros(rhs.shape), 1e-5, 0.0, 3000, 50)
    print("oracle fp64", list(io.iterations))
pc.check_resident_with_flags=probe
sys.argv=['fuzz','--first','64244','--count','1']
F.main()
