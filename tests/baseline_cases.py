"""
Oracle parity on the BASELINE.json configurations themselves: every function runs the workload a number
is quoted for through the C ABI and compares it with the NumPy oracle on the same inputs -- pressure rel-L2 modulo the mean
(north-star: 1e-4 fp32; 1e-9 fp64), velocities, and the solver's own residual after an EXACT number of iterations (tolerances 0).

The size is a parameter: tests/test_gpu_baseline_sizes.py runs the full sizes on the MI355X (the launch plans only large grids
select: chunks of 16-64 planes, (4,64) tiles, bidirectional MATVEC, the deferred x update over 100 iterations),
tests/test_emu_kernels.py the same code at toy sizes under emulation (guards the test logic itself).
"""
import math

import numpy as np

import parity_cases as pc
from parity_cases import C, O, CLO, PER


def _check_residual(r_gpu, r_ref, rtol, floor):
    """ squared relative residual after the same number of iterations; below the rounding floor of the dtype only its level counts """
    if r_ref > floor:
        assert abs(r_gpu - r_ref) <= rtol * r_ref, (r_gpu, r_ref)
    else:
        assert r_gpu <= 100 * floor, (r_gpu, r_ref)


def _p_err(p_gpu, p_ref):
    a, b = pc.demean(np.asarray(p_gpu, np.float64)), pc.demean(np.asarray(p_ref, np.float64))
    return pc.rel_l2(a, b)


def taylor_green_velocity(n, dtype=np.float32):
    """ u = cos x sin y, v = -sin x cos y at the face centres, extruded along z, w = 0 on [0, 2 pi]^3 """
    h = 2 * math.pi / n
    idx = np.arange(n)
    face, cent = idx * h, (idx + 0.5) * h
    u = np.broadcast_to((np.cos(face)[:, None] * np.sin(cent)[None, :])[:, :, None], (n, n, n))
    v = np.broadcast_to((-np.sin(cent)[:, None] * np.cos(face)[None, :])[:, :, None], (n, n, n))
    return [np.ascontiguousarray(a, dtype=dtype)[None] for a in (u, v, np.zeros((n, n, n)))]


def config2_step(ctx, mem, n=256, iters=100, report=None):
    """ BASELINE configs[1]: one benchmark step (semi-Lagrangian self-advection + projection with exactly `iters` CG iterations,
    refresh at 50) of the 3-D periodic Taylor-Green case in fp32 vs the oracle. """
    L = 2 * math.pi
    dom, grid = pc.make_case((n, n, n), ((PER, PER),) * 3, np.float32, upper=(L,) * 3)
    vel = taylor_green_velocity(n)
    dt = 0.5 * L / n
    dv = [mem.to_dev(a) for a in vel]
    dv2 = [mem.empty(a.shape, np.float32) for a in vel]
    dp = mem.to_dev(np.zeros((1, n, n, n), np.float32))
    solve = C.Solve(0.0, 0.0, iters, 50, 0, 0)
    P = lambda ts: [mem.ptr(t) for t in ts]
    ctx.advect_staggered(grid, P(dv), P(dv), P(dv2), dt)
    info = ctx.make_incompressible(grid, P(dv2), None, 0, 1, True, mem.ptr(dp), 0, solve)
    mem.sync()
    vo = O.semi_lagrangian_staggered(vel, vel, dt, dom)
    vo, po, io, _ = O.make_incompressible(vo, dom, rtol=0.0, atol=0.0, max_iter=iters, refresh=50)
    assert info[0].iterations == iters == int(io.iterations[0]) and not info[0].diverged
    p_err = _p_err(mem.to_host(dp), po)
    v_err = max(float(np.abs(mem.to_host(a) - b).max()) for a, b in zip(dv2, vo))
    r_gpu, r_ref = info[0].residual_sq / info[0].rhs_sq, float(io.residual_sq[0] / io.rhs_sq[0])
    if report is not None:
        report.update(size=n, iterations=iters, pressure_rel_l2=p_err, velocity_max_abs=v_err, rel_residual_sq=r_gpu, rel_residual_sq_oracle=r_ref)
    assert p_err <= 1e-4, f"pressure rel-L2 {p_err:.3e} vs oracle at {n}^3 / {iters} iterations"
    assert v_err <= 2e-5, f"velocity max abs error {v_err:.3e}"
    # the Taylor-Green rhs is one smooth mode: CG converges within a few iterations and then sits on the fp32 rounding floor, where
    # the residual is noise -- only its level is comparable (the pressure itself is compared above)
    _check_residual(r_gpu, r_ref, 1.0, 1e-10)
    return p_err, v_err


def config3_solve(ctx, mem, n=512, iters=20, report=None):
    """ BASELINE configs[2]: pressure solve only, periodic fp32, mean-free pseudo-random rhs (numpy default_rng(0)), x0 = 0,
    exactly `iters` iterations vs the oracle's CG on the same rhs. """
    L = 2 * math.pi
    dom, grid = pc.make_case((n, n, n), ((PER, PER),) * 3, np.float32, upper=(L,) * 3)
    rhs = np.random.default_rng(0).standard_normal((1, n, n, n), dtype=np.float32)
    rhs -= rhs.mean(dtype=np.float64).astype(np.float32)
    drhs, dx = mem.to_dev(rhs), mem.to_dev(np.zeros_like(rhs))
    info = ctx.cg_solve(grid, 0, 1, mem.ptr(drhs), mem.ptr(dx), C.Solve(0.0, 0.0, iters, 50, 0, 0))
    mem.sync()
    xo, io = O.cg(lambda q: O.masked_laplace(q, dom, None, None), rhs, np.zeros_like(rhs), 0.0, 0.0, iters, 50)
    assert info[0].iterations == iters == int(io.iterations[0]) and not info[0].diverged
    p_err = _p_err(mem.to_host(dx), xo)
    r_gpu, r_ref = info[0].residual_sq / info[0].rhs_sq, float(io.residual_sq[0] / io.rhs_sq[0])
    if report is not None:
        report.update(size=n, iterations=iters, pressure_rel_l2=p_err, rel_residual_sq=r_gpu, rel_residual_sq_oracle=r_ref)
    assert p_err <= 1e-4, f"pressure rel-L2 {p_err:.3e} vs oracle at {n}^3 / {iters} iterations"
    _check_residual(r_gpu, r_ref, 1e-2, 1e-10)
    return p_err


def config5_cavity(ctx, mem, n=256, iters=20, report=None):
    """ BASELINE configs[4] (size class): fp64 lid-driven cavity, closed box with the lid velocity (1,0,0) on z+, one solid box in
    the centre -> active / hard_bcs flags; advect + apply_boundary_conditions + projection with exactly `iters` iterations. """
    bcv = np.zeros((3, 2, 3)); bcv[2, 1, 0] = 1.0
    dom, grid = pc.make_case((n, n, n), ((CLO, CLO),) * 3, np.float64, bc_val=bcv, upper=(1.0, 1.0, 1.0))
    rng = np.random.default_rng(5)
    vel = [0.05 * rng.standard_normal((1,) + dom.comp_shape(d)) for d in range(3)]
    obstacles = [O.BoxObstacle((0.375, 0.375, 0.375), (0.625, 0.625, 0.625))]
    items = pc.obstacle_items(obstacles, 3)
    cobs = C.make_obstacles(items)
    dt = 0.5 / n / 0.15                                   # CFL ~ 0.5 for |u| ~ 3 sigma
    g1 = C.make_grid(3, grid.dtype, 1, dom.res, dom.lower, dom.upper, dom.bc, dom.bc_val)
    dacc, dflags = mem.empty(dom.res, np.uint8), mem.empty(dom.res, np.uint8)
    ctx.obstacle_accessible(g1, cobs, len(items), mem.ptr(dacc))
    ctx.build_cellflags(g1, mem.ptr(dacc), 0, 1, mem.ptr(dflags))
    dv = [mem.to_dev(a) for a in vel]
    dv2 = [mem.empty(a.shape, np.float64) for a in vel]
    dp = mem.to_dev(np.zeros((1, n, n, n), np.float64))
    P = lambda ts: [mem.ptr(t) for t in ts]
    ctx.advect_staggered(grid, P(dv), P(dv), P(dv2), dt)
    ctx.apply_obstacles(grid, cobs, len(items), P(dv2))
    info = ctx.make_incompressible(grid, P(dv2), None, mem.ptr(dflags), 1, True, mem.ptr(dp), 0, C.Solve(0.0, 0.0, iters, 50, 0, 0))
    mem.sync()
    vo = O.semi_lagrangian_staggered(vel, vel, dt, dom)
    vo, po, io, _ = O.make_incompressible(vo, dom, obstacles, rtol=0.0, atol=0.0, max_iter=iters, refresh=50)
    assert info[0].iterations == iters == int(io.iterations[0]) and not info[0].diverged
    pg = mem.to_host(dp)
    p_err = pc.rel_l2(pg, po)                              # with obstacles the oracle balances on the active cells: no free mean
    v_err = max(float(np.abs(mem.to_host(a) - b).max()) for a, b in zip(dv2, vo))
    r_gpu, r_ref = info[0].residual_sq / info[0].rhs_sq, float(io.residual_sq[0] / io.rhs_sq[0])
    if report is not None:
        report.update(size=n, iterations=iters, pressure_rel_l2=p_err, velocity_max_abs=v_err, rel_residual_sq=r_gpu, rel_residual_sq_oracle=r_ref)
    assert p_err <= 1e-9, f"fp64 pressure rel-L2 {p_err:.3e}"
    assert v_err <= 1e-10, f"fp64 velocity max abs error {v_err:.3e}"
    _check_residual(r_gpu, r_ref, 1e-8, 1e-26)
    return p_err, v_err


def config4_batched_smoke(ctx, mem, n=512, B=8, steps=3, iters=60, report=None):
    """ BASELINE configs[3]: B x n^2 batched smoke plumes (closed box, per-entry inflow position x in linspace(30, 70, B)), `steps`
    steps of Smoke_Plume.ipynb cell 5 (mac_cormack smoke + inflow, semi-Lagrangian velocity + buoyancy resample, projection from
    the previous pressure) with exactly `iters` CG iterations per projection, every field of every step vs the oracle. """
    dom, grid = pc.make_case((n, n), ((CLO, CLO),) * 2, np.float32, batch=B, upper=(100.0, 100.0))
    cp = O.cell_positions(dom, np.float64)
    xs = np.linspace(30, 70, B)
    inflow = np.stack([(((cp[0] - x0) ** 2 + (cp[1] - 9.5) ** 2) <= 25).astype(np.float32) for x0 in xs]) * np.float32(0.2)
    s_codes = ((O.OPEN, O.OPEN),) * 2                      # ZERO_GRADIENT smoke
    P = lambda ts: [mem.ptr(t) for t in ts]
    smoke = np.zeros((B, n, n), np.float32)
    v = [np.zeros((B,) + dom.comp_shape(d), np.float32) for d in range(2)]
    p = np.zeros((B, n, n), np.float32)
    d_smoke, d_smoke2 = mem.to_dev(smoke), mem.empty(smoke.shape, np.float32)
    d_inflow = mem.to_dev(inflow)
    dv, dv2 = [mem.to_dev(a) for a in v], [mem.empty(a.shape, np.float32) for a in v]
    dp = mem.to_dev(p)
    solve = C.Solve(0.0, 0.0, iters, 50, 0, 0)
    worst = dict(smoke=0.0, v=0.0, p=0.0, res=0.0)
    for step in range(steps):
        ctx.mac_cormack_centered(grid, mem.ptr(d_smoke), s_codes, None, P(dv), mem.ptr(d_smoke2), 1.0, 1.0)
        mem.sync()
        d_smoke2 += d_inflow                               # `+ inflow` (elementwise glue on the device tensors / host arrays)
        ctx.advect_staggered(grid, P(dv), P(dv), P(dv2), 1.0)
        ctx.centered_to_staggered(grid, mem.ptr(d_smoke2), s_codes, None, (0.0, 0.1), True, P(dv2))
        info = ctx.make_incompressible(grid, P(dv2), None, 0, 1, True, mem.ptr(dp), 0, solve)
        mem.sync()
        d_smoke, d_smoke2 = d_smoke2, d_smoke
        dv, dv2 = dv2, dv
        smoke = O.mac_cormack_centered(smoke, v, 1.0, dom, s_codes) + inflow
        buoy = O.centered_to_staggered(smoke, dom, s_codes, None, (0.0, 0.1))
        v = O.semi_lagrangian_staggered(v, v, 1.0, dom)
        v = [a + b for a, b in zip(v, buoy)]
        v, p, io, _ = O.make_incompressible(v, dom, x0=p, rtol=0.0, atol=0.0, max_iter=iters, refresh=50)
        assert [i.iterations for i in info] == [iters] * B == [int(k) for k in io.iterations]
        worst['smoke'] = max(worst['smoke'], pc.rel_l2(mem.to_host(d_smoke), smoke))
        worst['p'] = max(worst['p'], max(_p_err(mem.to_host(dp)[b:b + 1], p[b:b + 1]) for b in range(B)))
        scale = max(float(np.abs(a).max()) for a in v)
        worst['v'] = max(worst['v'], max(float(np.abs(mem.to_host(a) - b).max()) for a, b in zip(dv, v)) / scale)
        worst['res'] = max(worst['res'], max([0.0] + [abs(i.residual_sq / i.rhs_sq - float(ro / yo)) / float(ro / yo)
                                             for i, ro, yo in zip(info, io.residual_sq, io.rhs_sq) if float(ro / yo) > 1e-10]))
    if report is not None:
        report.update(size=n, batch=B, steps=steps, iterations=iters, **{f"worst_{k}": v_ for k, v_ in worst.items()})
    assert worst['smoke'] <= 2e-5 and worst['p'] <= 1e-4 and worst['v'] <= 1e-4 and worst['res'] <= 2e-2, worst
    return worst


def max_size_step(ctx, mem, n=1024, iters=20, report=None):
    """ The largest grid this repository runs (1024^3 fp32: 2^30 cells, 4.3 GB per array, ~50 GB for the step) through size-INDEPENDENT properties -- the
    oracle cannot hold it. The velocity is a Taylor-Green vortex plus a seeded rough part in the (y, z) plane, extruded along x, the SLOWEST axis
    (where the large element offsets live), u_x = 0. One benchmark step (self-advection + projection with exactly `iters` CG iterations):
      * every x-plane of every result equals plane 0 BIT FOR BIT (each cell of a plane sees the same operands in the same order whatever its plane; an
        element offset that wraps at 2^31 / 2^32 bytes, a chunk boundary or a ring slot that is treated differently breaks it);
      * plane 0 equals the 2-D oracle's step on the n x n grid of the (y, z) plane (a field that does not depend on x has no x-flux: the 3-D operator is
        the 2-D one, every dot product is n times the 2-D one) -- velocity to the advection tolerance, pressure rel-L2 to the north-star's 1e-4. """
    L = 2 * math.pi
    dom, grid = pc.make_case((n, n, n), ((PER, PER),) * 3, np.float32, upper=(L,) * 3)
    h = L / n
    idx = np.arange(n)
    face, cent = idx * h, (idx + 0.5) * h
    # Taylor-Green in the (y, z) plane + a seeded rough part (0.3 N(0, 1): CFL < 1 at dt = h / 2): the vortex alone leaves a right-hand side of ONE smooth mode whose
    # solve sits on the fp32 rounding floor after three iterations, where two summation orders agree only to ~2e-4 (measured at 1024^3); a rough right-hand side keeps
    # CG working for all its iterations, like the seeded right-hand side of config3_solve
    rng = np.random.default_rng(1024)
    pv = (np.cos(face)[:, None] * np.sin(cent)[None, :] + 0.3 * rng.standard_normal((n, n))).astype(np.float32)          # y component at (y face, z centre)
    pw = (-np.sin(cent)[:, None] * np.cos(face)[None, :] + 0.3 * rng.standard_normal((n, n))).astype(np.float32)         # z component at (y centre, z face)
    dt = 0.5 * h
    dv = [mem.extrude(np.zeros((n, n), np.float32), n), mem.extrude(pv, n), mem.extrude(pw, n)]
    dv2 = [mem.empty((1, n, n, n), np.float32) for _ in range(3)]
    dp = mem.extrude(np.zeros((n, n), np.float32), n)
    solve = C.Solve(0.0, 0.0, iters, 50, 0, 0)
    P = lambda ts: [mem.ptr(t) for t in ts]
    ctx.advect_staggered(grid, P(dv), P(dv), P(dv2), dt)
    mem.sync()
    adv0 = [mem.first_plane(a) for a in dv2[1:]]                                     # the advected velocity, before the projection works on it in place
    info = ctx.make_incompressible(grid, P(dv2), None, 0, 1, True, mem.ptr(dp), 0, solve)
    mem.sync()
    assert info[0].iterations == iters and not info[0].diverged, (info[0].iterations, info[0].diverged)
    names = ("velocity x", "velocity y", "velocity z", "pressure")
    for name, t in zip(names, dv2 + [dp]):
        assert mem.planes_equal_first(t), f"{name}: an x-plane differs from plane 0 at {n}^3"
    dom2 = O.Domain((n, n), (0.0, 0.0), (L, L), ((PER, PER),) * 2)
    # the 2-D oracle in DOUBLE precision on the same fp32 inputs: at index ~1000 the fp32 oracle's lookup coordinate (index - dt u / dx, one rounding at ulp(1000) =
    # 6e-5 cells) costs 1.7e-4 on this rough field (measured), the kernels look up displacement-relative and do not pay it -- the reference must not be
    # the less accurate side
    vel2 = [pv[None].astype(np.float64), pw[None].astype(np.float64)]
    vo = O.semi_lagrangian_staggered(vel2, vel2, dt, dom2)
    a_err = max(float(np.abs(a - b).max()) for a, b in zip(adv0, vo))
    vo, po, io, _ = O.make_incompressible(vo, dom2, rtol=0.0, atol=0.0, max_iter=iters, refresh=50)
    assert int(io.iterations[0]) == iters
    ux = mem.first_plane(dv2[0])
    assert float(np.abs(ux).max()) == 0.0, "the x component of an x-invariant flow without x velocity stays exactly zero"
    v_err = max(float(np.abs(mem.first_plane(a) - b).max()) for a, b in zip(dv2[1:], vo))
    p_err = _p_err(mem.first_plane(dp), po)
    if report is not None:
        report.update(size=n, iterations=iters, cells=n ** 3, advected_max_abs_vs_2d_oracle=a_err, velocity_max_abs_vs_2d_oracle=v_err, pressure_rel_l2_vs_2d_oracle=p_err)
    assert a_err <= 2e-5, f"advected velocity of plane 0 vs the 2-D oracle: {a_err:.3e}"
    # after the projection: v = v' - grad p with 1 / dx = n / (2 pi): rounding differences of p between two summation orders (3-D kernels: 2^30-term dot products;
    # 2-D oracle) are multiplied by 163 at n = 1024 (41 at the 256^3 of config2_step, whose bound is 2e-5): the velocity bound scales with n / 256
    v_tol = 2e-5 * max(1.0, n / 256.0)
    assert v_err <= v_tol, f"velocity of plane 0 vs the 2-D oracle: {v_err:.3e} (bound {v_tol:.1e})"
    assert p_err <= 1e-4, f"pressure of plane 0 vs the 2-D oracle: rel-L2 {p_err:.3e}"
    return v_err, p_err
