"""
Advection schemes backed by libphihip (reference: phi/physics/advect.py).
Implemented on the HIP backend: `semi_lagrangian` / `advect` / `mac_cormack` with the `euler` (fused kernels) and `rk4` / `finite_rk4` integrators for StaggeredGrid
and CenteredGrid fields advected by a StaggeredGrid velocity. Anything else raises `NotImplementedError`.
"""
from typing import Callable

import torch

from . import autodiff
from .extrapolation import ConstantExtrapolation, resolve
from .field import Field, _ptrs


def euler(data: Field, velocity: Field, dt: float, v0=None):
    """ Euler integrator marker (phi/physics/advect.py:20-24). The back-trace itself is fused into the HIP kernel, so this
    function only identifies the integrator when passed to `semi_lagrangian`. """
    raise NotImplementedError("euler() is fused into the HIP semi-Lagrangian kernel; pass it as `integrator=euler`")


def rk4(data: Field, velocity: Field, dt: float, v0=None):
    """ Runge-Kutta-4 integrator marker (phi/physics/advect.py:27-36): `semi_lagrangian(..., integrator=rk4)` traces the sample points
    back with four velocity evaluations (sampling.integrate_points) instead of one. """
    raise NotImplementedError("pass rk4 as `integrator=rk4`")


def finite_rk4(data: Field, velocity: Field, dt: float, v0=None):
    """ rk4 with Euler fallback where the velocity is not finite (phi/physics/advect.py:38-47); marker like `rk4` """
    raise NotImplementedError("pass finite_rk4 as `integrator=finite_rk4`")


def semi_lagrangian(field: Field, velocity: Field, dt: float, integrator: Callable = euler) -> Field:
    """ Semi-Lagrangian advection with backward Euler lookup (phi/physics/advect.py:156-179).

    Args:
        field: quantity to be advected (`StaggeredGrid` or `CenteredGrid`)
        velocity: `StaggeredGrid`; on the same grid the fused kernels run, otherwise the general sampling path (sampling.py)
        dt: time increment
        integrator: `euler` (fused kernels), `rk4` or `finite_rk4` (general sampling path)

    Returns:
        Field with the same sample points and boundary as `field`
    """
    return _advect(field, velocity, dt, integrator, None)


def _scalar_boundary(field: Field):
    s_codes, s_vals = resolve(field.boundary, field.dims)
    s_val = [[s_vals[a][s][0] if isinstance(field.boundary.side(d, bool(s)), ConstantExtrapolation) else 0.0 for s in range(2)]
             for a, d in enumerate(field.dims)]
    return s_codes, s_val


def _advect(field: Field, velocity: Field, dt: float, integrator: Callable, correction_strength) -> Field:
    """ shared argument handling of semi_lagrangian (correction_strength None) and mac_cormack """
    from .field import require_plain
    require_plain(field, 'advect'); require_plain(velocity, 'advect (velocity)')
    if integrator not in (euler, rk4, finite_rk4):
        raise NotImplementedError("HIP backend: grid advection supports the integrators euler, rk4 and finite_rk4")
    if not velocity.is_staggered:
        raise NotImplementedError("HIP backend: the advecting velocity must be a StaggeredGrid")
    from . import sampling
    if integrator is not euler:   # Runge-Kutta back-trace: explicit velocity evaluations at the intermediate points
        return sampling.advect_general(field, velocity, float(dt), correction_strength, integrator.__name__)
    if not sampling.same_grid(field, velocity):
        # "velocity need not be sampled at same locations as field" (advect.py:193): gathers at explicit coordinates instead of the
        # fused kernels (Batched_Smoke.ipynb: 200^2 smoke advected by a 64^2 velocity)
        return sampling.advect_general(field, velocity, float(dt), correction_strength)
    be = velocity.backend
    assert field.dtype == velocity.dtype, "field and velocity must have the same precision"
    B = max(field.batch_size, velocity.batch_size)
    vel = [_expand(t, B) for t in velocity.values]
    grid = velocity.grid_struct(batch=B)
    if field.is_staggered:
        same_layout = [tuple(a.shape[1:]) for a in field.values] == [tuple(b.shape[1:]) for b in velocity.values]
        if not same_layout or resolve(field.boundary, field.dims) != resolve(velocity.boundary, velocity.dims):
            raise NotImplementedError("HIP backend: an advected StaggeredGrid must share the velocity's boundary conditions")
        src = vel if field is velocity else [_expand(t, B) for t in field.values]
        if autodiff.needs_grad(*src, *vel):
            if correction_strength is not None:
                out = list(autodiff.MacCormackStaggered.apply(dict(be=be, grid=grid, dt=float(dt), strength=correction_strength), *src, *vel))
            else:
                out = list(autodiff.SemiLagrangianStaggered.apply(dict(be=be, grid=grid, dt=float(dt)), *src, *vel))
            return Field(field.resolution, field.bounds, field.boundary, out, True, be, field.batched or velocity.batched)
        out = [torch.empty_like(t) for t in src]
        if correction_strength is None:
            be.ctx.advect_staggered(grid, _ptrs(src), _ptrs(vel), _ptrs(out), dt, be.stream())
        else:
            be.ctx.mac_cormack_staggered(grid, _ptrs(src), _ptrs(vel), _ptrs(out), dt, correction_strength, be.stream())
        return Field(field.resolution, field.bounds, field.boundary, out, True, be, field.batched or velocity.batched)
    src = _expand(field.values, B)
    s_codes, s_val = _scalar_boundary(field)
    if autodiff.needs_grad(src, *vel):
        meta = dict(be=be, grid=grid, dt=float(dt), s_codes=s_codes, s_val=s_val, strength=correction_strength)
        fn = autodiff.SemiLagrangianCentered if correction_strength is None else autodiff.MacCormackCentered
        return Field(field.resolution, field.bounds, field.boundary, fn.apply(meta, src, *vel), False, be, field.batched or velocity.batched)
    out = torch.empty_like(src)
    if correction_strength is None:
        be.ctx.advect_centered(grid, src.data_ptr(), s_codes, s_val, _ptrs(vel), out.data_ptr(), dt, be.stream())
    else:
        be.ctx.mac_cormack_centered(grid, src.data_ptr(), s_codes, s_val, _ptrs(vel), out.data_ptr(), dt, correction_strength, be.stream())
    return Field(field.resolution, field.bounds, field.boundary, out, False, be, field.batched or velocity.batched)


def advect(field: Field, velocity: Field, dt: float, integrator: Callable = euler) -> Field:
    """ `advect.advect` for grids == `semi_lagrangian` (phi/physics/advect.py:50-75) """
    return semi_lagrangian(field, velocity, dt, integrator=integrator)


def mac_cormack(field: Field, velocity: Field, dt: float, correction_strength=1.0, integrator: Callable = euler) -> Field:
    """ MacCormack advection (phi/physics/advect.py:182-215): forward + backward semi-Lagrangian lookups estimate the first-order
    error, the corrected value is clamped to the grid values around the backward lookup.

    Args:
        field: `CenteredGrid` or `StaggeredGrid` to be advected
        velocity: `StaggeredGrid` on the same grid
        dt: time increment
        correction_strength: factor on the error estimate (0 = semi-Lagrangian)
        integrator: `euler` (fused kernels), `rk4` or `finite_rk4` (general sampling path; centred fields only)
    """
    return _advect(field, velocity, dt, integrator, float(correction_strength))


def _expand(t: torch.Tensor, B: int) -> torch.Tensor:
    if t.shape[0] == B:
        return t.contiguous()
    assert t.shape[0] == 1
    return t.expand(B, *t.shape[1:]).contiguous()
