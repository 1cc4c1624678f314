"""
Pressure projection on the HIP backend (reference: phi/physics/fluid.py).

`make_incompressible` keeps the reference's signature and semantics for the order-2 StaggeredGrid path:
    obstacle masks (device kernels; the reference: `with NUMPY:` fluid.py:130-137)  ->  divergence [* active]  ->  balance (non-flexible
    boundaries)  ->  CG on the masked Laplacian from x0  ->  v -= hard_bcs * grad p.
The linear operator is applied matrix-free by HIP kernels (the reference assembles a sparse matrix per call).
"""
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _capi, autodiff
from .extrapolation import pressure_extrapolation
from .field import Field, _check_pressure_padding, _ptrs, same_grid
from .geom import Box, Geometry, Sphere
from .geom import Embedded, Union as Union_
from .solve import Diverged, NotConverged, Solve, SolveInfo


class Obstacle:
    """ An obstacle defines boundary conditions inside a geometry; it can have a linear and an angular velocity
    (phi/physics/fluid.py:21-91). Geometries: `Box` / `Cuboid` and `Sphere`. """

    def __init__(self, geometry: Geometry, velocity=0, angular_velocity=0):
        self.geometry = geometry
        D = len(geometry.dims)
        if isinstance(velocity, dict):
            velocity = [velocity.get(d, 0.0) for d in geometry.dims]
        self.velocity = tuple(float(c) for c in velocity) if isinstance(velocity, (tuple, list)) else (float(velocity),) * D
        if isinstance(angular_velocity, dict):
            angular_velocity = [angular_velocity.get(d, 0.0) for d in geometry.dims]
        if isinstance(angular_velocity, (tuple, list)):
            assert D == 3 and len(angular_velocity) == 3, "a vector-valued angular velocity needs a 3-D geometry"
            self.angular_velocity = tuple(float(c) for c in angular_velocity)
        else:
            assert D == 2 or float(angular_velocity) == 0.0, "3-D obstacles rotate about a vector: pass angular_velocity=(wx, wy, wz)"
            self.angular_velocity = (float(angular_velocity), 0.0, 0.0) if D == 2 else (0.0, 0.0, 0.0)

    @property
    def is_rotating(self):
        return any(c != 0 for c in self.angular_velocity)

    @property
    def is_moving(self):
        return any(c != 0 for c in self.velocity)

    @property
    def is_stationary(self):
        return not self.is_moving and not self.is_rotating

    def with_geometry(self, geometry):
        return Obstacle(geometry, self.velocity, self.angular_velocity if len(geometry.dims) == 3 else self.angular_velocity[0])

    def shifted(self, delta):
        return self.with_geometry(self.geometry.shifted(delta))

    def at(self, position):
        return self.with_geometry(self.geometry.at(position))

    def rotated(self, angle):
        return self.with_geometry(self.geometry.rotated(angle))

    def __eq__(self, other):
        return isinstance(other, Obstacle) and repr(self.geometry) == repr(other.geometry) and self.velocity == other.velocity and \
            self.angular_velocity == other.angular_velocity


def _get_obstacles_for(obstacles, velocity: Field) -> List[Obstacle]:
    if obstacles is None:
        return []
    if isinstance(obstacles, (Obstacle, Geometry)):
        obstacles = [obstacles]
    out = []
    for ob in obstacles:
        ob = ob if isinstance(ob, Obstacle) else Obstacle(ob)
        assert set(ob.geometry.dims) == set(velocity.dims), \
            f"Obstacles must live in the same physical space as the velocity field {velocity.dims} but got {ob.geometry.dims}"
        out.append(ob)
    return out


def _obstacle_batch(obstacles: Sequence[Obstacle]) -> int:
    """ batch size of the obstacle geometries (1: the same obstacles for every batch entry) """
    sizes = {ob.geometry.batch_size for ob in obstacles} - {1}
    assert len(sizes) <= 1, f"obstacle geometries have different batch sizes {sizes}"
    return sizes.pop() if sizes else 1


def _obstacle_array(obstacles: Sequence[Obstacle], velocity: Field, entry: int = 0):
    """ (ctypes array of `phihip_obstacle` in the velocity's dimension order, entry count) for batch entry `entry` of batched
    geometries; a `union` geometry becomes one group of consecutive entries """
    items = []
    group = 0
    for ob in obstacles:
        geometry = ob.geometry.entry(entry)
        members = geometry.geometries if isinstance(geometry, Union_) else (geometry,)
        if len(members) > 1:
            group += 1
            if ob.is_rotating:
                raise NotImplementedError("HIP backend: a union obstacle cannot have an angular velocity")
        for geo in members:
            embed_mask = 0
            if isinstance(geo, Embedded):   # infinitely long along the dims the inner geometry lacks
                if ob.is_rotating:
                    raise NotImplementedError("HIP backend: an embedded geometry (embed / infinite_cylinder) cannot have an angular velocity")
                embed_mask = sum(1 << i for i, d in enumerate(velocity.dims) if d not in geo.geometry.dims)
                inner = geo.geometry
                pad = lambda values, fill: [values[inner.dims.index(d)] if d in inner.dims else fill for d in velocity.dims]
                if isinstance(inner, Sphere):
                    geo = Sphere(inner.radius, **dict(zip(velocity.dims, pad(inner.center, 0.0))))
                elif isinstance(inner, Box) and inner.rot is None:
                    geo = Box(**{d: (l, u) for d, l, u in zip(velocity.dims, pad(inner.lower, 0.0), pad(inner.upper, 0.0))})
                else:
                    raise NotImplementedError(f"HIP backend: embed({type(inner).__name__}) is not supported as an obstacle")
            order = [geo.dims.index(d) for d in velocity.dims]
            if isinstance(geo, Sphere):
                kind, half = _capi.OBSTACLE_SPHERE, [geo.radius] * len(order)
            elif isinstance(geo, Box):
                kind, half = _capi.OBSTACLE_BOX, [geo.half_size[i] for i in order]
            else:
                raise NotImplementedError(f"HIP backend: obstacle geometry {type(geo).__name__} is not supported (Box / Cuboid / Sphere / union)")
            vel_order = [ob.geometry.dims.index(d) for d in velocity.dims]
            ang = ob.angular_velocity if len(order) == 2 else [ob.angular_velocity[i] for i in vel_order]
            rot = None
            if isinstance(geo, Box) and geo.rot is not None:
                rot = [[geo.rot[i][j] for j in order] for i in order]
            items.append(dict(kind=kind, center=[geo.center[i] for i in order], half_size=half, velocity=[ob.velocity[i] for i in vel_order],
                              angular_velocity=ang, rotation=rot, group=group if len(members) > 1 else 0, embed_mask=embed_mask))
    return _capi.make_obstacles(items), len(items)


class _MaskCache:
    """ packed stencil flags depend only on (grid, boundary, obstacle geometry): rasterise once per configuration, on the device """

    def __init__(self):
        self.entries = {}

    def get(self, velocity: Field, obstacles: Sequence[Obstacle], user_active: Optional[Field]):
        key = (repr(velocity.resolution), repr(velocity.bounds), repr(velocity.boundary), tuple(repr(o.geometry) for o in obstacles),
               id(user_active) if user_active is not None else None, id(velocity.backend))
        if key not in self.entries or user_active is not None:
            if len(self.entries) > 64:
                self.entries.clear()   # moving obstacles produce a new key every step
            self.entries[key] = _build_flags(velocity, obstacles, user_active)
        return self.entries[key]


_MASKS = _MaskCache()


def _build_flags(velocity: Field, obstacles: Sequence[Obstacle], user_active: Optional[Field]) -> torch.Tensor:
    """ accessible = ~union(obstacles) at the cell centres and hard_bcs / active packed into one byte per cell -- both kernels
    (the reference evaluates this part `with NUMPY:`, fluid.py:130-136) """
    be = velocity.backend
    res = tuple(velocity.resolution.values())
    grid1 = velocity.grid_struct(batch=1)
    accessible_t = None
    OB = _obstacle_batch(obstacles)
    if obstacles:
        accessible_t = be.empty((OB,) + res, torch.uint8)
        for b in range(OB):   # batched geometries (Batched_Smoke.ipynb): one mask per batch entry
            be.ctx.obstacle_accessible(grid1, *_obstacle_array(obstacles, velocity, b), accessible_t[b].data_ptr(), be.stream())
    active_t = None
    AB = 1
    if user_active is not None:
        assert user_active.is_centered and user_active.resolution == velocity.resolution
        active_t = (user_active.values != 0).to(torch.uint8).contiguous()       # (batch | 1, *res)
        AB = active_t.shape[0]
    FB = max(OB, AB)      # per-batch masks (batched geometries, Batched_Smoke.ipynb, or a batched `active` field)
    if FB > 1:
        assert OB in (1, FB) and AB in (1, FB), f"obstacle batch {OB} and `active` batch {AB} do not match"
        if accessible_t is not None and OB == 1:
            accessible_t = accessible_t.expand(FB, *res).contiguous()
        if active_t is not None and AB == 1:
            active_t = active_t.expand(FB, *res).contiguous()
        flags = be.empty((FB,) + res, torch.uint8)
        be.ctx.build_cellflags(velocity.grid_struct(batch=FB), accessible_t.data_ptr() if accessible_t is not None else 0,
                               active_t.data_ptr() if active_t is not None else 0, FB, flags.data_ptr(), be.stream())
        return flags
    flags = be.empty(res, torch.uint8)
    be.ctx.build_cellflags(grid1, accessible_t.data_ptr() if accessible_t is not None else 0,
                           active_t.data_ptr() if active_t is not None else 0, 1, flags.data_ptr(), be.stream())
    return flags


def make_incompressible(velocity: Field,
                        obstacles: Union[Obstacle, Geometry, tuple, list] = (),
                        solve: Solve = Solve(),
                        active: Optional[Field] = None,
                        order: int = 2,
                        correct_skew=False,
                        wide_stencil: bool = None) -> Tuple[Field, Field]:
    """
    Projects the given velocity field by solving for the pressure and subtracting its spatial_gradient
    (phi/physics/fluid.py:94-162).

    Args:
        velocity: `StaggeredGrid`.
        obstacles: `Obstacle` or `Geometry` or tuple/list thereof (Box / Cuboid / Sphere; stationary, moving or rotating).
        solve: `Solve` object specifying tolerances, `x0` (pressure guess) and `max_iterations`.
        active: (Optional) `CenteredGrid` mask for which cells the pressure should be solved. If given, the total
            divergence is never subtracted, even if all values are 1.
        order: only 2 is implemented on the HIP backend.

    Returns:
        velocity: divergence-free velocity of type `type(velocity)`
        pressure: solved pressure field, `CenteredGrid`
    """
    assert not correct_skew
    if order != 2:
        raise NotImplementedError("HIP backend: make_incompressible implements order=2 only")
    if not velocity.is_staggered or wide_stencil:
        raise NotImplementedError("HIP backend: make_incompressible implements the StaggeredGrid path (wide_stencil=False) only")
    if solve.method not in Solve.METHODS:
        raise NotImplementedError(f"HIP backend: Solve(method={solve.method!r}) is not available, use one of {tuple(Solve.METHODS)}")
    obstacles = _get_obstacles_for(obstacles, velocity)
    be = velocity.backend
    OB = _obstacle_batch(obstacles)
    MB = max(OB, active.batch_size if active is not None else 1)     # batch size of the masks
    if MB > 1:
        assert velocity.batch_size in (1, MB), f"velocity batch {velocity.batch_size} does not match the batch {MB} of the obstacles / `active`"
        if velocity.batch_size == 1:   # the same velocity meets a different obstacle / mask in every batch entry
            velocity = Field(velocity.resolution, velocity.bounds, velocity.boundary,
                             [t.expand(MB, *t.shape[1:]).contiguous() for t in velocity.values], True, be, True)
    mask_batch = velocity.batch_size if MB > 1 else 1
    all_active = active is None
    flags = None
    if obstacles or active is not None:
        flags = _MASKS.get(velocity, obstacles, active)
    fp64 = velocity.dtype == torch.float64
    solve = solve.with_defaults(fp64)
    balance = (not velocity.boundary.is_flexible) and all_active   # fluid.py:145
    # `balance` argument of the C ABI: PHIHIP_DIV_BALANCE | PHIHIP_DIV_FINITE_GUARD -- "if not all_active: div = where(is_finite(div), div, 0)"
    # (fluid.py:143-144: with a user-supplied `active` the velocity may hold NaN where it does not contribute to the pressure)
    div_bits = (_capi.DIV_BALANCE if balance else 0) | (0 if all_active else _capi.DIV_FINITE_GUARD)
    p_ext = pressure_extrapolation(velocity.boundary, velocity.dims)
    B = velocity.batch_size
    res_shape = tuple(velocity.resolution.values())
    if solve.x0 is None:
        pressure = be.zeros((B,) + res_shape, velocity.dtype)
    else:
        x0 = solve.x0
        assert isinstance(x0, Field) and x0.is_centered and same_grid(x0, velocity), "x0 must be a CenteredGrid on the velocity's grid"
        _check_pressure_padding(x0.boundary, velocity.boundary, velocity.dims)
        pressure = x0.values.to(velocity.dtype)
        pressure = (pressure.expand(B, *res_shape) if pressure.shape[0] != B else pressure).clone().contiguous()
    csolve = solve.to_c(fp64)
    from .jit import is_tracing
    traced = is_tracing()       # inside a jit_compile'd function: no host read-back (jit.py) -- info = NULL, the host never polls the continue flags
    if traced:
        csolve.check_every = 0
        if autodiff.needs_grad(*velocity.values):
            raise NotImplementedError("HIP backend: gradients through a jit_compile'd function are not implemented (capture the forward step or differentiate it, not both)")
    if autodiff.needs_grad(*velocity.values):
        # differentiable path: the same kernels behind torch.autograd.Function nodes (adjoint kernels in csrc/adjoint.hip)
        vin = [t.contiguous() for t in velocity.values]
        shapes = [tuple(t.shape) for t in vin]
        if MB > 1:
            raise NotImplementedError("HIP backend: gradients through batched obstacle geometries / `active` masks are not implemented")
        if obstacles:
            vin = list(_apply_obstacles_autograd(velocity, obstacles, vin))
        gsolve = solve.gradient_solve if getattr(solve, 'gradient_solve', None) is not None else solve
        csolve_bwd = gsolve.to_c(fp64)
        meta = dict(be=be, grid=velocity.grid_struct(), flags_ptr=flags.data_ptr() if flags is not None else 0, flags=flags, balance=div_bits,
                    csolve=csolve, csolve_bwd=csolve_bwd, shapes=shapes, dtype=velocity.dtype)
        *new_v, pressure = autodiff.MakeIncompressible.apply(meta, pressure.detach(), *vin)
        new_v = list(new_v)
        infos = meta['infos']
    else:
        new_v = [t.clone() for t in velocity.values]
        if obstacles:   # v = apply_boundary_conditions(v, obstacles)   (fluid.py:137)
            _apply_obstacles_in_place(velocity, obstacles, new_v)
        infos = be.ctx.make_incompressible(velocity.grid_struct(), _ptrs(new_v), None,
                                           flags.data_ptr() if flags is not None else 0, mask_batch, div_bits, pressure.data_ptr(), 0, csolve,
                                           not traced, be.stream())
    info = None
    if infos is not None:
        info = SolveInfo(solve, [i.iterations for i in infos], [i.residual_sq for i in infos], [i.rhs_sq for i in infos],
                         [bool(i.converged) for i in infos], [bool(i.diverged) for i in infos])
        _raise_if_failed(info)
    v_out = Field(velocity.resolution, velocity.bounds, velocity.boundary, new_v, True, be, velocity.batched)
    p_out = Field(velocity.resolution, velocity.bounds, p_ext, pressure, False, be, velocity.batched)
    p_out.solve_info = info
    return v_out, p_out


def _apply_obstacles_in_place(velocity: Field, obstacles, values: List[torch.Tensor]):
    """ apply_boundary_conditions on contiguous component tensors; batched geometries: entry b of the batch meets obstacle entry b """
    be = velocity.backend
    if _obstacle_batch(obstacles) == 1:
        be.ctx.apply_obstacles(velocity.grid_struct(), *_obstacle_array(obstacles, velocity), _ptrs(values), be.stream())
        return
    grid1 = velocity.grid_struct(batch=1)
    for b in range(velocity.batch_size):
        be.ctx.apply_obstacles(grid1, *_obstacle_array(obstacles, velocity, b), [t[b:b + 1].data_ptr() for t in values], be.stream())


def _apply_obstacles_autograd(velocity: Field, obstacles, vin):
    still = [Obstacle(ob.geometry) for ob in obstacles]
    arr, count = _obstacle_array(obstacles, velocity)
    meta = dict(be=velocity.backend, grid=velocity.grid_struct(), obstacles=arr, obstacles_still=_obstacle_array(still, velocity)[0], count=count,
                shapes=[tuple(t.shape) for t in vin], dtype=velocity.dtype)
    return autodiff.ApplyObstacles.apply(meta, *vin)


def _raise_if_failed(info: SolveInfo):
    """ phiml.math.solve_linear raises Diverged / NotConverged unless suppressed """
    suppress = tuple(info.solve.suppress or ())
    if any(info.diverged):
        info.msg = f"CG diverged (residual^2 {info.residual_sq}, rhs^2 {info.rhs_sq}, iterations {info.iterations})"
        if Diverged not in suppress:
            raise Diverged(info)
    elif not all(info.converged):
        info.msg = f"CG did not converge to rel_tol={info.solve.rel_tol}, abs_tol={info.solve.abs_tol} within " \
                   f"{info.solve.max_iterations} iterations (residual^2 {info.residual_sq}, rhs^2 {info.rhs_sq})"
        if NotConverged not in suppress:
            raise NotConverged(info)


def apply_boundary_conditions(velocity: Field, obstacles) -> Field:
    """ Enforces velocity boundary conditions on a velocity grid (phi/physics/fluid.py:212-240): cells inside obstacles get their
    velocity from the obstacle movement (linear + angular), cells far away are unaffected; soft transition over one cell. """
    obstacles = _get_obstacles_for(obstacles, velocity)
    if not obstacles:
        return velocity
    be = velocity.backend
    OB = _obstacle_batch(obstacles)
    if autodiff.needs_grad(*velocity.values):
        if OB > 1:
            raise NotImplementedError("HIP backend: gradients through batched obstacle geometries are not implemented")
        return velocity.with_values(list(_apply_obstacles_autograd(velocity, obstacles, [t.contiguous() for t in velocity.values])))
    if OB > 1 and velocity.batch_size == 1:
        velocity = Field(velocity.resolution, velocity.bounds, velocity.boundary,
                         [t.expand(OB, *t.shape[1:]).contiguous() for t in velocity.values], True, be, True)
    new_v = [t.clone().contiguous() for t in velocity.values]
    _apply_obstacles_in_place(velocity, obstacles, new_v)
    return velocity.with_values(new_v)


def masked_laplace(pressure: Field, v_boundary, hard_bcs=None, active=None, flags: Optional[torch.Tensor] = None) -> Field:
    """ `fluid.masked_laplace` (phi/physics/fluid.py:165-202) applied matrix-free. `flags` = packed obstacle flags from
    `phihip_build_cellflags` (replaces the reference's `hard_bcs` / `active` fields). """
    if hard_bcs is not None or active is not None:
        raise NotImplementedError("pass packed `flags` instead of hard_bcs / active fields on the HIP backend")
    from .extrapolation import as_extrapolation
    vb = as_extrapolation(v_boundary)
    _check_pressure_padding(pressure.boundary, vb, pressure.dims)
    be = pressure.backend
    proto = Field(pressure.resolution, pressure.bounds, vb, None, True, be, pressure.batched)
    grid = _capi.make_grid(pressure.spatial_rank, _capi.PHIHIP_F64 if pressure.dtype == torch.float64 else _capi.PHIHIP_F32,
                           pressure.batch_size, list(pressure.resolution.values()), pressure.bounds.lower, pressure.bounds.upper,
                           proto._codes, proto._bc_val)
    out = torch.empty_like(pressure.values)
    be.ctx.laplace_apply(grid, flags.data_ptr() if flags is not None else 0, 1, pressure.values.contiguous().data_ptr(), out.data_ptr(),
                         be.stream())
    return pressure.with_values(out)


from .linear import _balance_divergence   # noqa: E402,F401  (fluid._balance_divergence, phi/physics/fluid.py:205-209)
