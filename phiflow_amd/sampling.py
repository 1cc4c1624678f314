"""
Sampling grid fields at arbitrary points -- the general form of `field.sample` / `resample` / `advect.semi_lagrangian` for fields that
do NOT share one grid (reference: phi/field/_resample.py:66-72,145-161,241-259 `sample` -> `grid_sample`; phi/physics/advect.py:156-215;
examples/grids/Batched_Smoke.ipynb advects a 200^2 smoke field by a 64^2 velocity).

Same-grid fields take the fused kernels (advect.py, field.resample). Here every gather is one `phihip_grid_sample` launch
(csrc/advect.hip, the same tap resolution as the fused kernels) and the coordinate arithmetic between the gathers is elementwise
torch glue on device tensors.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi
from .extrapolation import ConstantExtrapolation, resolve
from .field import Field, _ptrs, _sample_points, _torch_dtype_code, component_shape, same_grid   # noqa: F401


class GridSample(torch.autograd.Function):
    """ out[b, i] = multilinear(values[b], coords[:][b, i]); backward through `phihip_grid_sample_backward` """

    @staticmethod
    def forward(ctx, meta, values, *coords):
        be, grid = meta['be'], meta['grid']
        out = torch.empty_like(coords[0])
        be.ctx.grid_sample(grid, values.data_ptr(), values.shape[0], _ptrs(coords), coords[0].shape[1], out.data_ptr(), 0, 0, be.stream())
        ctx.meta = meta
        ctx.save_for_backward(values, *coords)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        values, *coords = ctx.saved_tensors
        be, grid = ctx.meta['be'], ctx.meta['grid']
        g = grad_out.contiguous()
        gv = torch.zeros_like(values) if ctx.needs_input_grad[1] else None
        gc = [torch.zeros_like(c) for c in coords] if any(ctx.needs_input_grad[2:]) else None
        be.ctx.grid_sample_backward(grid, values.data_ptr(), values.shape[0], _ptrs(coords), coords[0].shape[1], g.data_ptr(),
                                    gv.data_ptr() if gv is not None else 0, _ptrs(gc) if gc is not None else None, be.stream())
        return (None, gv, *(gc if gc is not None else [None] * len(coords)))


def _array_rule(field: Field, comp: Optional[int]):
    """ (codes[D][2], consts[D][2]) of the array that holds a centred field (comp None) or component `comp` of a staggered one """
    codes, vals = resolve(field.boundary, field.dims)
    c = 0 if comp is None else comp
    consts = [[vals[a][s][c] if isinstance(field.boundary.side(d, bool(s)), ConstantExtrapolation) else 0.0 for s in range(2)]
              for a, d in enumerate(field.dims)]
    return codes, consts


def sample_array(field: Field, comp: Optional[int], coords: Sequence[torch.Tensor], limits: bool = False):
    """ gathers from the array of `field` (component `comp` if staggered) at fractional index coordinates coords[d] of shape (B, points);
    `limits`: also the min / max over the taps (Field.closest_values). Values of batch 1 are shared by all B coordinate sets. """
    be = field.backend
    values = (field.values if comp is None else field.values[comp]).contiguous()
    coords = [c.contiguous() for c in coords]
    B, npts = coords[0].shape
    assert values.shape[0] in (1, B), f"values batch {values.shape[0]} vs coordinates batch {B}"
    codes, consts = _array_rule(field, comp)
    D = field.spatial_rank
    grid = _capi.make_grid(D, _torch_dtype_code(values.dtype), B, list(values.shape[1:]), (0.0,) * D, (1.0,) * D, codes,
                           [[[consts[a][s], 0.0, 0.0] for s in range(2)] for a in range(D)])
    if limits:
        out, lo, hi = (torch.empty_like(coords[0]) for _ in range(3))
        be.ctx.grid_sample(grid, values.data_ptr(), values.shape[0], _ptrs(coords), npts, out.data_ptr(), lo.data_ptr(), hi.data_ptr(), be.stream())
        return out, lo, hi
    if values.requires_grad or any(c.requires_grad for c in coords):
        return GridSample.apply(dict(be=be, grid=grid), values, *coords)
    out = torch.empty_like(coords[0])
    be.ctx.grid_sample(grid, values.data_ptr(), values.shape[0], _ptrs(coords), npts, out.data_ptr(), 0, 0, be.stream())
    return out


def sample_points(field: Field, comp: Optional[int] = None) -> List[torch.Tensor]:
    """ world coordinates of the field's samples (cell centres, or the stored faces of component `comp`): D tensors of shape (1, points) """
    pts = _sample_points(field.resolution, field.bounds, comp, field.boundary)
    return [torch.as_tensor(np.ascontiguousarray(p).reshape(1, -1), dtype=field.dtype, device=field.backend.device) for p in pts]


def index_coords(field: Field, comp: Optional[int], points: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """ world coordinates -> fractional indices into the array of `field` / its component `comp`
    (`bounds.global_to_local(points) * resolution - 0.5` on the (sub-)grid, phi/field/_resample.py:257-258) """
    out = []
    for a, dim in enumerate(field.dims):
        dx = field.dx[a]
        first = 0.5
        if comp is not None and a == comp:
            lo, _ = field.boundary.valid_outer_faces(dim)
            first = 0.0 if lo else 1.0          # index 0 is the lower boundary face, or the first interior face
        out.append((points[a] - field.bounds.lower[a]) / dx - first)
    return out


def sample_field(field: Field, points: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """ the field's value at world points (B, n): [scalar] for a centred field, one tensor per component for a staggered one
    (`sample(velocity, geometry)`: every component interpolated on its own staggered sub-grid) """
    comps = [None] if field.is_centered else range(field.spatial_rank)
    B = max([p.shape[0] for p in points] + [field.batch_size])
    pts = [p if p.shape[0] == B else p.expand(B, -1) for p in points]
    return [sample_array(field, c, index_coords(field, c, pts)) for c in comps]




def resample_general(value: Field, to: Field) -> Field:
    """ `resample(value, to)` between different grids: centred -> centred / staggered faces (times the lazy constant vector of
    `scalar * (0, 0.1)`), staggered -> staggered (component-wise). """
    be = to.backend
    assert value.dims == to.dims, f"fields live in different spaces: {value.dims} vs {to.dims}"
    if value.is_staggered and to.is_centered:
        raise NotImplementedError("HIP backend: resampling a StaggeredGrid to cell centres would need centred vector fields")
    batched = value.batched          # the target only lends its sample points
    if to.is_centered:
        out = sample_field(value, sample_points(to))[0]
        return Field(to.resolution, to.bounds, to.boundary, out.reshape(out.shape[0], *to.resolution.values()), False, be, batched)
    scale = value.vector_scale or [1.0] * value.spatial_rank
    comps = []
    for d in range(to.spatial_rank):
        pts = sample_points(to, d)
        if value.is_centered:
            t = sample_field(value, pts)[0] * scale[d]
        else:
            B = value.batch_size
            t = sample_array(value, d, index_coords(value, d, [p.expand(B, -1) for p in pts]))
        comps.append(t.reshape(t.shape[0], *component_shape(to.resolution, to.boundary, d)))
    B = max(c.shape[0] for c in comps)
    comps = [c if c.shape[0] == B else c.expand(B, *c.shape[1:]).contiguous() for c in comps]
    return Field(to.resolution, to.bounds, to.boundary, comps, True, be, batched)


def integrate_points(points: Sequence[torch.Tensor], velocity: Field, dt: float, integrator: str) -> List[torch.Tensor]:
    """ `advect.euler` / `advect.rk4` / `advect.finite_rk4` (phi/physics/advect.py:20-47): where the points end up after dt """
    v0 = sample_field(velocity, points)
    if integrator == 'euler':
        return [p + dt * u for p, u in zip(points, v0)]
    v_half = sample_field(velocity, [p + (0.5 * dt) * u for p, u in zip(points, v0)])
    v_half2 = sample_field(velocity, [p + (0.5 * dt) * u for p, u in zip(points, v_half)])
    v_full = sample_field(velocity, [p + dt * u for p, u in zip(points, v_half2)])
    v_rk4 = [(1 / 6.) * (a + 2 * (b + c) + d) for a, b, c, d in zip(v0, v_half, v_half2, v_full)]
    if integrator == 'finite_rk4':      # Euler fallback where the RK4 velocity is not finite
        v_rk4 = [torch.where(torch.isfinite(u), u, u0) for u, u0 in zip(v_rk4, v0)]
    return [p + dt * u for p, u in zip(points, v_rk4)]


def advect_general(field: Field, velocity: Field, dt: float, correction_strength: Optional[float], integrator: str = 'euler') -> Field:
    """ semi-Lagrangian / MacCormack advection of `field` by a velocity sampled on another grid and / or with a Runge-Kutta back-trace
    (phi/physics/advect.py:156-215): `lookup = integrator(x, velocity, -dt)`, `new = field(lookup)`; MacCormack adds the backward pass
    and clamps to the min / max of the field's values around the lookup. """
    assert field.dims == velocity.dims and field.dtype == velocity.dtype
    if correction_strength is not None and field.is_staggered:
        raise NotImplementedError("HIP backend: MacCormack advection of a StaggeredGrid needs the euler integrator and the velocity on the same grid")
    comps = [None] if field.is_centered else list(range(field.spatial_rank))
    B = max(field.batch_size, velocity.batch_size)
    outs = []
    for c in comps:
        pts = [p.expand(B, -1) for p in sample_points(field, c)]
        back = integrate_points(pts, velocity, -dt, integrator)
        if correction_strength is None:
            new = sample_array(field, c, index_coords(field, c, back))
        else:
            coords_back = index_coords(field, c, back)
            needs_grad = field.values.requires_grad or any(t.requires_grad for t in coords_back)
            with torch.no_grad():
                fwd_vals, lo, hi = sample_array(field, c, [t.detach() for t in coords_back], limits=True)
            if needs_grad:   # differentiable gather; the clamp window stays constant (a clamped sample gets no gradient here)
                fwd_vals = sample_array(field, c, coords_back)
            fwd = Field(field.resolution, field.bounds, field.boundary, fwd_vals.reshape(B, *field.resolution.values()), False, field.backend, True)
            ahead = integrate_points(pts, velocity, dt, integrator)
            bwd = sample_array(fwd, None, index_coords(fwd, None, ahead))
            own = (field.values if field.values.shape[0] == B else field.values.expand(B, *field.values.shape[1:])).reshape(B, -1)
            new = fwd_vals + (0.5 * correction_strength) * (own - bwd)
            new = torch.minimum(torch.maximum(new, lo), hi)               # math.clip(new, min, max)
        shape = tuple(field.resolution.values()) if c is None else component_shape(field.resolution, field.boundary, c)
        outs.append(new.reshape(B, *shape))
    batched = field.batched or velocity.batched
    return Field(field.resolution, field.bounds, field.boundary, outs[0] if field.is_centered else outs, field.is_staggered, field.backend, batched)


def backend_grid_sample(be, grid: torch.Tensor, coordinates: torch.Tensor, extrapolation: str):
    """ PhiML's `Backend.grid_sample(grid, coordinates, extrapolation)` on natives (reference call site phi/field/_resample.py:259
    `math.grid_sample`): `grid` (batch, x, y[, z], channels), `coordinates` (batch, *points, D) as fractional
    indices, extrapolation 'periodic' | 'boundary' | 'zeros'. One `phihip_grid_sample` launch per channel. Returns None for anything
    else (the caller falls back to its generic gather). """
    rule = {'periodic': (_capi.BC_PERIODIC, 0.0), 'boundary': (_capi.BC_OPEN, 0.0), 'zeros': (_capi.BC_CLOSED, 0.0)}.get(extrapolation)
    D = coordinates.shape[-1]
    if rule is None or D not in (2, 3) or grid.dim() != D + 2 or grid.dtype not in (torch.float32, torch.float64):
        return None
    code, const = rule
    B = max(grid.shape[0], coordinates.shape[0])
    pts_shape = tuple(coordinates.shape[1:-1])
    npts = int(np.prod(pts_shape)) if pts_shape else 1
    coords = coordinates.to(device=be.device, dtype=grid.dtype).reshape(coordinates.shape[0], npts, D)
    coords = coords.expand(B, npts, D)
    cs = [coords[..., d].contiguous() for d in range(D)]
    res = list(grid.shape[1:-1])
    g = _capi.make_grid(D, _torch_dtype_code(grid.dtype), B, res, (0.0,) * D, (1.0,) * D, ((code, code),) * D,
                        [[[const, 0.0, 0.0] for _ in range(2)] for _ in range(D)])
    out = torch.empty((B, npts, grid.shape[-1]), dtype=grid.dtype, device=be.device)
    for c in range(grid.shape[-1]):
        values = grid[..., c].to(be.device).contiguous()
        oc = torch.empty((B, npts), dtype=grid.dtype, device=be.device)
        be.ctx.grid_sample(g, values.data_ptr(), values.shape[0], _ptrs(cs), npts, oc.data_ptr(), 0, 0, be.stream())
        out[..., c] = oc
    return out.reshape((B,) + pts_shape + (grid.shape[-1],))
