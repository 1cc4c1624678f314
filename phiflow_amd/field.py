"""
`Field` / `CenteredGrid` / `StaggeredGrid`: the host-side mirror of PhiFlow's grid fields
(reference: phi/field/_field.py:51-211, phi/field/_grid.py:21-176, phi/geom/_grid.py:41-122,204-209).

Only uniform grids are supported. A field owns device tensors (torch, ROCm) laid out exactly as the C ABI wants them:
  * CenteredGrid  -> one tensor (batch, x, y[, z])
  * StaggeredGrid -> one tensor per component d with shape (batch, *res + (lo+up-1) e_d)   (faces stored iff
    `boundary.valid_outer_faces(d)`, tests/commit/field/test__grid.py:25-36)
Fields are immutable from the user's point of view: every operator returns new fields (phi/field/_field.py:49-51).
PhiML's named batch dims are reduced to ONE leading batch dimension (size 1 when the field is not batched).
"""
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _capi
from .backend import HipBackend, default_backend, float_dtype
from .extrapolation import BOUNDARY, PERIODIC, ZERO, ConstantExtrapolation, Extrapolation, as_extrapolation, resolve
from .geom import Box, Geometry


def _torch_dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return _capi.PHIHIP_F32
    if dtype == torch.float64:
        return _capi.PHIHIP_F64
    raise TypeError(f"unsupported dtype {dtype}; the HIP backend computes in float32 or float64")


_GRID_CACHE = {}      # (id of the resolved boundary codes, dtype, batch, resolution, lower, upper) -> (phihip_grid, codes): Field.grid_struct


class Field:
    """ A sampled scalar (centred) or vector (staggered) grid field with boundary conditions. """
    __array_ufunc__ = None     # `numpy_array * field` defers to Field.__rmul__ (a batch vector, one number per batch entry)
    solve_info = None          # set on the pressure that a solve returns (solve.SolveInfo); None on every other field and on results of a captured function (jit.py)

    def __init__(self, resolution: Dict[str, int], bounds: Box, boundary: Extrapolation, values, staggered: bool,
                 backend: HipBackend, batched: bool, vector_scale: Optional[Sequence[float]] = None):
        # vector_scale: `scalar * (0, 0.1)` -- a centred scalar times a constant vector, kept lazily (values stay the scalar's) until
        # it is resampled to a staggered grid with `@` / `resample`. Arithmetic and resample carry it along; every other consumer of
        # `values` (mean, gradients, divergence, advection, diffusion, losses, file output) calls `require_plain` and refuses
        assert vector_scale is None or (not staggered and len(vector_scale) == len(resolution))
        self.vector_scale = [float(c) for c in vector_scale] if vector_scale is not None else None
        self.resolution = dict(resolution)
        self._dims = tuple(self.resolution)
        self.bounds = bounds
        self.boundary = boundary
        self.values = values           # Tensor (centred) or List[Tensor] (staggered)
        self.is_staggered = staggered
        self.backend = backend
        self.batched = batched
        self._codes, self._bc_val = resolve(boundary, self.dims)

    # --- geometry -------------------------------------------------------------------------------------------------
    @property
    def dims(self) -> Tuple[str, ...]:
        return self._dims

    @property
    def spatial_rank(self) -> int:
        return len(self.resolution)

    @property
    def extrapolation(self) -> Extrapolation:
        return self.boundary

    @property
    def is_centered(self) -> bool:
        return not self.is_staggered

    @property
    def is_grid(self) -> bool:
        return True

    @property
    def dx(self) -> Tuple[float, ...]:
        return tuple(s / r for s, r in zip(self.bounds.size, self.resolution.values()))

    @property
    def dtype(self) -> torch.dtype:
        return (self.values[0] if self.is_staggered else self.values).dtype

    @property
    def batch_size(self) -> int:
        return int((self.values[0] if self.is_staggered else self.values).shape[0])

    @property
    def shape(self):
        """ (batch, *resolution) -- for staggered fields the per-component shapes are in `component_shapes` """
        return (self.batch_size,) + tuple(self.resolution.values())

    @property
    def component_shapes(self) -> List[Tuple[int, ...]]:
        return [component_shape(self.resolution, self.boundary, d) for d in range(self.spatial_rank)]

    def grid_struct(self, batch: Optional[int] = None, dtype: Optional[torch.dtype] = None) -> _capi.Grid:
        """ the `phihip_grid` descriptor of this field's grid + boundary """
        # r6: the descriptor is built once per (grid, boundary, dtype, batch) and reused by every Field on it (an eager 128^2 plume step built four per step,
        # 13 us each, out of ~0.2 ms of host work; nothing mutates a Grid after make_grid). `_codes` is the list cached on the extrapolation object (resolve).
        key = (id(self._codes), _torch_dtype_code(dtype or self.dtype), batch or self.batch_size, tuple(self.resolution.values()),
               tuple(self.bounds.lower), tuple(self.bounds.upper))
        hit = _GRID_CACHE.get(key)
        if hit is not None and hit[1] is self._codes:
            return hit[0]
        g = _capi.make_grid(self.spatial_rank, key[1], key[2], key[3], key[4], key[5], self._codes, self._bc_val)
        if len(_GRID_CACHE) >= 512:
            _GRID_CACHE.clear()
        _GRID_CACHE[key] = (g, self._codes)
        return g

    # --- access ---------------------------------------------------------------------------------------------------
    def numpy(self):
        """ host copy; batch dim squeezed when the field is not batched. Staggered: list of component arrays. """
        def conv(t):
            a = t.detach().cpu().numpy()
            return a if self.batched else a[0]
        return [conv(t) for t in self.values] if self.is_staggered else conv(self.values)

    def __getitem__(self, item) -> 'Field':
        """ `v['x']`: component as a centred field on its staggered sub-grid (phi/field/_field.py:657-689) """
        assert self.is_staggered and item in self.dims, f"can only select a vector component of a staggered field, got {item!r}"
        d = self.dims.index(item)
        lo, up = self.boundary.valid_outer_faces(item)
        dx = self.dx[d]
        kw = {dim: (l, u) for dim, l, u in zip(self.dims, self.bounds.lower, self.bounds.upper)}
        kw[item] = (self.bounds.lower[d] + (-0.5 if lo else 0.5) * dx, self.bounds.upper[d] + (0.5 if up else -0.5) * dx)
        res = dict(self.resolution)
        res[item] = self.values[d].shape[1 + d]
        comp_boundary = self.boundary
        return Field(res, Box(**kw), comp_boundary, self.values[d], False, self.backend, self.batched)

    def with_values(self, values) -> 'Field':
        return Field(self.resolution, self.bounds, self.boundary, values, self.is_staggered, self.backend, self.batched, self.vector_scale)

    def with_boundary(self, boundary) -> 'Field':
        """ change the extrapolation; staggered fields re-store their boundary faces accordingly: faces that become
        stored take the OLD boundary value (0 on walls), faces determined by the new boundary are dropped
        (phi/field/_field.py:451-472; tests/commit/field/test__grid.py:85-94). """
        boundary = as_extrapolation(boundary)
        if not self.is_staggered:
            return Field(self.resolution, self.bounds, boundary, self.values, False, self.backend, self.batched, self.vector_scale)
        new_vals = []
        for d, dim in enumerate(self.dims):
            t = self.values[d]
            lo_old, up_old = self.boundary.valid_outer_faces(dim)
            lo_new, up_new = boundary.valid_outer_faces(dim)
            ax = 1 + d
            n = self.resolution[dim]
            # bake to all n+1 faces first
            parts = []
            if not lo_old:
                parts.append(_boundary_slab(t, ax, self._codes[d][0], self._bc_val[d][0][d], lower=True, periodic_src=None))
            parts.append(t)
            if not up_old:
                parts.append(_boundary_slab(t, ax, self._codes[d][1], self._bc_val[d][1][d], lower=False,
                                            periodic_src=parts[0] if lo_old is False else t))
            full = torch.cat(parts, dim=ax) if len(parts) > 1 else t
            assert full.shape[ax] == n + 1
            start = 0 if lo_new else 1
            stop = n + 1 if up_new else n
            new_vals.append(full.narrow(ax, start, stop - start).contiguous())
        return Field(self.resolution, self.bounds, boundary, new_vals, True, self.backend, self.batched)

    with_extrapolation = with_boundary

    # --- arithmetic (elementwise glue on device tensors; not part of the kernel hot path) --------------------------
    def _op(self, other, fn, additive: bool = False) -> 'Field':
        """ elementwise `fn(self, other)`. additive: + / - (a lazy `scalar * vector` field only combines with one of the same vector) """
        scale = self.vector_scale
        if isinstance(other, Field):
            assert other.is_staggered == self.is_staggered and same_grid(self, other), \
                f"incompatible fields (sample points differ): {self!r} vs {other!r}; resample one with `@` first"
            if additive and scale != other.vector_scale:
                raise NotImplementedError("adding a lazy `scalar * vector` field to a field with a different (or no) vector factor; "
                                          "resample it to the staggered grid first (`field @ velocity`)")
            if not additive and other.vector_scale is not None:
                scale = other.vector_scale if scale is None else [a * b for a, b in zip(scale, other.vector_scale)]
            if self.is_staggered:
                assert [tuple(a.shape[1:]) for a in self.values] == [tuple(b.shape[1:]) for b in other.values], \
                    "staggered fields with different face layouts (boundaries) cannot be combined"
                vals = [fn(a, b) for a, b in zip(self.values, other.values)]
            else:
                vals = fn(self.values, other.values)
            batched = self.batched or other.batched
        elif isinstance(other, (tuple, list)):
            assert self.is_staggered and len(other) == self.spatial_rank, "vector operand requires a staggered field"
            vals = [fn(a, float(c)) for a, c in zip(self.values, other)]
            batched = self.batched
        elif additive and scale is not None:
            raise NotImplementedError("adding a number to a lazy `scalar * vector` field; resample it to the staggered grid first")
        elif isinstance(other, (np.ndarray, torch.Tensor)) and other.ndim == 1:
            # one number per batch entry, e.g. `inflow_rate * resample(inflow, to=s, soft=True)` (Batched_Smoke.ipynb)
            assert self.batch_size in (1, len(other)), f"batch vector of length {len(other)} does not match batch size {self.batch_size}"
            w = torch.as_tensor(np.asarray(other) if isinstance(other, np.ndarray) else other, dtype=self.dtype, device=self.backend.device)
            w = w.reshape(-1, *([1] * self.spatial_rank))
            vals = [fn(a, w) for a in self.values] if self.is_staggered else fn(self.values, w)
            batched = True
        else:
            vals = [fn(a, other) for a in self.values] if self.is_staggered else fn(self.values, other)
            batched = self.batched
        return Field(self.resolution, self.bounds, self.boundary, vals, self.is_staggered, self.backend, batched, scale)

    def __add__(self, other): return self._op(other, lambda a, b: a + b, additive=True)
    def __radd__(self, other): return self._op(other, lambda a, b: b + a, additive=True)
    def __sub__(self, other): return self._op(other, lambda a, b: a - b, additive=True)
    def __rsub__(self, other): return self._op(other, lambda a, b: b - a, additive=True)
    def __mul__(self, other):
        if isinstance(other, (tuple, list)) and self.is_centered:
            return vector_scaled(self, other)   # `smoke * (0, 0.1)`; becomes a vector field when resampled with `@`
        return self._op(other, lambda a, b: a * b)

    def __rmul__(self, other): return self.__mul__(other)
    def __truediv__(self, other):
        if isinstance(other, Field) and other.vector_scale is not None:
            raise NotImplementedError("division by a lazy `scalar * vector` field")
        return self._op(other, lambda a, b: a / b)
    def __neg__(self): return self._op(-1.0, lambda a, b: a * b)

    def __matmul__(self, other: 'Field') -> 'Field':
        """ `value @ target`: resample to the sample points of `target`, keeping `target`'s boundary
        (phi/field/_field.py `__matmul__` -> resample). """
        return resample(self, to=other)

    def __repr__(self):
        kind = "StaggeredGrid" if self.is_staggered else "CenteredGrid"
        return f"{kind}[{self.resolution}, batch={self.batch_size if self.batched else None}, {self.boundary}, {self.dtype}, {self.backend}]"


def require_plain(field, what: str):
    """ refuse a lazily vector-scaled scalar (`smoke * (0, 0.1)`) where its `values` would be taken for the field itself: the result would
    silently be that of the plain scalar. Resample it to a staggered grid first (`field @ velocity`). """
    if isinstance(field, Field) and field.vector_scale is not None:
        raise NotImplementedError(f"{what}: the field is a scalar times the constant vector {tuple(field.vector_scale)} (a centred VECTOR field, not "
                                  f"stored as such); resample it to a StaggeredGrid with `field @ velocity` / `resample(field, to=velocity)` first")


def same_grid(a: 'Field', b: 'Field') -> bool:
    """ same sample points up to staggering: resolution, bounds and dims (the reference short-circuits `resample` only when
    `value.geometry == to.geometry`, phi/field/_resample.py:42-48) """
    return a.resolution == b.resolution and a.dims == b.dims and tuple(a.bounds.lower) == tuple(b.bounds.lower) \
        and tuple(a.bounds.upper) == tuple(b.bounds.upper)


def _boundary_slab(t: torch.Tensor, ax: int, code: int, const: float, lower: bool, periodic_src) -> torch.Tensor:
    """ one layer of boundary faces along tensor axis `ax` for a normal velocity component """
    if code == _capi.BC_CLOSED:
        shp = list(t.shape)
        shp[ax] = 1
        return torch.full(shp, const, dtype=t.dtype, device=t.device)
    if code == _capi.BC_PERIODIC:
        return t.narrow(ax, 0, 1) if not lower else t.narrow(ax, t.shape[ax] - 1, 1)
    return t.narrow(ax, 0 if lower else t.shape[ax] - 1, 1)


def component_shape(resolution: Dict[str, int], boundary: Extrapolation, d: int) -> Tuple[int, ...]:
    dims = list(resolution.keys())
    lo, up = boundary.valid_outer_faces(dims[d])
    shape = list(resolution.values())
    shape[d] += int(lo) + int(up) - 1
    return tuple(shape)


def _resolve_grid_args(bounds, resolution, resolution_):
    res = dict(resolution or {}, **resolution_)
    assert res, "resolution must be given, e.g. x=64, y=64"
    res = {d: int(r) for d, r in res.items()}
    if bounds is None:
        bounds = Box(**{d: float(r) for d, r in res.items()})   # default: dx = 1 (phi/field/_grid.py:55-60)
    else:
        assert set(bounds.dims) == set(res.keys()), f"bounds {bounds} do not match resolution {res}"
        bounds = Box(**{d: (bounds.lower[bounds.dims.index(d)], bounds.upper[bounds.dims.index(d)]) for d in res})
    return res, bounds


def _sample_points(res: Dict[str, int], bounds: Box, comp: Optional[int], boundary: Extrapolation):
    """ numpy float64 coordinate arrays of cell centres (comp None) or of the stored faces of component `comp` """
    axes = []
    dims = list(res.keys())
    for a, dim in enumerate(dims):
        n = res[dim]
        dx = (bounds.upper[a] - bounds.lower[a]) / n
        if comp is not None and a == comp:
            lo, up = boundary.valid_outer_faces(dim)
            cnt = n + int(lo) + int(up) - 1
            first = 0 if lo else 1
            axes.append(bounds.lower[a] + (first + np.arange(cnt)) * dx)
        else:
            axes.append(bounds.lower[a] + (np.arange(n) + 0.5) * dx)
    return np.meshgrid(*axes, indexing='ij')


def _to_batched(array: np.ndarray, spatial_shape: Tuple[int, ...]) -> Tuple[np.ndarray, bool]:
    array = np.asarray(array)
    if array.shape == tuple(spatial_shape):
        return array[None], False
    if array.ndim == len(spatial_shape) + 1 and array.shape[1:] == tuple(spatial_shape):
        return array, True
    if array.ndim == 0:
        return np.broadcast_to(array, (1,) + tuple(spatial_shape)), False
    raise ValueError(f"values of shape {array.shape} do not match the expected sample shape {spatial_shape} (optionally with a leading batch dim)")


def _tensor_from(value, spatial_shape, backend: HipBackend, dtype) -> Tuple[torch.Tensor, bool]:
    if isinstance(value, torch.Tensor):
        t = value.to(device=backend.device, dtype=dtype)
        if tuple(t.shape) == tuple(spatial_shape):
            return t.unsqueeze(0).contiguous(), False
        assert t.dim() == len(spatial_shape) + 1 and tuple(t.shape[1:]) == tuple(spatial_shape), \
            f"tensor of shape {tuple(t.shape)} does not match sample shape {spatial_shape}"
        return t.contiguous(), True
    arr, batched = _to_batched(value, spatial_shape)
    return backend.as_tensor(np.ascontiguousarray(arr), dtype), batched


def CenteredGrid(values=0., boundary=0., bounds: Optional[Box] = None, resolution: Optional[Dict[str, int]] = None,
                 batch: Optional[int] = None, backend: Optional[HipBackend] = None, **resolution_) -> Field:
    """ `CenteredGrid(values, boundary, bounds, x=.., y=..)` (phi/field/_grid.py:21-86).
    values: number | array/tensor (optionally with leading batch dim) | callable(*coords) | Geometry (hard mask) | Field """
    backend = backend or default_backend()
    boundary = as_extrapolation(boundary)
    res, bounds = _resolve_grid_args(bounds, resolution, resolution_)
    shape = tuple(res.values())
    dtype = float_dtype()
    if isinstance(values, Field):
        assert values.is_centered and values.resolution == res
        t, batched = values.values.to(dtype), values.batched
    elif isinstance(values, Geometry):
        pts = _sample_points(res, bounds, None, boundary)
        t, batched = _tensor_from(values.lies_inside(pts).astype(np.float64), shape, backend, dtype)
    elif callable(values):
        pts = _sample_points(res, bounds, None, boundary)
        t, batched = _tensor_from(np.asarray(values(*pts), dtype=np.float64), shape, backend, dtype)
    elif isinstance(values, (int, float)):
        t, batched = torch.full((1,) + shape, float(values), dtype=dtype, device=backend.device), False
    else:
        t, batched = _tensor_from(values, shape, backend, dtype)
    if batch is not None and t.shape[0] != batch:
        assert t.shape[0] == 1
        t, batched = t.expand(batch, *shape).contiguous(), True
    return Field(res, bounds, boundary, t, False, backend, batched)


def StaggeredGrid(values=0., boundary=0., bounds: Optional[Box] = None, resolution: Optional[Dict[str, int]] = None,
                  batch: Optional[int] = None, backend: Optional[HipBackend] = None, **resolution_) -> Field:
    """ `StaggeredGrid(values, boundary, bounds, x=.., y=..)` (phi/field/_grid.py:89-176).
    values: number | per-component tuple of numbers | list of per-component arrays/tensors | callable(*coords) returning one
    array per component (evaluated at that component's face centres) | Geometry (hard mask at faces) | Field """
    backend = backend or default_backend()
    boundary = as_extrapolation(boundary)
    res, bounds = _resolve_grid_args(bounds, resolution, resolution_)
    D = len(res)
    dtype = float_dtype()
    shapes = [component_shape(res, boundary, d) for d in range(D)]
    comps, batched = [], False
    if isinstance(values, Field):
        assert values.is_staggered and values.resolution == res
        return values.with_boundary(boundary) if values.boundary != boundary else values
    for d in range(D):
        if isinstance(values, Geometry):
            pts = _sample_points(res, bounds, d, boundary)
            t, b = _tensor_from(values.lies_inside(pts).astype(np.float64), shapes[d], backend, dtype)
        elif callable(values):
            pts = _sample_points(res, bounds, d, boundary)
            out = values(*pts)
            t, b = _tensor_from(np.asarray(out[d], dtype=np.float64), shapes[d], backend, dtype)
        elif isinstance(values, (int, float)):
            t, b = torch.full((1,) + shapes[d], float(values), dtype=dtype, device=backend.device), False
        elif isinstance(values, (tuple, list)) and len(values) == D and all(isinstance(v, (int, float)) for v in values):
            t, b = torch.full((1,) + shapes[d], float(values[d]), dtype=dtype, device=backend.device), False
        elif isinstance(values, (tuple, list)) and len(values) == D:
            t, b = _tensor_from(values[d], shapes[d], backend, dtype)
        else:
            raise ValueError(f"cannot build a StaggeredGrid from {type(values)}")
        comps.append(t)
        batched = batched or b
    B = max(t.shape[0] for t in comps) if batch is None else batch
    if B > 1:
        comps = [t if t.shape[0] == B else t.expand(B, *t.shape[1:]).contiguous() for t in comps]
        batched = True
    return Field(res, bounds, boundary, comps, True, backend, batched)


# ---------------------------------------------------------------------------------------------------------------------
# operators backed by libphihip
# ---------------------------------------------------------------------------------------------------------------------
def _ptrs(tensors: Sequence[torch.Tensor]) -> List[int]:
    return [t.data_ptr() for t in tensors]


def divergence(field: Field, order: int = 2) -> Field:
    """ `field.divergence` of a StaggeredGrid, order 2 (phi/field/_field_math.py:589,617-626) -> CenteredGrid with
    extrapolation `field.extrapolation.spatial_gradient()`. """
    require_plain(field, 'divergence')
    if order != 2 or not field.is_staggered:
        raise NotImplementedError("the HIP backend implements divergence for StaggeredGrid, order=2 only")
    be = field.backend
    vals = [t.contiguous() for t in field.values]
    out = be.empty((field.batch_size,) + tuple(field.resolution.values()), field.dtype)
    be.ctx.divergence(field.grid_struct(), _ptrs(vals), 0, 1, False, out.data_ptr(), be.stream())
    return Field(field.resolution, field.bounds, field.boundary.spatial_gradient(), out, False, be, field.batched)


def spatial_gradient(field: Field, boundary=None, at: str = 'face', order: int = 2) -> Field:
    """ `field.spatial_gradient(p, boundary, at='face')` (phi/field/_field_math.py:148-236): gradient of a centred scalar
    at the faces that a StaggeredGrid with `boundary` stores; `field.boundary` pads p. """
    require_plain(field, 'spatial_gradient')
    if at != 'face' or order != 2 or field.is_staggered:
        raise NotImplementedError("the HIP backend implements spatial_gradient(at='face', order=2) of a CenteredGrid only")
    vb = as_extrapolation(boundary if boundary is not None else field.boundary.spatial_gradient())
    be = field.backend
    proto = Field(field.resolution, field.bounds, vb, None, True, be, field.batched)
    grid = _capi.make_grid(field.spatial_rank, _torch_dtype_code(field.dtype), field.batch_size, list(field.resolution.values()),
                           field.bounds.lower, field.bounds.upper, proto._codes, proto._bc_val)
    _check_pressure_padding(field.boundary, vb, field.dims)
    comps = [be.zeros((field.batch_size,) + component_shape(field.resolution, vb, d), field.dtype) for d in range(field.spatial_rank)]
    be.ctx.grad_subtract(grid, 0, 1, field.values.contiguous().data_ptr(), _ptrs(comps), be.stream())
    comps = [-c for c in comps]
    return Field(field.resolution, field.bounds, vb, comps, True, be, field.batched)


def _check_pressure_padding(p_ext: Extrapolation, v_ext: Extrapolation, dims):
    from .extrapolation import pressure_extrapolation
    expect = pressure_extrapolation(v_ext, dims)
    for d in dims:
        for upper in (False, True):
            have, want = p_ext.side(d, upper), expect.side(d, upper)
            if type(have) is not type(want) or (isinstance(have, ConstantExtrapolation) and have.component_value(0, d) != 0):
                raise NotImplementedError(f"gradient at faces: scalar extrapolation {have} on side {d}{'+' if upper else '-'} is not the one "
                                          f"the staggered boundary implies ({want}); only the pressure/velocity pairing is implemented")


def mean(field: Field):
    """ `field.mean`: mean over the spatial dims per batch entry (phi/field/_field_math.py:780-796) """
    require_plain(field, 'mean')
    if field.is_staggered:
        raise NotImplementedError
    m = field.values.reshape(field.batch_size, -1).mean(dim=1)
    return m if field.batched else m[0]


def resample(value, to: Field, soft: bool = False, balance: float = 0.5) -> Field:
    """ `resample(value, to=target)` / `value @ target` (phi/field/_resample.py:13-63). Implemented: same sample points (boundary
    change only), centred scalar [x constant vector] -> staggered faces on the same grid (sample_grid_at_faces,
    phi/field/_resample.py:272-276: mean of the two adjacent cells, padded with the scalar's extrapolation) as one HIP kernel
    per component (`phihip_centered_to_staggered`), and a `Geometry` -> mask on the target's sample points (set-up work, evaluated on
    the host): hard `lies_inside` or, with `soft=True`, `clip(balance - sdf / cell_radius, 0, 1)`
    (`resample(Sphere(...), to=smoke, soft=True)`, Smoke_Plume.ipynb cell 3; phi/field/_resample.py:192-210, phi/geom/_geom.py:278-308). """
    if isinstance(value, Geometry):
        radius = float(np.sqrt(sum((0.5 * h) ** 2 for h in to.dx)))

        def mask(pts):
            if soft:
                return np.clip(balance - value.approximate_signed_distance(pts) / radius, 0, 1)
            return value.lies_inside(pts).astype(np.float64)
        if to.is_staggered:
            comps = [_tensor_from(mask(_sample_points(to.resolution, to.bounds, d, to.boundary)), component_shape(to.resolution, to.boundary, d),
                                  to.backend, to.dtype) for d in range(to.spatial_rank)]
            B = max(t.shape[0] for t, _ in comps)
            return Field(to.resolution, to.bounds, to.boundary, [t if t.shape[0] == B else t.expand(B, *t.shape[1:]).contiguous() for t, _ in comps],
                         True, to.backend, any(b for _, b in comps))
        t, batched = _tensor_from(mask(_sample_points(to.resolution, to.bounds, None, to.boundary)), tuple(to.resolution.values()), to.backend, to.dtype)
        return Field(to.resolution, to.bounds, to.boundary, t, False, to.backend, batched)
    if value.is_staggered == to.is_staggered and same_grid(value, to) and value.vector_scale is None:
        if value.is_staggered and value.boundary != to.boundary:
            return value.with_boundary(to.boundary)
        return Field(value.resolution, value.bounds, to.boundary, value.values, value.is_staggered, value.backend, value.batched)
    if value.vector_scale is not None and to.is_centered:
        raise NotImplementedError("HIP backend: a `scalar * vector` field can only be resampled to a StaggeredGrid (centred vector fields "
                                  "are not implemented)")
    if value.is_centered and to.is_staggered and same_grid(value, to):
        be = value.backend
        scale = value.vector_scale or [1.0] * value.spatial_rank
        B = max(value.batch_size, to.batch_size)
        src = value.values if value.batch_size == B else value.values.expand(B, *value.values.shape[1:])
        src = src.contiguous()
        proto = Field(to.resolution, to.bounds, to.boundary, None, True, be, False)
        grid = _capi.make_grid(value.spatial_rank, _torch_dtype_code(value.dtype), B, list(to.resolution.values()), to.bounds.lower,
                               to.bounds.upper, proto._codes, proto._bc_val)
        comps = [be.empty((B,) + component_shape(to.resolution, to.boundary, d), value.dtype) for d in range(value.spatial_rank)]
        s_codes, s_vals = resolve(value.boundary, value.dims)
        s_val = [[s_vals[a][s][0] if isinstance(value.boundary.side(d, bool(s)), ConstantExtrapolation) else 0.0 for s in range(2)]
                 for a, d in enumerate(value.dims)]
        from . import autodiff
        if autodiff.needs_grad(src):
            meta = dict(be=be, grid=grid, s_codes=s_codes, s_val=s_val, vector=scale, shapes=[tuple(c.shape) for c in comps], dtype=value.dtype)
            comps = list(autodiff.CenteredToStaggered.apply(meta, src))
        else:
            be.ctx.centered_to_staggered(grid, src.data_ptr(), s_codes, s_val, scale, False, _ptrs(comps), be.stream())
        return Field(to.resolution, to.bounds, to.boundary, comps, True, be, value.batched or to.batched)
    from . import sampling
    return sampling.resample_general(value, to)     # different grids: one gather per (component of the) target


def vector_scaled(s: Field, vector: Sequence[float]) -> Field:
    """ `smoke * (0, 0.1)`: a centred scalar times a constant vector, kept lazily until it is resampled with `@`. """
    assert s.is_centered and len(vector) == s.spatial_rank
    vec = [float(c) for c in vector]
    if s.vector_scale is not None:             # (s * (0, 1)) * (2, 3): component-wise product of the vectors
        vec = [a * b for a, b in zip(s.vector_scale, vec)]
    return Field(s.resolution, s.bounds, s.boundary, s.values, False, s.backend, s.batched, vec)


def assert_close(*fields, rel_tolerance: float = 1e-5, abs_tolerance: float = 0, msg: str = ""):
    """ `field.assert_close` (values only) """
    ref = fields[0]
    ref_np = ref.numpy() if isinstance(ref, Field) else ref
    for other in fields[1:]:
        oth_np = other.numpy() if isinstance(other, Field) else other
        pairs = zip(ref_np, oth_np) if isinstance(ref_np, list) else [(ref_np, oth_np)]
        for a, b in pairs:
            np.testing.assert_allclose(np.asarray(b), np.asarray(a), rtol=rel_tolerance, atol=abs_tolerance, err_msg=msg)
