"""
Plug-in glue for a PhiFlow installation (reference: phi/__init__.py:41-63 backend detection, phi/torch/flow.py:15-35 backend
selection shim, phi/physics/fluid.py:94-162, phi/physics/advect.py:156-215).

`phiml` -- the package that owns PhiFlow's tensors, backends and `solve_linear` -- is neither vendored in the reference nor
installable in the build environment of this repository, so NOTHING in this module can be exercised by the test-suite beyond
"it imports and reports that phiml is missing". It is kept deliberately thin: all arithmetic stays in `phiflow_amd`
(libphihip); this file only converts `phi.field.Field` objects to `phiflow_amd.field.Field` and back, and installs the drop-in
functions. PhiML internals used here are limited to the public API PhiFlow itself uses ([PHIML-RECALL]):
`Tensor.native(order)`, `math.tensor(native, shape)`, `Extrapolation` singletons / `ConstantExtrapolation.value`, `Solve` fields.

    import phiflow_amd.phiml_plugin as hip
    hip.install()            # patches phi.physics.fluid.make_incompressible / phi.physics.advect.{semi_lagrangian, mac_cormack, advect}
    ...                      # unchanged PhiFlow user code; grids that are not order-2 uniform StaggeredGrids fall back to the reference
    hip.uninstall()
"""
from typing import Optional

from . import flow as _hip

_ORIGINALS = {}


def phiml_available() -> bool:
    try:
        import phiml  # noqa: F401
        import phi.field  # noqa: F401
        return True
    except Exception:
        return False


def _to_hip_extrapolation(ext, dims):
    """ phiml Extrapolation -> phiflow_amd Extrapolation (PERIODIC / ZERO / constants / BOUNDARY / per-side mixes) """
    from phiml.math import extrapolation as e
    if ext is e.PERIODIC:
        return _hip.PERIODIC
    if ext in (e.BOUNDARY, e.ZERO_GRADIENT):
        return _hip.BOUNDARY
    if isinstance(ext, e.ConstantExtrapolation):
        value = ext.value
        if value.shape.volume == 1:
            return _hip.ConstantExtrapolation(float(value))
        return _hip.ConstantExtrapolation({d: float(value.vector[d]) for d in dims})
    if hasattr(ext, 'ext'):                       # _MixedExtrapolation: {dim: (lower, upper)}
        return _hip.combine_sides({d: (_to_hip_extrapolation(lo, dims), _to_hip_extrapolation(up, dims)) for d, (lo, up) in ext.ext.items()})
    raise NotImplementedError(f"extrapolation {ext} has no HIP counterpart")


def to_hip(field):
    """ phi.field.Field (uniform grid, torch-ROCm natives) -> phiflow_amd Field. Batch dims are packed into one leading dim. """
    from phiml import math
    dims = field.resolution.names
    res = {d: int(field.resolution.get_size(d)) for d in dims}
    box = _hip.Box(**{d: (float(field.bounds.lower.vector[d]), float(field.bounds.upper.vector[d])) for d in dims})
    ext = _to_hip_extrapolation(field.extrapolation, dims)
    batch = math.batch(field.values) if hasattr(math, 'batch') else field.values.shape.batch
    packed = 'hipbatch'
    if field.is_staggered:
        comps = []
        for d in dims:
            c = field.values[{'~vector': d}]
            c = math.pack_dims(c, batch, math.batch(packed)) if batch else math.expand(c, math.batch(**{packed: 1}))
            comps.append(c.native([packed, *dims]))
        return _hip.StaggeredGrid(comps, ext, box, **res), batch
    v = math.pack_dims(field.values, batch, math.batch(packed)) if batch else math.expand(field.values, math.batch(**{packed: 1}))
    return _hip.CenteredGrid(v.native([packed, *dims]), ext, box, **res), batch


def from_hip(hfield, like, batch, extrapolation=None):
    """ phiflow_amd Field -> phi.field.Field with the geometry of `like` """
    from phiml import math
    from phi.field import Field
    dims = list(hfield.dims)
    spatial = lambda t: math.spatial(**{d: int(n) for d, n in zip(dims, t.shape[1:])})
    packed = math.batch(hipbatch=hfield.batch_size)

    def wrap(t):
        x = math.tensor(t, packed & spatial(t))
        return math.unpack_dim(x, 'hipbatch', batch) if batch else x.hipbatch[0]
    if hfield.is_staggered:
        values = math.stack({d: wrap(t) for d, t in zip(dims, hfield.values)}, math.dual(vector=dims))
        return Field(like.geometry, values, extrapolation if extrapolation is not None else like.extrapolation)
    return Field(like.geometry, wrap(hfield.values), extrapolation if extrapolation is not None else like.extrapolation)


def _to_hip_solve(solve):
    x0 = None
    if solve.x0 is not None:
        x0, _ = to_hip(solve.x0)
    return _hip.Solve(solve.method, solve.rel_tol, solve.abs_tol, x0=x0, max_iterations=int(solve.max_iterations),
                      suppress=tuple(_hip_exception(t) for t in (solve.suppress or ())))


def _hip_exception(phiml_type):
    name = getattr(phiml_type, '__name__', '')
    return {'NotConverged': _hip.NotConverged, 'Diverged': _hip.Diverged, 'ConvergenceException': _hip.ConvergenceException}.get(name, phiml_type)


def _supported(velocity, order=2, **kw) -> bool:
    try:
        return bool(velocity.is_grid and velocity.is_staggered and order == 2 and not kw.get('wide_stencil') and not kw.get('correct_skew'))
    except Exception:
        return False


def make_incompressible(velocity, obstacles=(), solve=None, active=None, order: int = 2, correct_skew=False, wide_stencil: Optional[bool] = None):
    """ drop-in for phi.physics.fluid.make_incompressible: HIP path for order-2 uniform StaggeredGrids, reference otherwise """
    from phi.physics import fluid as ref
    from phiml.math import Solve
    solve = Solve() if solve is None else solve
    if active is not None or not _supported(velocity, order, wide_stencil=wide_stencil, correct_skew=correct_skew) or solve.method not in _hip.Solve.METHODS:
        return _ORIGINALS.get('make_incompressible', ref.make_incompressible)(velocity, obstacles, solve, active, order, correct_skew, wide_stencil)
    hv, batch = to_hip(velocity)
    obs = [_to_hip_obstacle(o) for o in ref._get_obstacles_for(obstacles, velocity)]
    v, p = _hip.fluid.make_incompressible(hv, obs, _to_hip_solve(solve))
    # the pressure lives on the same UniformGrid (cell centres), like `Field(div.geometry, ...)` in fluid.py:149-151
    return from_hip(v, velocity, batch), from_hip(p, velocity, batch, ref._pressure_extrapolation(velocity.extrapolation))


def _to_hip_geometry(geo):
    """ phi.geom Box / Sphere, their unions (`union(boxes)` = one Box stacked along an instance dim, phi/geom/_geom_ops.py:316-317) and
    embedded geometries (`geom.infinite_cylinder`, phi/geom/_embed.py) -> phiflow_amd geometry """
    from phi.geom import Box, Sphere
    from phiml import math
    inst = math.instance(geo)
    if inst:
        return _hip.union([_to_hip_geometry(g) for g in math.unstack(geo, inst)])
    if type(geo).__name__ == '_EmbeddedGeometry':
        return _hip.embed(_to_hip_geometry(geo.geometry), tuple(geo.axes))
    dims = geo.vector.item_names
    if isinstance(geo, Sphere):
        return _hip.Sphere(float(geo.radius), **{d: float(geo.center.vector[d]) for d in dims})
    if isinstance(geo, Box):
        return _hip.Box(**{d: (float(geo.lower.vector[d]), float(geo.upper.vector[d])) for d in dims})
    raise NotImplementedError(f"obstacle geometry {type(geo).__name__}")


def _to_hip_obstacle(obstacle):
    geo = obstacle.geometry
    dims = geo.vector.item_names
    vel = obstacle.velocity
    vel = [float(vel.vector[d]) for d in dims] if hasattr(vel, 'vector') else float(vel)
    return _hip.Obstacle(_to_hip_geometry(geo), vel, float(obstacle.angular_velocity) if len(dims) == 2 else 0)


def _advect(name):
    def fn(field, velocity, dt, *args, **kwargs):
        from phi.physics import advect as ref
        integrator = kwargs.get('integrator', ref.euler)
        if integrator is not ref.euler or not (field.is_grid and velocity.is_grid and velocity.is_staggered):
            return _ORIGINALS.get(name, getattr(ref, name))(field, velocity, dt, *args, **kwargs)
        hf, batch = to_hip(field)
        hv, _ = to_hip(velocity)
        extra = {'correction_strength': kwargs['correction_strength']} if 'correction_strength' in kwargs else {}
        out = getattr(_hip.advect, name)(hf, hv, float(dt), **extra)
        return from_hip(out, field, batch)
    fn.__name__ = name
    return fn


semi_lagrangian = _advect('semi_lagrangian')
mac_cormack = _advect('mac_cormack')
advect = _advect('advect')


def install() -> bool:
    """ patches the reference's hot-path entry points with the HIP drop-ins; returns False (and does nothing) without phiml """
    if not phiml_available():
        return False
    from phi.physics import advect as ref_advect, fluid as ref_fluid
    if _ORIGINALS:
        return True
    _ORIGINALS.update(make_incompressible=ref_fluid.make_incompressible, semi_lagrangian=ref_advect.semi_lagrangian,
                      mac_cormack=ref_advect.mac_cormack, advect=ref_advect.advect)
    ref_fluid.make_incompressible = make_incompressible
    ref_advect.semi_lagrangian, ref_advect.mac_cormack, ref_advect.advect = semi_lagrangian, mac_cormack, advect
    return True


def uninstall():
    if not _ORIGINALS:
        return
    from phi.physics import advect as ref_advect, fluid as ref_fluid
    ref_fluid.make_incompressible = _ORIGINALS.pop('make_incompressible')
    ref_advect.semi_lagrangian = _ORIGINALS.pop('semi_lagrangian')
    ref_advect.mac_cormack = _ORIGINALS.pop('mac_cormack')
    ref_advect.advect = _ORIGINALS.pop('advect')
