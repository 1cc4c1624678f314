"""
Boundary conditions (= extrapolations) for grids, restricted to what PhiFlow's grid fluid step uses
(`phiml.math.extrapolation`: PERIODIC, ZERO / constants, BOUNDARY = ZERO_GRADIENT, per-side mixes via `combine_sides`;
reference usage: tests/commit/physics/test_fluid.py:34-53, docs/Fields.md:95-125).
"""
from typing import Dict, Optional, Sequence, Tuple, Union

from ._capi import BC_CLOSED, BC_OPEN, BC_PERIODIC


class Extrapolation:
    """ Base class. Concrete kinds: `_Periodic`, `_Boundary`, `ConstantExtrapolation`, `_Mixed`. """

    def side(self, dim: str, upper: bool) -> 'Extrapolation':
        return self

    # --- protocol used by phi.field / phi.physics ---
    def valid_outer_faces(self, dim: str) -> Tuple[bool, bool]:
        lo, up = self.side(dim, False), self.side(dim, True)
        return lo._face_valid(False), up._face_valid(True)

    def _face_valid(self, upper: bool) -> bool:
        raise NotImplementedError

    def _code(self) -> int:
        raise NotImplementedError

    @property
    def is_flexible(self) -> bool:
        """ True if the boundary values adapt to the field (BOUNDARY): no divergence balancing needed (fluid.py:145) """
        return False

    def spatial_gradient(self) -> 'Extrapolation':
        raise NotImplementedError

    def is_periodic(self, dim: str) -> bool:
        return isinstance(self.side(dim, False), _Periodic)

    def __eq__(self, other):
        return isinstance(other, Extrapolation) and repr(self) == repr(other)

    def __hash__(self):
        return hash(repr(self))


class _Periodic(Extrapolation):
    def _face_valid(self, upper):
        return not upper

    def _code(self):
        return BC_PERIODIC

    def spatial_gradient(self):
        return self

    def __repr__(self):
        return "periodic"


class _Boundary(Extrapolation):
    """ BOUNDARY / ZERO_GRADIENT: copy the edge value outwards """

    def _face_valid(self, upper):
        return True

    def _code(self):
        return BC_OPEN

    @property
    def is_flexible(self):
        return True

    def spatial_gradient(self):
        return ZERO

    def __repr__(self):
        return "zero-gradient"


class ConstantExtrapolation(Extrapolation):
    """ constant value outside; `value` is a number or a per-component vector (dict dim->value or sequence) """

    def __init__(self, value: Union[float, Dict[str, float], Sequence[float]]):
        if isinstance(value, dict):
            value = {k: float(v) for k, v in value.items()}
        elif isinstance(value, (tuple, list)):
            value = tuple(float(v) for v in value)
        else:
            value = float(value)
        self.value = value

    def component_value(self, comp: int, comp_name: Optional[str]) -> float:
        v = self.value
        if isinstance(v, dict):
            return float(v.get(comp_name, 0.0))
        if isinstance(v, (tuple, list)):
            return float(v[comp])
        return float(v)

    def _face_valid(self, upper):
        return False

    def _code(self):
        return BC_CLOSED

    def spatial_gradient(self):
        return ZERO

    def __repr__(self):
        return f"{self.value}"


class _Mixed(Extrapolation):
    """ per-axis / per-side extrapolations (`combine_sides`) """

    def __init__(self, ext_by_dim: Dict[str, Tuple[Extrapolation, Extrapolation]]):
        self.ext = ext_by_dim

    def side(self, dim, upper):
        lo, up = self.ext[dim]
        return up if upper else lo

    @property
    def is_flexible(self):
        return any(e.is_flexible for pair in self.ext.values() for e in pair)

    def spatial_gradient(self):
        return _Mixed({d: (lo.spatial_gradient(), up.spatial_gradient()) for d, (lo, up) in self.ext.items()})

    def __repr__(self):
        return repr({d: pair for d, pair in self.ext.items()})


PERIODIC = _Periodic()
BOUNDARY = _Boundary()
ZERO_GRADIENT = BOUNDARY
ZERO = ConstantExtrapolation(0)
ONE = ConstantExtrapolation(1)


def as_extrapolation(obj) -> Extrapolation:
    """ numbers -> constant; dict {'x': ext, 'y-': ext, 'y+': ext} -> combine_sides (reference: `as_boundary`,
    phi/field/_field.py:850; Lid_Driven_Cavity.ipynb passes {'x': 0, 'y-': 0, 'y+': vec(x=1, y=0)}) """
    if isinstance(obj, Extrapolation):
        return obj
    if obj is None:
        return ZERO
    from .geom import Vector
    if isinstance(obj, Vector):
        return ConstantExtrapolation(dict(obj))
    if isinstance(obj, dict):
        sides: Dict[str, list] = {}
        for key, val in obj.items():
            e = as_extrapolation(val)
            if key.endswith('-'):
                sides.setdefault(key[:-1], [None, None])[0] = e
            elif key.endswith('+'):
                sides.setdefault(key[:-1], [None, None])[1] = e
            else:
                sides[key] = [e, e]
        assert all(lo is not None and up is not None for lo, up in sides.values()), f"incomplete boundary specification {obj}"
        return _Mixed({d: (lo, up) for d, (lo, up) in sides.items()})
    if isinstance(obj, (int, float, tuple, list)):
        return ConstantExtrapolation(obj)
    raise ValueError(f"cannot interpret {obj!r} as an extrapolation")


def combine_sides(boundary_dict: Optional[dict] = None, **extrapolations) -> Extrapolation:
    """ `combine_sides(x=BOUNDARY, y=(ZERO, BOUNDARY))` (tests/commit/physics/test_fluid.py:51) """
    spec = dict(boundary_dict or {}, **extrapolations)
    out = {}
    for dim, e in spec.items():
        if isinstance(e, (tuple, list)):
            assert len(e) == 2, "per-axis boundaries are given as (lower, upper)"
            out[dim] = (as_extrapolation(e[0]), as_extrapolation(e[1]))
        else:
            out[dim] = (as_extrapolation(e), as_extrapolation(e))
    return _Mixed(out)


def resolve(ext: Extrapolation, dims: Sequence[str]):
    """ -> (codes[D][2], values[D][2][D]) for the C ABI `phihip_grid.bc / bc_val`. Extrapolations are immutable: the result is cached on
    the object (a field is constructed ~10 times per simulation step; callers must not modify the returned lists). """
    cache = ext.__dict__.setdefault('_resolved', {})
    key = tuple(dims)
    if key not in cache:
        cache[key] = _resolve(ext, key)
    return cache[key]


def _resolve(ext: Extrapolation, dims: Sequence[str]):
    D = len(dims)
    codes = [[0, 0] for _ in range(D)]
    vals = [[[0.0] * D for _ in range(2)] for _ in range(D)]
    for a, dim in enumerate(dims):
        for s, upper in enumerate((False, True)):
            e = ext.side(dim, upper)
            if isinstance(e, _Mixed):
                raise ValueError("nested mixed extrapolations are not supported")
            codes[a][s] = e._code()
            if isinstance(e, ConstantExtrapolation):
                for c, cname in enumerate(dims):
                    vals[a][s][c] = e.component_value(c, cname)
        if (codes[a][0] == BC_PERIODIC) != (codes[a][1] == BC_PERIODIC):
            raise ValueError(f"axis {dim}: PERIODIC must be used on both sides")
    return codes, vals


def pressure_extrapolation(vext: Extrapolation, dims: Sequence[str]) -> Extrapolation:
    """ fluid._pressure_extrapolation (phi/physics/fluid.py:264-274) """
    def conv(e):
        if isinstance(e, _Periodic):
            return PERIODIC
        if isinstance(e, _Boundary):
            return ZERO
        return BOUNDARY
    cache = vext.__dict__.setdefault('_pressure_ext', {})       # (extrapolations are immutable: one result object per velocity boundary, so `resolve` of it is cached too)
    key = tuple(dims)
    if key not in cache:
        cache[key] = _Mixed({d: (conv(vext.side(d, False)), conv(vext.side(d, True))) for d in dims})
    return cache[key]
