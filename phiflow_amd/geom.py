"""
Geometry subset needed by the grid fluid step: `Box` / `Cuboid` (domain bounds and box obstacles, optionally rotated) and
`Sphere` (reference: phi/geom/_box.py:46-236, phi/geom/_sphere.py). Obstacles are rasterised by device kernels
(`phihip_obstacle_accessible`, `phihip_apply_obstacles`); the host-side `lies_inside` / `approximate_signed_distance` below only
serve field initialisers such as `CenteredGrid(Sphere(...))`.
"""
from typing import Dict, Sequence, Tuple

import numpy as np


class Vector(dict):
    """ named vector `vec(x=1, y=0)` with elementwise arithmetic (`center + velocity * dt`, `x % domain.size`) """

    def _zip(self, other, fn):
        if isinstance(other, dict):
            return Vector({k: fn(v, other[k]) for k, v in self.items()})
        if isinstance(other, (tuple, list)):
            assert len(other) == len(self)
            return Vector({k: fn(v, o) for (k, v), o in zip(self.items(), other)})
        return Vector({k: fn(v, other) for k, v in self.items()})

    def __add__(self, o): return self._zip(o, lambda a, b: a + b)
    def __radd__(self, o): return self._zip(o, lambda a, b: b + a)
    def __sub__(self, o): return self._zip(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._zip(o, lambda a, b: b - a)
    def __mul__(self, o): return self._zip(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._zip(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._zip(o, lambda a, b: a / b)
    def __mod__(self, o): return self._zip(o, lambda a, b: a % b)
    def __neg__(self): return Vector({k: -v for k, v in self.items()})


def vec(**components) -> Vector:
    """ `vec(x=1, y=0)`: named vector, e.g. a wall velocity for a constant extrapolation """
    return Vector(components)


class Geometry:
    dims: Tuple[str, ...]
    batch_size = 1          # > 1: `Batched` (one geometry per batch entry)

    def lies_inside(self, points: Sequence[np.ndarray]) -> np.ndarray:
        raise NotImplementedError

    def approximate_signed_distance(self, points: Sequence[np.ndarray]) -> np.ndarray:
        raise NotImplementedError

    def entry(self, b: int) -> 'Geometry':
        """ the geometry of batch entry b """
        return self


def _batch_len(*values) -> int:
    """ length of the batch dimension among constructor arguments (numbers or 1-D sequences of equal length), 0 if none """
    n = 0
    for v in values:
        if isinstance(v, (list, tuple, np.ndarray)) and np.ndim(v) == 1:
            assert n in (0, len(v)), "batched geometry arguments must have the same length"
            n = len(v)
    return n


def _pick(value, b):
    if isinstance(value, (list, tuple, np.ndarray)) and np.ndim(value) == 1:
        return float(value[b])
    if isinstance(value, (list, tuple)):        # (lower, upper) pair of a Box bound, either may be batched
        return tuple(_pick(v, b) for v in value)
    return value


class Batched(Geometry):
    """ One geometry per batch entry, e.g. `Sphere(x=[40, 50, 60], y=9.5, radius=5)` or `Cuboid(vec(x=[15, 50, 70], y=60), half_size=...)`
    (examples/grids/Batched_Smoke.ipynb: `tensor([...], batch('setting'))` coordinates). `lies_inside` / `approximate_signed_distance`
    return arrays with a leading batch axis, so fields initialised from it are batched. """

    def __init__(self, geometries: Sequence[Geometry]):
        self.geometries = tuple(geometries)
        self.dims = self.geometries[0].dims
        self.batch_size = len(self.geometries)

    def entry(self, b):
        return self.geometries[b % len(self.geometries)]

    def lies_inside(self, points):
        return np.stack([g.lies_inside(points) for g in self.geometries])

    def approximate_signed_distance(self, points):
        return np.stack([g.approximate_signed_distance(points) for g in self.geometries])

    def shifted(self, delta):
        return Batched([g.shifted(delta) for g in self.geometries])

    def at(self, center):
        return Batched([g.at(center) for g in self.geometries])

    def rotated(self, angle):
        return Batched([g.rotated(angle) for g in self.geometries])

    def __repr__(self):
        return "batched(" + ", ".join(repr(g) for g in self.geometries) + ")"


class Union(Geometry):
    """ `union(*geometries)` (phi/geom/_geom_ops.py:297-319): a point lies inside if it lies inside any member; the signed distance is
    the minimum over the members (phi/geom/_geom_ops.py:96-102, phi/geom/_box.py:235) """

    def __init__(self, geometries: Sequence[Geometry]):
        self.geometries = tuple(geometries)
        assert self.geometries, "empty union"
        self.dims = self.geometries[0].dims
        assert all(set(g.dims) == set(self.dims) for g in self.geometries), "members of a union must share their dimensions"
        self.batch_size = max(g.batch_size for g in self.geometries)

    def entry(self, b):
        return self if self.batch_size == 1 else Union([g.entry(b) for g in self.geometries])

    def lies_inside(self, points):
        if self.batch_size > 1:
            return np.stack([self.entry(b).lies_inside(points) for b in range(self.batch_size)])
        out = None
        for g in self.geometries:
            pts = [points[self.dims.index(d)] for d in g.dims]
            inside = g.lies_inside(pts)
            out = inside if out is None else (out | inside)
        return out

    def approximate_signed_distance(self, points):
        if self.batch_size > 1:
            return np.stack([self.entry(b).approximate_signed_distance(points) for b in range(self.batch_size)])
        out = None
        for g in self.geometries:
            pts = [points[self.dims.index(d)] for d in g.dims]
            dist = g.approximate_signed_distance(pts)
            out = dist if out is None else np.minimum(out, dist)
        return out

    def shifted(self, delta):
        return Union([g.shifted(delta) for g in self.geometries])

    def __repr__(self):
        return "union(" + ", ".join(repr(g) for g in self.geometries) + ")"


def union(*geometries) -> Geometry:
    """ `union(geometries)` / `union(g1, g2, ...)`; nested unions are flattened, a single geometry is returned as it is """
    geometries = geometries[0] if len(geometries) == 1 and isinstance(geometries[0], (tuple, list)) else geometries
    flat = []
    for g in geometries:
        flat.extend(g.geometries if isinstance(g, Union) else [g])
    return flat[0] if len(flat) == 1 else Union(flat)


class Embedded(Geometry):
    """ `embed(geometry, dims)`: the geometry is constant along the added dims, as if it were infinitely long there
    (phi/geom/_embed.py:15-66,106-136); like the reference's it cannot be shifted or rotated """

    def __init__(self, geometry: Geometry, dims: Sequence[str]):
        self.geometry = geometry
        self.dims = tuple(dims)
        self.batch_size = geometry.batch_size
        assert set(geometry.dims) < set(self.dims), f"embedding {geometry.dims} in {self.dims} adds no dimension"

    def entry(self, b):
        return self if self.batch_size == 1 else Embedded(self.geometry.entry(b), self.dims)

    def _down_project(self, points):
        return [points[self.dims.index(d)] for d in self.geometry.dims]

    def lies_inside(self, points):
        return self.geometry.lies_inside(self._down_project(points))

    def approximate_signed_distance(self, points):
        return self.geometry.approximate_signed_distance(self._down_project(points))

    def __repr__(self):
        return f"embed({self.geometry!r}, {self.dims})"


def embed(geometry: Geometry, projected_dims) -> Geometry:
    """ `embed(geometry, 'z')` / `embed(geometry, ('x', 'y', 'z'))`: dims the geometry already has are ignored; dims of the geometry
    that are not listed come first (phi/geom/_embed.py:106-136) """
    if projected_dims is None:
        return geometry
    axes = tuple(d.strip() for d in projected_dims.split(',')) if isinstance(projected_dims, str) else tuple(projected_dims)
    if all(a in geometry.dims for a in axes):
        return geometry
    for name in reversed(geometry.dims):
        if name not in axes:
            axes = (name,) + axes
    return Embedded(geometry, axes)


def infinite_cylinder(center=None, radius=None, inf_dim=None, **center_) -> Geometry:
    """ `geom.infinite_cylinder(x=20, y=50, radius=10, inf_dim='z')` (examples/grids/Wake_Flow.ipynb; phi/geom/_embed.py:139-158):
    an n-dimensional `Sphere` embedded in n + 1 dimensions """
    if center is not None:
        center_ = dict(center)
    return embed(Sphere(radius, **center_), inf_dim)


class _BoxType(type):
    def __getitem__(cls, item):
        """ `Box['x,y', 0:100, 0:100]` (tests/commit/physics/test_fluid.py:23) """
        assert isinstance(item, tuple) and isinstance(item[0], str), "use Box['x,y', 0:1, 0:1]"
        dims = [d.strip() for d in item[0].split(',')]
        assert len(item) == len(dims) + 1
        kwargs = {}
        for d, sl in zip(dims, item[1:]):
            kwargs[d] = (0 if sl.start is None else sl.start, sl.stop)
        return cls(**kwargs)


class Box(Geometry, metaclass=_BoxType):
    """ box; `Box(x=100, y=(10, 20))`: a number means (0, number). `rot`: optional (D, D) matrix box frame -> world
    (set by `rotated`); bounds describe the box in its own frame around `center`. """

    def __new__(cls, _rot=None, **bounds):
        # a bound is a number (0, upper), a (lower, upper) pair, or -- batched -- a pair holding 1-D sequences / a 1-D ndarray of uppers
        pairs = {d: (b if isinstance(b, (tuple, list)) and len(b) == 2 else (0.0, b)) for d, b in bounds.items()}
        n = _batch_len(*[x for pair in pairs.values() for x in pair])
        if n:
            return Batched([Box(_rot=_rot, **{d: (_pick(lo, i), _pick(up, i)) for d, (lo, up) in pairs.items()}) for i in range(n)])
        return super().__new__(cls)

    def __init__(self, _rot=None, **bounds):
        self.dims = tuple(bounds.keys())
        self.lower = tuple(float(b[0]) if isinstance(b, (tuple, list)) else 0.0 for b in bounds.values())
        self.upper = tuple(float(b[1]) if isinstance(b, (tuple, list)) else float(b) for b in bounds.values())
        self.rot = None if _rot is None else np.asarray(_rot, dtype=float)

    @property
    def size(self):
        return tuple(u - l for l, u in zip(self.lower, self.upper))

    @property
    def center(self):
        return tuple((u + l) / 2 for l, u in zip(self.lower, self.upper))

    @property
    def half_size(self):
        return tuple((u - l) / 2 for l, u in zip(self.lower, self.upper))

    def _local(self, points):
        """ global_to_local(scale=False, origin='center') = R^T (x - center)  (phi/geom/_box.py:134-152) """
        r = [points[a] - self.center[a] for a in range(len(self.dims))]
        if self.rot is None:
            return r
        return [sum(self.rot[c][a] * r[c] for c in range(len(r))) for a in range(len(r))]

    def lies_inside(self, points):
        """ |local| <= half, inclusive (phi/geom/_box.py:174-185) """
        ok = np.ones(np.shape(points[0]), dtype=bool)
        for a, p in enumerate(self._local(points)):
            ok &= np.abs(p) <= self.half_size[a]
        return ok

    def approximate_signed_distance(self, points):
        """ L-infinity distance to the surface (phi/geom/_box.py:217-236) """
        dist = None
        for a, p in enumerate(self._local(points)):
            da = np.abs(p) - self.half_size[a]
            dist = da if dist is None else np.maximum(dist, da)
        return dist

    def rotated(self, angle):
        """ `Box.rotated(angle)` (phi/geom/_box.py:127-129): 2-D: counter-clockwise angle in radians; 3-D: a (3, 3) rotation matrix """
        if len(self.dims) == 2:
            c, s = float(np.cos(angle)), float(np.sin(angle))
            R = np.array([[c, -s], [s, c]])
        else:
            R = np.asarray(angle, dtype=float)
            assert R.shape == (3, 3), "3-D boxes rotate by a (3, 3) rotation matrix"
        rot = R if self.rot is None else self.rot @ R
        return Box(_rot=rot, **{d: (l, u) for d, l, u in zip(self.dims, self.lower, self.upper)})

    def shifted(self, delta):
        """ `geometry.shifted(delta)`; delta: sequence or dict dim -> offset """
        delta = [delta.get(d, 0.0) for d in self.dims] if isinstance(delta, dict) else list(delta)
        return Box(_rot=self.rot, **{d: (l + dd, u + dd) for d, l, u, dd in zip(self.dims, self.lower, self.upper, delta)})

    def at(self, center):
        center = [center.get(d, c) for d, c in zip(self.dims, self.center)] if isinstance(center, dict) else list(center)
        return Box(_rot=self.rot, **{d: (c - h, c + h) for d, c, h in zip(self.dims, center, self.half_size)})

    def __repr__(self):
        rot = "" if self.rot is None else ", rot=" + np.array2string(self.rot, precision=12, separator=",").replace("\n", "")
        return "Box(" + ", ".join(f"{d}=({l}, {u})" for d, l, u in zip(self.dims, self.lower, self.upper)) + rot + ")"


def Cuboid(center=None, half_size=None, **size) -> Box:
    """ `Cuboid(vec(x=20, y=80), x=20, y=20)`: box given by centre and edge lengths, or `Cuboid(center, half_size=vec(x=15, y=10))`
    (phi/geom/_box.py:418-440); coordinates may be batched (1-D sequences); without a centre the arguments are bounds like `Box(...)` """
    if center is None and half_size is None:
        return Box(**size)
    if half_size is not None:
        half = dict(half_size) if isinstance(half_size, dict) else {d: half_size for d in center}
    else:
        half = {d: np.asarray(s, dtype=float) / 2 for d, s in size.items()}
    c = dict(center) if isinstance(center, dict) else dict(zip(half, center))
    n = _batch_len(*c.values(), *[h for h in half.values() if np.ndim(h) == 1])
    def one(i):
        return Box(**{d: (float(_pick(c[d], i)) - float(_pick(half[d], i)), float(_pick(c[d], i)) + float(_pick(half[d], i))) for d in half})
    return Batched([one(i) for i in range(n)]) if n else one(0)


class Sphere(Geometry):
    """ `Sphere(x=50, y=10, radius=5)` """

    def __new__(cls, radius=None, **center):
        n = _batch_len(radius, *center.values())
        if n:
            return Batched([Sphere(_pick(radius, i), **{d: _pick(c, i) for d, c in center.items()}) for i in range(n)])
        return super().__new__(cls)

    def __init__(self, radius: float, **center):
        self.dims = tuple(center.keys())
        self.center = tuple(float(c) for c in center.values())
        self.radius = float(radius)

    def lies_inside(self, points):
        d2 = sum((p - c) ** 2 for p, c in zip(points, self.center))
        return d2 <= self.radius ** 2

    def approximate_signed_distance(self, points):
        d2 = sum((p - c) ** 2 for p, c in zip(points, self.center))
        return np.sqrt(d2) - self.radius

    def shifted(self, delta):
        delta = [delta.get(d, 0.0) for d in self.dims] if isinstance(delta, dict) else list(delta)
        return Sphere(self.radius, **{d: c + dd for d, c, dd in zip(self.dims, self.center, delta)})

    def at(self, center):
        center = [center.get(d, c) for d, c in zip(self.dims, self.center)] if isinstance(center, dict) else list(center)
        return Sphere(self.radius, **dict(zip(self.dims, center)))

    def __repr__(self):
        return f"Sphere({dict(zip(self.dims, self.center))}, radius={self.radius})"
